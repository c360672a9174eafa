set -u
OUT=gpurun_out/f16lds; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0; files=""
for ctrs in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); d=$OUT/pass_$i; mkdir -p $d
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -o p -- python3 tools/profile_step.py --steps 1 --arch resnet101 --height 832 --width 1344 --precision f16 --meta $d/meta.json > $d/log.txt 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo killed; exit $rc; fi
  find $d -name "*_kernel_trace.csv" -delete
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "pass $i failed rc=$rc"; tail -3 $d/log.txt; continue; fi
  files="$files $f"; meta=$d/meta.json
done
python3 profiles/summarize_pmc.py counters $meta $files > $OUT/f16_lds_counters.json && find $OUT -name "*counter_collection.csv" -delete
python3 - $OUT/f16_lds_counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in list(d["by_instantiation"].items()):
    if "f16p" in k or "igemm_f16" in k:
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
        la = v.get("SQ_LDS_IDX_ACTIVE", 0)
        print(k[:46].ljust(46), "launches", v["launches_per_step"], "cyc/step(M)", round(cyc/1e6,3), "LDS_IDX_ACTIVE/(256 CU x cyc)", round(la/(256*cyc),3) if cyc else None, "conflict share", v.get("lds_bank_conflict_share_of_lds_active"))
PY
