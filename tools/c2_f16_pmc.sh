#!/bin/bash
# Counter passes over bottleneck_c2_f16 at configs[4]'s size: gpurun_out/$1/c2_pmc_<kind>_<pass>/
set -u
TAG=${1:-r05c2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
for kind in identity first; do
  python3 tools/c2_f16_probe.py $kind 30 >> $OUT/c2_probe.jsonl
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1)); d=$OUT/c2_pmc_${kind}_$i; mkdir -p $d
    C2_PROBE_PER_LAYER=0 timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -o p -- python3 tools/c2_f16_probe.py $kind 6 > $d.log 2>&1 || tail -3 $d.log
    find $d -name "*_kernel_trace.csv" -delete
  done
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("$OUT/c2_pmc_*/*counter_collection.csv")):
    kind = f.split("c2_pmc_")[1].split("_")[0]
    acc = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f)):
        if "bottleneck_c2_f16" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    e = res.setdefault(kind, {})
    for c, v in acc.items():
        e[c] = round(sum(v[2:]) / max(1, len(v) - 2), 1)
    if dur:
        e.setdefault("us", []).append(round(sorted(dur)[len(dur) // 2], 1))
json.dump(res, open("$OUT/c2_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*counter_collection.csv" -delete
cat $OUT/c2_probe.jsonl
