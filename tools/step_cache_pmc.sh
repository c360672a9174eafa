#!/bin/bash
# L2 / fabric / LDS counters of every conv kernel of the benchmark step (tools/profile_step.py), one counter group per
# rocprofv3 pass -> gpurun_out/$1/cache_lds_counters.json (committed as profiles/r03_cache_lds_counters.json)
set -u
OUT=gpurun_out/${1:-cc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0; files=""
for ctrs in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"; do
    i=$((i+1)); d=$OUT/pass_$i; mkdir -p $d
    timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -o p -- python3 tools/profile_step.py --steps 1 --meta $d/meta.json > $d/log.txt 2>&1
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i ($ctrs) was killed: stopping"; exit $rc; fi
    find $d -name "*_kernel_trace.csv" -delete
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ $rc -ne 0 ] || [ -z "$f" ]; then echo "pass $i ($ctrs) failed (rc=$rc):"; tail -3 $d/log.txt; continue; fi
    files="$files $f"; meta=$d/meta.json
done
python3 profiles/summarize_pmc.py counters $meta $files > $OUT/cache_lds_counters.json && find $OUT -name "*counter_collection.csv" -delete
python3 - $OUT/cache_lds_counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in list(d["by_instantiation"].items())[:12]:
    print(k[:50].ljust(50), {a: v[a] for a in ("launches_per_step", "l2_hit_rate", "fabric_read_GB_at_128B_per_request", "lds_bank_conflict_share_of_lds_active") if a in v})
PY
