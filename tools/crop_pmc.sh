#!/bin/bash
# PMC passes over both NCHW crop kernels on the configs[1] shape (level $1, default 256): gpurun_out/$2/crop_pmc_<pass>/
set -u
LVL=${1:-256}; TAG=${2:-r03a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"; do
    i=$((i+1))
    for mode in 1 0; do
    d=$OUT/crop_pmc_${LVL}_${i}m$mode
    mkdir -p $d
    timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $d -o p -- python3 tools/crop_pmc.py $LVL $mode > $d.log 2>&1 || { tail -5 $d.log; }
    find $d -name "*_kernel_trace.csv" -delete
    done
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/crop_pmc_${LVL}_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "crop" not in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k[:60], {c: round(sum(v[1:]) / max(1, len(v) - 1), 1) for c, v in d.items()})
PY
