#!/bin/bash
# time + fabric reads (FETCH_SIZE) of the F(4x4) kernel on the P2 / P3 RPN layers (plain and heads variants): gpurun_out/$1/
set -u
TAG=${1:-r03w}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 120 python tools/w4_time.py 8 256 256 256 512 > $OUT/w4_time_p2.txt 2>&1; cat $OUT/w4_time_p2.txt
timeout -k 10 120 python tools/w4_time.py 8 128 128 256 512 > $OUT/w4_time_p3.txt 2>&1; cat $OUT/w4_time_p3.txt
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  d=$OUT/w4_pmc_$(echo $pass | cut -c1-5)
  mkdir -p $d
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -o p -- python3 tools/w4_time.py 8 256 256 256 512 > $d.log 2>&1
  find $d -name "*_kernel_trace.csv" -delete
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/w4_pmc_*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wino4_f32" not in k: continue
        k = "heads" if "true, true" in k or "1, 1" in k.split("<")[1] else k.split("<")[1][:20]
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: round(sum(v[1:]) / max(1, len(v) - 1), 1) for c, v in d.items()})
PY
