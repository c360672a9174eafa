set -u
OUT=gpurun_out/r05s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o det -- python3 bench.py --cpu-images 0 --alt-precision none --alt-injected 0 --alt-config5 0 --measure-traffic 0 --roofline-steps 0 --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
find $OUT/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/prof
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r05s/kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if any(s in n for s in ("conv","wino","gemm","Cijk")): continue
    print(r["Calls"], round(float(r["AverageNs"])/1e3,1), round(float(r["TotalDurationNs"])/1e6,2), n[:110])
PY
