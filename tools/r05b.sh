set -u
OUT=gpurun_out/r05b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -p no:cacheprovider -k "stem" > $OUT/tests_stem.log 2>&1; echo "stem tests rc=$?"; tail -5 $OUT/tests_stem.log
timeout -k 10 300 python bench.py --cpu-images 0 --alt-precision none --alt-config5 0 --alt-injected 0 --measure-traffic 0 --dump-conv $OUT/conv_on.json > $OUT/bench_on.json 2> $OUT/bench_on.err; echo "on rc=$?"
MRCNN_STEM_POOL=0 timeout -k 10 300 python bench.py --cpu-images 0 --alt-precision none --alt-config5 0 --alt-injected 0 --measure-traffic 0 --dump-conv $OUT/conv_off.json > $OUT/bench_off.json 2> $OUT/bench_off.err; echo "off rc=$?"
timeout -k 10 300 python bench.py --cpu-images 0 --alt-precision none --alt-config5 0 --alt-injected 0 --measure-traffic 0 > $OUT/bench_on2.json 2> $OUT/bench_on2.err; echo "on2 rc=$?"
timeout -k 10 200 python tools/fc1_probe.py > $OUT/fc1_probe.jsonl 2> $OUT/fc1_probe.err; echo "probe rc=$?"
MRCNN_CONV_TILE=5 timeout -k 10 200 python tools/fc1_probe.py >> $OUT/fc1_probe.jsonl 2>> $OUT/fc1_probe.err
python - <<'PY'
import json
for n in ("on","off","on2"):
    try:
        d=json.load(open(f"gpurun_out/r05b/bench_{n}.json")); print(n, d["value"], d["ms_per_step"], d["roofline"]["by_kernel"].get("stem"))
    except Exception as e: print(n, "ERR", e)
PY
cat $OUT/fc1_probe.jsonl
