set -u
OUT=gpurun_out/r05c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_conv.py -m gpu -q -p no:cacheprovider -k "nms or stem or cpu_tensors" > $OUT/tests_a.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests_a.log
timeout -k 10 300 python bench.py --cpu-images 0 --alt-precision none --alt-config5 0 --alt-injected 0 --measure-traffic 0 --dump-conv $OUT/conv_on.json > $OUT/bench_on.json 2> $OUT/bench_on.err; echo "on rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05c/bench_on.json")); print(d["value"], d["ms_per_step"], d["roofline"]["by_kernel"].get("stem")); print([o for o in d["roofline_ops"] if "nms" in o["op"]])
PY
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_dist.py -m gpu -q -p no:cacheprovider > $OUT/tests_b.log 2>&1; echo "tests_b rc=$?"; tail -5 $OUT/tests_b.log
