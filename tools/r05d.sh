set -u
OUT=gpurun_out/r05d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_image.py -m gpu -q -p no:cacheprovider -x > $OUT/tests_image.log 2>&1; echo "image rc=$?"; tail -5 $OUT/tests_image.log
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider > $OUT/tests_full.log 2>&1; echo "fullsize rc=$?"; tail -8 $OUT/tests_full.log
cp gpurun_out/parity_fullsize.json $OUT/parity_fullsize.json 2>/dev/null
timeout -k 10 900 python tools/fp64_truth.py --attribute 1 > $OUT/fp64_attribution.jsonl 2> $OUT/fp64_attribution.err; echo "attr rc=$?"; tail -1 $OUT/fp64_attribution.jsonl | cut -c1-1800
timeout -k 10 300 python bench.py --cpu-images 0 --alt-precision none --alt-config5 0 --alt-injected 0 --measure-traffic 0 > $OUT/bench_detect.json 2> $OUT/bench_detect.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05d/bench_detect.json")); print(d["value"]); print([a for a in d["alt_configs"]])
PY
