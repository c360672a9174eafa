set -u
OUT=gpurun_out/r05t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_image.py -m gpu -q -p no:cacheprovider > $OUT/tests_image.log 2>&1; echo "image rc=$?"; tail -4 $OUT/tests_image.log
timeout -k 10 300 python bench.py --cpu-images 0 --alt-precision none --alt-injected 0 --alt-config5 0 --measure-traffic 0 --roofline-steps 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05t/bench.json")); print(d["value"])
for a in d["alt_configs"]: print({k:v for k,v in a.items() if k in ("value","ms_per_step","detect_over_predict","split_ms","predict_only_on_the_same_molded_batch")})
PY
