set -u
OUT=gpurun_out/r05f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -12 $OUT/tests.log
cp gpurun_out/parity_fullsize.json $OUT/parity_fullsize.json 2>/dev/null
timeout -k 10 400 python bench.py --cpu-images 0 --alt-precision none --alt-injected 0 --measure-traffic 0 --dump-conv $OUT/conv_layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05f/bench.json")); print(d["value"], d["roofline"]["frac"])
for a in d["alt_configs"]: print({k:v for k,v in a.items() if k in ("value","ms_per_step","detect_over_predict","split_ms","conv_ms_per_step","conv_frac_of_per_launch_roofline","predict_only_on_the_same_molded_batch")})
rows=json.load(open("gpurun_out/r05f/conv_layers.json"))
for r in rows:
    if r["K"]>=1024 and r["kernel"]=="direct": print(r["i"],r["M"],r["N"],r["K"],r["ms"],r["executed_tflops"])
PY
