set -u
OUT=gpurun_out/r05e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_image.py -m gpu -q -p no:cacheprovider -x > $OUT/tests_image.log 2>&1; echo "image rc=$?"; tail -3 $OUT/tests_image.log
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "agreement_rate or config5" > $OUT/tests_rate.log 2>&1; echo "rate rc=$?"; tail -6 $OUT/tests_rate.log
cp gpurun_out/parity_fullsize.json $OUT/parity_rate.json 2>/dev/null
timeout -k 10 300 python bench.py --cpu-images 0 --alt-precision none --alt-injected 0 --measure-traffic 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05e/bench.json")); print(d["value"])
for a in d["alt_configs"]: print({k:v for k,v in a.items() if k in ("value","ms_per_step","detect_over_predict","split_ms","conv_ms_per_step","conv_frac_of_per_launch_roofline","predict_only_on_the_same_molded_batch")})
z=json.load(open("gpurun_out/r05e/parity_rate.json")); print(z.get("config5/f16/detection_agreement_rate",{}).get("without_a_match_iou_ge_0.5"))
PY
timeout -k 10 600 python tools/f16_small_probe.py > $OUT/f16_small_probe.jsonl 2> $OUT/f16_small_probe.err; echo "probe rc=$?"
python - <<'PY'
import json, collections
rows=[json.loads(l) for l in open("gpurun_out/r05e/f16_small_probe.jsonl")]
by=collections.OrderedDict()
for r in rows:
    if "us" in r: by.setdefault(r["layer"],[]).append((r["us"], tuple(r["tile"])))
for k,v in by.items():
    auto=[u for u,t in v if t==(0,0)][0]; best=min(v)
    print(f"{k:22s} auto {auto:7.1f}  best {best[0]:7.1f} {best[1]}  all {sorted(v)[:4]}")
PY
