set -u
OUT=gpurun_out/r05m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MRCNN_CROP_BAND=0 timeout -k 10 200 python tools/crop_band_probe.py > $OUT/crop_band_off.jsonl 2>&1; cat $OUT/crop_band_off.jsonl
MRCNN_CROP_BAND=1 timeout -k 10 200 python tools/crop_band_probe.py > $OUT/crop_band_on.jsonl 2>&1; cat $OUT/crop_band_on.jsonl
MRCNN_CROP_BAND=1 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "crop or roi_align or cpu_tensors" > $OUT/tests_crop_band.log 2>&1; echo "crop tests (band) rc=$?"; tail -4 $OUT/tests_crop_band.log
timeout -k 10 600 python -m pytest tests/test_gpu_image.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "image or routing or agreement_rate" > $OUT/tests_misc.log 2>&1; echo "misc rc=$?"; tail -6 $OUT/tests_misc.log
timeout -k 10 300 python bench.py --cpu-images 0 --alt-precision none --alt-injected 0 --alt-config5 0 --measure-traffic 0 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05m/bench.json")); print(d["value"])
for a in d["alt_configs"]: print({k:v for k,v in a.items() if k in ("value","ms_per_step","detect_over_predict","split_ms","predict_only_on_the_same_molded_batch")})
PY
timeout -k 10 900 python tools/soak_probe.py 30000 wino2_linear_mask_head,f16p_c4_conv2,f16p_rpn_heads_p2,stem_pool_f16,stem_pool_f32,wino2_spatial,direct_1x1_256 > $OUT/soak_others.jsonl 2>&1; cat $OUT/soak_others.jsonl | cut -c1-220
