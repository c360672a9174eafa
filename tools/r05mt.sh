set -u
OUT=gpurun_out/r05mt2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -m gpu -q -p no:cacheprovider -k "mask_tail or bottleneck_c2_f16" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.log
C5="--arch resnet101 --height 832 --width 1344 --precision f16 --cpu-images 0 --alt-precision none --alt-config5 0 --alt-injected 0 --alt-detect 0 --measure-traffic 0"
timeout -k 10 300 python bench.py $C5 --dump-conv $OUT/conv_on.json > $OUT/bench_on.json 2> $OUT/bench_on.err; echo "on rc=$?"
MRCNN_F16_FUSED_MASK_TAIL=0 timeout -k 10 300 python bench.py $C5 --dump-conv $OUT/conv_off.json > $OUT/bench_off.json 2> $OUT/bench_off.err; echo "off rc=$?"
python - <<'PY'
import json
for t in ("on","off"):
    d=json.load(open(f"gpurun_out/r05mt2/bench_{t}.json")); print(t, d["value"], d["ms_per_step"])
    L=json.load(open(f"gpurun_out/r05mt2/conv_{t}.json"))
    for l in L[-8:]: print("  ", l["i"], l["kernel"], l["M"], l["N"], l["K"], l["ms"], l["algorithmic_GBps"])
PY
