#!/bin/bash
# round 6, call B: the WAR race made deterministic (ablation build), then the fixed product kernel: Winograd tests, soak, bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06b; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/maskrcnn_amd/csrc/build/variants
MRCNN_LIB=$V/w4_abl/libmaskrcnn_hip.so timeout -k 10 200 python3 tools/w4_war_demo.py $OUT/war_demo.npz 2>$OUT/war_demo.err | tee $OUT/war_demo.json
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "winograd4 or wino4" > $OUT/tests_w4.log 2>&1; tail -2 $OUT/tests_w4.log
timeout -k 10 700 python3 tools/soak_probe.py ${1:-150000} wino4_plain_both,wino4_plain_relu_kblocked,wino4_heads,wino4_conv3 2>$OUT/soak.err | tee $OUT/soak_fixed.jsonl | cut -c1-300
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2>$OUT/bench.err; cut -c1-400 $OUT/bench.json
