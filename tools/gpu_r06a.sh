#!/bin/bash
# round 6, call A (provenance of profiles/r06_w4_root_cause.jsonl): (1) wrong tiles of the round-5 kernel WITHOUT its wait states
# captured whole (tools/w4_forensics.py). The variant "w4_nofix" was round 5's conv_wino4.hip (commit 7c66b25) built with
# -DMRCNN_W4_NO_RACE_FIX; that macro is gone with the pads — today the same kernel is MRCNN_W4_DEBUG=8192 of an ablation build
# (maskrcnn_amd/build.py --variant w4_abl --only conv_wino4.hip -DMRCNN_W4_ABLATIONS; tools/w4_war_demo.py);
# (2) the restructured kernel (no pads, dummy DMAs into the dump): Winograd tests, then the reproducibility soak
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06a; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/maskrcnn_amd/csrc/build/variants
MRCNN_LIB=$V/w4_nofix/libmaskrcnn_hip.so timeout -k 10 400 python3 tools/w4_forensics.py capture ${1:-120000} $OUT/capture_nofix.npz > $OUT/capture_nofix.log 2>&1
tail -2 $OUT/capture_nofix.log | cut -c1-300
timeout -k 10 300 python3 -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "winograd4 or wino4" > $OUT/tests_w4.log 2>&1; tail -3 $OUT/tests_w4.log
timeout -k 10 500 python3 tools/soak_probe.py ${2:-80000} wino4_plain_both,wino4_plain_relu_kblocked,wino4_heads,wino4_conv3 2>$OUT/soak.err | tee $OUT/soak_new.jsonl | cut -c1-300
