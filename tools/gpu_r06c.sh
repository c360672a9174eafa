#!/bin/bash
# round 6, call C: schedule fuzzing — every kernel-level GPU test on a build in which each wave sleeps a pseudo-random 0..30 000
# cycles behind every workgroup barrier (maskrcnn_amd/build.py --variant sync_fuzz -DMRCNN_SYNC_FUZZ -DMRCNN_W4_ABLATIONS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06c; mkdir -p $OUT
export MRCNN_LIB=$GRAFT_REPO_ROOT/maskrcnn_amd/csrc/build/variants/sync_fuzz/libmaskrcnn_hip.so
# does the fuzz find round 5's kernel (no barrier behind the prologue's reads, no artificial delay)?
[ -n "$SKIP_DEMO" ] || timeout -k 10 200 python3 - > $OUT/fuzz_finds_r05.json 2>$OUT/fuzz_finds_r05.err <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import w4_forensics as f
from maskrcnn_amd import ops
dev = torch.device("cuda:0")
x, w, shift = f.operands()
xk, u4, shift = ops.nhwc_to_kblocked(x.to(dev)), ops.winograd4_weights(w.to(dev)), shift.to(dev)
def run(dbg):
    os.environ["MRCNN_W4_DEBUG"] = str(dbg)
    y = ops.conv3x3_winograd4(xk, u4, None, shift, False, None, "nhwc"); torch.cuda.synchronize(); return y
ref = ops.conv3x3_winograd(xk, ops.winograd_weights(w.to(dev)), None, shift, False)
res = {}
for name, dbg in (("shipped_kernel_fuzzed", 0), ("round5_kernel_fuzzed(no barrier behind the prologue's reads)", 8192)):
    ys = [run(dbg) for _ in range(5)]
    res[name] = {"max_abs_vs_F2x2": [round(float((y - ref).abs().max()), 6) for y in ys],
                 "launches_equal_to_first": sum(bool(torch.equal(y, ys[0])) for y in ys)}
print(json.dumps(res))
PY
cat $OUT/fuzz_finds_r05.json; tail -2 $OUT/fuzz_finds_r05.err
unset MRCNN_W4_DEBUG
for t in test_gpu_conv test_gpu_ops test_gpu_image; do
  S=$SECONDS; timeout -k 10 900 python3 -m pytest tests/$t.py -q -m gpu > $OUT/fuzz_$t.log 2>&1; echo "$t $((SECONDS-S)) s"; tail -4 $OUT/fuzz_$t.log
done
