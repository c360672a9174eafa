#!/usr/bin/env python3
"""Tuning aid: is an fp32 MFMA kernel power-bound? The same launches on random and on all-zero operands (hipGraph replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"; g = torch.Generator().manual_seed(0)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(iters): fn()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters)
for fill in ("random", "zeros"):
    mk = (lambda *s: torch.randn(*s, generator=g).to(dev)) if fill == "random" else (lambda *s: torch.zeros(*s, device=dev))
    # F(4x4) RPN level on P2 with the heads fused
    x = ops.nhwc_to_kblocked(mk(8, 256, 256, 256)); u4 = ops.winograd4_weights(mk(512, 3, 3, 256) * 0.02)
    w32 = mk(32, 512) * 0.02; sh = torch.zeros(512, device=dev)
    t = timeit(lambda: ops.conv3x3_winograd4_heads(x, u4, None, sh, w32, True))
    print(fill, "wino4 heads P2", round(t, 4), "ms", flush=True)
    u4b = ops.winograd4_weights(mk(256, 3, 3, 256) * 0.02); shb = torch.zeros(256, device=dev)
    t = timeit(lambda: ops.conv3x3_winograd4(x, u4b, None, shb, relu=False))
    print(fill, "wino4 plain P2", round(t, 4), "ms", flush=True)
    del x, u4, u4b
    # direct kernel: classifier fc1 (K = 12544) and a C4 conv1 (K = 1024)
    a = mk(1, 8000, 1, 12544); wt = mk(1024, 1, 1, 12544) * 0.01
    t = timeit(lambda: ops.conv_bn_act(a, wt, None, None, relu=True))
    print(fill, "direct fc1", round(t, 4), "ms", flush=True)
    a = mk(8, 64, 64, 1024); wt = mk(256, 1, 1, 1024) * 0.03
    t = timeit(lambda: ops.conv_bn_act(a, wt, None, None, relu=True))
    print(fill, "direct c4 conv1", round(t, 4), "ms", flush=True)
    # F(2x2) mask-head conv
    xm = ops.nhwc_to_kblocked(mk(400, 14, 14, 256)); um = ops.winograd_weights(mk(256, 3, 3, 256) * 0.02)
    t = timeit(lambda: ops.conv3x3_winograd(xm, um, None, shb, True))
    print(fill, "wino2 mask head", round(t, 4), "ms", flush=True)
