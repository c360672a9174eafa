"""Tuning aid: time the Winograd kernel alone (k-blocked input) on the P2-level 256 -> 256 layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = ops.nhwc_to_kblocked(torch.randn(8, 256, 256, 256, generator=g).to(dev))
wt = (torch.randn(256, 3, 3, 256, generator=g) * 0.02).to(dev)
u = ops.winograd_weights(wt)
sh = torch.zeros(256, device=dev)
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
t = timeit(lambda: ops.conv3x3_winograd(x, u, None, sh, relu=True))
print(round(t, 3), "ms; MFMA-ideal 1.966 ms ->", round(1.966 / t, 3))
