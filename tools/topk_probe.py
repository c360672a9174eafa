"""Tuning aid: mrcnn_topk_desc_f32 vs torch.topk on 8 x 261888 scores (uniform and RPN-like clustered data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator().manual_seed(0)
for name, s in (("uniform", torch.rand(8, 261888, generator=g)),
                ("rpn-like", torch.sigmoid(torch.randn(8, 261888, generator=g) * 2 - 4)),
                ("saturated", torch.sigmoid(torch.randn(8, 261888, generator=g) * 30))):
    s = s.to(dev)
    for k in (1000, 500):
        t0 = timeit(lambda: ops.topk_desc(s, k))
        t1 = timeit(lambda: s.topk(k, dim=1, sorted=True))
        print(f"{name:10s} k={k}: hip {t0:7.1f} us   torch.topk {t1:7.1f} us")
