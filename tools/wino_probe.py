"""Tuning aid: Winograd 3x3 kernel vs the direct implicit-GEMM kernel — accuracy against torch CPU (small) and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from maskrcnn_amd import ops
dev = "cuda:0"

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

g = torch.Generator().manual_seed(0)
for (b, h, w, cin, cout) in ((2, 14, 14, 64, 96), (1, 32, 48, 24, 40), (3, 16, 16, 256, 64)):
    x = torch.randn(b, h, w, cin, generator=g)
    wt = torch.randn(cout, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    ref = ref.relu().permute(0, 2, 3, 1)
    u = ops.winograd_weights(wt.to(dev).contiguous())
    y = ops.conv3x3_winograd(x.to(dev), u, sc.to(dev), sh.to(dev), relu=True).cpu()
    yd = ops.conv_bn_act(x.to(dev), wt.to(dev), sc.to(dev), sh.to(dev), 1, (1, 1, 1, 1), True).cpu()
    print(f"{(b,h,w,cin,cout)}: wino max|err| {float((y-ref).abs().max()):.2e}   direct {float((yd-ref).abs().max()):.2e}   max|ref| {float(ref.abs().max()):.2f}")

for (b, h, w, cin, cout) in ((8, 256, 256, 256, 512), (8, 256, 256, 256, 256), (8, 128, 128, 128, 128), (8, 64, 64, 256, 256),
                             (8, 32, 32, 512, 512), (400, 14, 14, 256, 256), (8, 256, 256, 64, 64)):
    x = torch.randn(b, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, 3, 3, cin, generator=g) * 0.02).to(dev)
    sh = torch.zeros(cout, device=dev)
    u = ops.winograd_weights(wt)
    tw = timeit(lambda: ops.conv3x3_winograd(x, u, None, sh, relu=True))
    td = timeit(lambda: ops.conv_bn_act(x, wt, None, sh, 1, (1, 1, 1, 1), True))
    fl = 2.0 * b * h * w * cout * 9 * cin
    print(f"{(b,h,w,cin,cout)}: wino {tw:7.3f} ms ({fl/tw/1e9:6.1f} TF algorithmic, {fl/2.25/tw/1e9:6.1f} TF executed)   direct {td:7.3f} ms ({fl/td/1e9:6.1f} TF)   speedup {td/tw:.2f}")
