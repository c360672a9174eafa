"""Tuning aid: time the Winograd kernel on the benchmark's big layers (k-blocked input, no layout pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator().manual_seed(0)
out = []
for (b, h, w, cin, cout) in ((8, 256, 256, 256, 512), (8, 256, 256, 256, 256), (8, 128, 128, 256, 512), (8, 64, 64, 256, 256), (8, 256, 256, 64, 64)):
    x = ops.nhwc_to_kblocked(torch.randn(b, h, w, cin, generator=g).to(dev))
    wt = (torch.randn(cout, 3, 3, cin, generator=g) * 0.02).to(dev)
    sh = torch.zeros(cout, device=dev)
    u = ops.winograd_weights(wt)
    t = timeit(lambda: ops.conv3x3_winograd(x, u, None, sh, relu=True))
    fl = 2.0 * b * h * w * cout * 9 * cin / 2.25
    out.append(f"{cin}->{cout}@{h}: {t:.3f} ms {fl/t/1e9:.1f} TF ({fl/t/1e9/157.3:.3f})")
print(" | ".join(out))
# the RPN layer with its two 1x1 heads fused, both tile shapes
out = []
for (b, h, w) in ((8, 256, 256), (8, 128, 128), (8, 64, 64)):
    x = ops.nhwc_to_kblocked(torch.randn(b, h, w, 256, generator=g).to(dev))
    u = ops.winograd_weights((torch.randn(512, 3, 3, 256, generator=g) * 0.02).to(dev))
    sh = torch.zeros(512, device=dev)
    w32 = torch.zeros(32, 512, device=dev); w32[:18] = torch.randn(18, 512, generator=g).to(dev) * 0.02
    for mode in (1, 2):
        t = timeit(lambda: ops.conv3x3_winograd_heads(x, u, None, sh, w32, True, tile_mode=mode))
        fl = 2.0 * b * h * w * 512 * 9 * 256 / 2.25
        out.append(f"heads@{h} mode{mode}: {t:.3f} ms ({fl/t/1e9/157.3:.3f})")
print(" | ".join(out))
# F(4x4,3x3) on the same big layers
out = []
for (b, h, w, cin, cout) in ((8, 256, 256, 256, 512), (8, 256, 256, 256, 256), (8, 128, 128, 256, 512), (8, 64, 64, 256, 256)):
    x = ops.nhwc_to_kblocked(torch.randn(b, h, w, cin, generator=g).to(dev))
    wt = (torch.randn(cout, 3, 3, cin, generator=g) * 0.02).to(dev)
    sh = torch.zeros(cout, device=dev)
    u4 = ops.winograd4_weights(wt)
    t = timeit(lambda: ops.conv3x3_winograd4(x, u4, None, sh, relu=True))
    fl = 2.0 * b * h * w * cout * 9 * cin / 4.0
    out.append(f"F4 {cin}->{cout}@{h}: {t:.3f} ms {fl/t/1e9:.1f} TF ({fl/t/1e9/157.3:.3f})")
print(" | ".join(out))
