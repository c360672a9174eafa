"""Tuning aid: time the F(4x4) Winograd kernel on the P2 RPN layer (MRCNN_W4_DEBUG ablations read by the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
b, h, w, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 256, 256, 256, 512))]
x = ops.nhwc_to_kblocked(torch.randn(b, h, w, cin, generator=g).to(dev))
u4 = ops.winograd4_weights((torch.randn(cout, 3, 3, cin, generator=g) * 0.02).to(dev))
sh = torch.zeros(cout, device=dev)
for _ in range(3): ops.conv3x3_winograd4(x, u4, None, sh, relu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.conv3x3_winograd4(x, u4, None, sh, relu=True)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
fl = 2.0 * b * h * w * cout * 9 * cin / 4.0
print(f"W4_DEBUG={os.environ.get('MRCNN_W4_DEBUG', '0')}: {t:.3f} ms ({fl/t/1e9/157.3:.3f})", flush=True)
# the same layer with the RPN heads fused
w32 = torch.zeros(32, cout, device=dev); w32[:18] = torch.randn(18, cout, generator=g).to(dev) * 0.02
for _ in range(3): ops.conv3x3_winograd4_heads(x, u4, None, sh, w32, True)
torch.cuda.synchronize()
e0.record()
for _ in range(10): ops.conv3x3_winograd4_heads(x, u4, None, sh, w32, True)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
print(f"  with heads: {t:.3f} ms ({fl/t/1e9/157.3:.3f})", flush=True)
