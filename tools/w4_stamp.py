"""Tuning aid: in-kernel cycle stamps of the F(4x4) kernel (MRCNN_W4_ABLATIONS build, MRCNN_W4_DEBUG=2048): workgroup 0's
second tile — tile start, setup, prologue, k loop, epilogue round 0, epilogue end."""
import os, sys
os.environ["MRCNN_W4_DEBUG"] = "2048"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
b, h, w, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (8, 256, 256, 256, 512))]
x = ops.nhwc_to_kblocked(torch.randn(b, h, w, cin, generator=g).to(dev))
u4 = ops.winograd4_weights((torch.randn(cout, 3, 3, cin, generator=g) * 0.02).to(dev))
sh = torch.zeros(cout, device=dev)
for _ in range(3):
    y, yk = ops.conv3x3_winograd4(x, u4, None, sh, relu=True, out="both")
torch.cuda.synchronize()
st = yk.view(-1)[:24].view(torch.int64).cpu().tolist()
d = [st[i + 1] - st[i] for i in range(11)]
print("cycles: setup %d | prologue %d | k loop %d (%d per k tile)" % (d[0], d[1], d[2], d[2] // (cin // 4)))
print("  round 0: Z write %d | barrier %d | read+transform+stores %d | barrier %d" % tuple(d[3:7]))
print("  round 1: Z write %d | barrier %d | read+transform+stores %d | barrier %d" % tuple(d[7:11]))

# the heads variant: stamps land in the shift vector
w32 = torch.zeros(32, cout, device=dev); w32[:18] = torch.randn(18, cout, generator=g).to(dev) * 0.02
sh2 = torch.zeros(cout, device=dev)
for _ in range(3):
    sh2.zero_()
    ops.conv3x3_winograd4_heads(x, u4, None, sh2, w32, True)
torch.cuda.synchronize()
st = sh2[:32].view(torch.int64).cpu().tolist()
d = [st[i + 1] - st[i] for i in range(13)]
print("heads: setup %d | prologue %d | k loop %d (%d per k tile)" % (d[0], d[1], d[2], d[2] // (cin // 4)))
print("  round 0: Z write %d | barrier %d | read+transform+T %d | barrier %d | head MFMAs + sums %d" % tuple(d[3:8]))
print("  round 1: Z write %d | barrier %d | read+transform+T %d | barrier %d | head MFMAs + sums %d" % tuple(d[8:13]))
