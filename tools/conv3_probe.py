"""Tuning aid: the C2 Bottleneck's conv2 + conv3 as two launches (F(4x4) Winograd, direct kernel) and as one (conv3 in the
Winograd kernel's epilogue). Batch 8, 256 x 256, 64 -> 64 -> 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
b, h, w = 8, 256, 256
xk = ops.nhwc_to_kblocked(torch.randn(b, h, w, 64, generator=g).to(dev))
u4 = ops.winograd4_weights((torch.randn(64, 3, 3, 64, generator=g) * 0.06).to(dev))
w3 = (torch.randn(256, 1, 1, 64, generator=g) * 0.17).to(dev)
s2 = t2 = torch.ones(64, device=dev); s3 = t3 = torch.ones(256, device=dev)
res = torch.randn(b, h, w, 256, generator=g).to(dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_c2 = timeit(lambda: ops.conv3x3_winograd4(xk, u4, s2, t2, True))
mid = ops.conv3x3_winograd4(xk, u4, s2, t2, True)
t_c3 = timeit(lambda: ops.conv_bn_act(mid, w3, s3, t3, relu=True, residual=res))
t_two = timeit(lambda: ops.conv_bn_act(ops.conv3x3_winograd4(xk, u4, s2, t2, True), w3, s3, t3, relu=True, residual=res))
t_f = timeit(lambda: ops.conv3x3_winograd4_conv3(xk, u4, s2, t2, w3, s3, t3, res))
print(f"conv2 {t_c2:.3f} ms | conv3 {t_c3:.3f} ms | two launches back to back {t_two:.3f} ms | fused {t_f:.3f} ms")
