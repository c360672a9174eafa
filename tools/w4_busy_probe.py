#!/usr/bin/env python3
"""One driver for counter passes over the three instantiations of conv3x3_wino4_f32 on their in-step shapes (batch 8):
plain = FPN P2 smoothing (256 x 256, 256 -> 256, no activation), heads = the RPN's shared conv + both 1x1 heads on P2
(256 -> 512), conv3 = a C2 bottleneck's conv2 + conv3 (64 -> 64 -> 256, residual). tools/w4_busy_pmc.sh wraps it in rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
B = 8
x256 = ops.nhwc_to_kblocked(torch.randn(B, 256, 256, 256, generator=g).to(dev))
u256 = ops.winograd4_weights((torch.randn(256, 3, 3, 256, generator=g) * 0.02).to(dev))
u512 = ops.winograd4_weights((torch.randn(512, 3, 3, 256, generator=g) * 0.02).to(dev))
sh256, sh512 = torch.randn(256, generator=g).to(dev), torch.randn(512, generator=g).to(dev)
w32 = torch.zeros(32, 512, device=dev); w32[:18] = torch.randn(18, 512, generator=g).to(dev) * 0.02
x64 = ops.nhwc_to_kblocked(torch.randn(B, 256, 256, 64, generator=g).to(dev))
u64 = ops.winograd4_weights((torch.randn(64, 3, 3, 64, generator=g) * 0.05).to(dev))
w3 = (torch.randn(256, 1, 1, 64, generator=g) * 0.1).to(dev)
res = torch.randn(B, 256, 256, 256, generator=g).to(dev)
s64, s256 = torch.randn(64, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
cases = {"plain": lambda: ops.conv3x3_winograd4(x256, u256, None, sh256, False, None, "nhwc"),
         "heads": lambda: ops.conv3x3_winograd4_heads(x256, u512, None, sh512, w32, True),
         "conv3": lambda: ops.conv3x3_winograd4_conv3(x64, u64, None, s64, w3, None, s256, res)}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, fn in cases.items():
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / n:.4f} ms", flush=True)
