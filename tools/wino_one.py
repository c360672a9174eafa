"""Tuning aid: run the Winograd kernel a few times on the P2 RPN layer (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
b, h, w, cin, cout = 8, 256, 256, 256, 512
x = ops.nhwc_to_kblocked(torch.randn(b, h, w, cin, generator=g).to(dev))
wt = (torch.randn(cout, 3, 3, cin, generator=g) * 0.02).to(dev)
u = ops.winograd_weights(wt)
sh = torch.zeros(cout, device=dev)
for _ in range(4):
    ops.conv3x3_winograd(x, u, None, sh, relu=True)
torch.cuda.synchronize()
