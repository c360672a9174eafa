"""Profiling aid: run the Winograd kernel on one P2-level layer (8 x 256 x 256, 256 -> 256) a few times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
x = torch.randn(8, 256, 256, 256, generator=g).to(dev)
wt = (torch.randn(256, 3, 3, 256, generator=g) * 0.02).to(dev)
u = ops.winograd_weights(wt)
sh = torch.zeros(256, device=dev)
for _ in range(3):
    y = ops.conv3x3_winograd(x, u, None, sh, relu=True)
torch.cuda.synchronize()
print(float(y.abs().mean()))
