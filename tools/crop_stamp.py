"""Tuning aid: in-kernel cycle stamps of the staged NCHW crop kernel (MRCNN_CROP_STAMPS=1 python maskrcnn_amd/build.py):
wave 0 of every box's first workgroup — start | samples + barrier | footprint scan | lane set-up | first group landed |
first group interpolated | end. Level: argv[1] (default 256); MRCNN_CROP_CPW as usual."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskrcnn_amd import ops  # noqa: E402

hl = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
fm = torch.randn(1, 256, hl, hl, generator=g).to(dev)
c = torch.rand(256, 2, generator=g)
hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
ind = torch.zeros(256, dtype=torch.int32, device=dev)
for _ in range(5):
    out = ops.crop(fm, boxes, ind, 0.0, 14, 14)
torch.cuda.synchronize()
st = out.view(256, -1)[:, :18].contiguous().view(torch.int64).cpu().numpy()   # [box][9]
d = np.diff(st[:, :7], axis=1)
names = ["samples+barrier", "scan", "lane setup", "dma0 landed", "group0 done", "rest"]
S = st[:, 8] >> 32
G = (st[:, 8] & 0xFFFFFFFF) >> 8
K = st[:, 8] & 0xFF
print("level", hl, "median cycles per phase:", {n: int(np.median(d[:, i])) for i, n in enumerate(names)},
      "total", int(np.median(st[:, 6] - st[:, 0])))
for lo, hi in ((0, 32), (32, 96), (96, 200), (200, 600)):
    m = (S >= lo) & (S < hi)
    if m.any():
        print(f"  S in [{lo},{hi}): n={int(m.sum())} G~{int(np.median(G[m]))} K~{int(np.median(K[m]))}",
              {n: int(np.median(d[m, i])) for i, n in enumerate(names)}, "total", int(np.median((st[:, 6] - st[:, 0])[m])))
span = st[:, 6].max() - st[:, 0].min()
print("  first start -> last end of these waves:", int(span), "cycles (100 MHz ticks if s_memtime is the constant clock)")
