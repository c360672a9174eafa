import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from maskrcnn_amd import ops
from oracle import oracle
oracle.build()
rng = np.random.default_rng(0)
for (h, w, oh, ow) in ((7, 9, 28, 28), (7, 9, 28, 30), (16, 16, 32, 32), (16, 16, 30, 32), (7, 9, 7, 28), (7, 9, 28, 9), (40, 64, 75, 120)):
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    got = ops.resize_bilinear_u8(torch.from_numpy(a).cuda(), oh, ow).cpu().numpy()
    want = oracle.pil_resize_u8(a, oh, ow)
    bad = np.argwhere(got != want)
    print((h, w, oh, ow), "mismatches", len(bad), "cols", sorted(set(bad[:, 1].tolist()))[:16], "rows", sorted(set(bad[:, 0].tolist()))[:8])
