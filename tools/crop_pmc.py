"""A few launches of both NCHW crop kernels on the BASELINE configs[1] shape, for rocprofv3 --pmc passes
(FETCH_SIZE / WRITE_SIZE / TCP_TCC_READ_REQ_sum / TCC_HIT_sum TCC_MISS_sum / TA_BUSY...). Level: argv[1] (default 256); kernel:
argv[2] = MRCNN_CROP_STAGED (1 staged, the default route; 0 gather) — one kernel per process: the library reads its tuning
switches once."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hl = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ["MRCNN_CROP_STAGED"] = sys.argv[2] if len(sys.argv) > 2 else "1"
from maskrcnn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
fm = torch.randn(1, 256, hl, hl, generator=g).to(dev)
c = torch.rand(256, 2, generator=g)
hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
ind = torch.zeros(256, dtype=torch.int32, device=dev)
for _ in range(6):
    ops.crop(fm, boxes, ind, 0.0, 14, 14)
torch.cuda.synchronize()
