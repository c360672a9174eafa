"""Where does the staged crop kernel differ from the gather kernel? (debugging aid)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskrcnn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def run(img, boxes, ind, ch, cw, tag, cpw=None):
    os.environ["MRCNN_CROP_STAGED"] = "0"
    ref = ops.crop(img, boxes, ind, 0.25, ch, cw)
    os.environ["MRCNN_CROP_STAGED"] = "1"
    if cpw:
        os.environ["MRCNN_CROP_CPW"] = str(cpw)
    else:
        os.environ.pop("MRCNN_CROP_CPW", None)
    for rep in range(3):
        got = ops.crop(img, boxes, ind, 0.25, ch, cw)
        bad = (got.view(torch.int32) != ref.view(torch.int32))
        nb = int(bad.sum())
        print(f"{tag} cpw={cpw} rep {rep}: {nb} of {bad.numel()} differ")
        if nb:
            idx = bad.nonzero().cpu().numpy()
            boxes_bad = np.unique(idx[:, 0])
            chans_bad = np.unique(idx[:, 1])
            print("  boxes:", boxes_bad[:20], "n", len(boxes_bad))
            print("  channels:", chans_bad[:40], "n", len(chans_bad))
            print("  positions y:", np.unique(idx[:, 2])[:30], " x:", np.unique(idx[:, 3])[:30])
            b0 = boxes_bad[0]
            H, W = img.shape[2:]
            bx = boxes[b0].cpu().numpy()
            print("  first bad box", b0, bx, "pixel rows", bx[0] * (H - 1), bx[2] * (H - 1), "cols", bx[1] * (W - 1), bx[3] * (W - 1))
            sub = idx[idx[:, 0] == b0]
            print("  its bad channels:", np.unique(sub[:, 1])[:40])
            c0 = sub[0, 1]
            print("  got", got[b0, c0].flatten()[:16].cpu().numpy())
            print("  ref", ref[b0, c0].flatten()[:16].cpu().numpy())
            print("  bad positions in (box,chan):", [(int(a), int(b)) for a, b in sub[sub[:, 1] == c0][:, 2:4][:20]])
            break


g = torch.Generator().manual_seed(21)
for (b, c, h, w, n, ch, cw) in [(1, 256, 64, 64, 50, 7, 7), (2, 33, 37, 19, 40, 14, 14), (3, 8, 128, 96, 64, 28, 28),
                                (1, 4, 9, 9, 30, 1, 1), (1, 300, 16, 16, 10, 3, 5), (2, 16, 32, 32, 33, 64, 2)]:
    img = torch.randn(b, c, h, w, generator=g)
    c2 = torch.rand(n, 2, generator=g)
    hw = torch.exp(torch.rand(n, 2, generator=g) * (np.log(0.9) - np.log(0.02)) + np.log(0.02))
    boxes = torch.cat([c2 - hw / 2, c2 + hw / 2], 1)
    ind = torch.randint(0, b, (n,), generator=g, dtype=torch.int32)
    if (ch, cw) == (64, 2):
        run(img.to(dev), boxes.to(dev), ind.to(dev), ch, cw, "case64x2")
g = torch.Generator().manual_seed(1234)
fm = torch.randn(1, 256, 256, 256, generator=g).to(dev)
c = torch.rand(256, 2, generator=g)
hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
ind = torch.zeros(256, dtype=torch.int32, device=dev)
for cpw in (16, 64):
    run(fm, boxes, ind, 14, 14, "config2", cpw)
