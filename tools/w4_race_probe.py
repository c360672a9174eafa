#!/usr/bin/env python3
"""Run-to-run reproducibility of conv3x3_wino4_f32 (plain variant) on the FPN P2 smoothing shape: N launches on the same
inputs, every output compared with the first launch's, bit for bit. Prints how many launches differed and the regions."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskrcnn_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 256, 256, 256, generator=g).to(dev)
xk = ops.nhwc_to_kblocked(x)
w = (torch.randn(256, 3, 3, 256, generator=g) * 0.02).to(dev)
u4 = ops.winograd4_weights(w)
shift = torch.randn(256, generator=g).to(dev)
for mode in ("both", "nhwc"):
    first = None
    bad = []
    for it in range(N):
        out = ops.conv3x3_winograd4(xk, u4, None, shift, False, None, mode)
        y = out[0] if mode == "both" else out
        if first is None:
            first = y.clone()
            ref = ops.conv3x3_winograd(xk, ops.winograd_weights(w), None, shift, False)
            print("first launch vs F(2x2) kernel: max |diff|", float((first - ref).abs().max()))
            continue
        if not torch.equal(y, first):
            d = (y != first)
            # geometry of the event: per (image, 16-row tile, 32-column tile, 64-channel N tile) the number of differing elements
            # (a whole tile is 32768; an epilogue round of 8 positions 8192) and the largest |difference|
            t = d.view(B, 16, 16, 8, 32, 4, 64).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, 16, 8, 4, -1).sum(-1)
            units = [(list(ix), int(t[tuple(ix)])) for ix in t.nonzero().tolist()]
            bad.append({"launch": it, "elements": int(d.sum()), "max_abs_diff": float((y - first).abs().max()), "tiles": units[:12]})
    print(json.dumps({"mode": mode, "batch": B, "launches": N, "differed": len(bad), "examples": bad[:5],
                      "flag": os.environ.get("W4_BUILD", "default")}), flush=True)
