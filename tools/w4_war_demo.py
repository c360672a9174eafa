#!/usr/bin/env python3
"""The cause of round 5's rare wrong tiles of conv3x3_wino4_f32, made deterministic. Needs an ablation build:
    python maskrcnn_amd/build.py --variant w4_abl --only conv_wino4.hip -DMRCNN_W4_ABLATIONS ; MRCNN_LIB=<that library>
MRCNN_W4_DEBUG=4096: wave 0 of every workgroup sleeps ~8 000 cycles behind the prologue's staging barrier, the kernel otherwise as
shipped -> output bit-identical to the undelayed kernel. 12288: the same delay WITHOUT the barrier behind the prologue's operand
reads (= round 5's kernel) -> wave 0 multiplies k tile 0 with the U(2) pieces / raw k tile 2 pixels the other waves have staged
over buffer 0 meanwhile: every tile wrong; a few are saved for tools/w4_forensics.py analyze (the same decomposition as the
captured rare events: wave 0's quadrant, k tile 0, U of k tile 2)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import w4_forensics as f
from maskrcnn_amd import ops, _lib
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/w4_war_demo.npz"
dev = torch.device("cuda:0")
x, w, shift = f.operands()
xk, u4, shift = ops.nhwc_to_kblocked(x.to(dev)), ops.winograd4_weights(w.to(dev)), shift.to(dev)
def run(dbg):
    os.environ["MRCNN_W4_DEBUG"] = str(dbg)
    y = ops.conv3x3_winograd4(xk, u4, None, shift, False, None, "nhwc")
    torch.cuda.synchronize()
    return y
ref = run(0)
delayed = run(4096)
broken = run(12288)
t = (broken != ref).view(f.B, 16, 16, 8, 32, 4, 64).permute(0, 1, 3, 5, 2, 4, 6).reshape(f.B, 16, 8, 4, -1).sum(-1)
units = [(list(ix), int(t[tuple(ix)])) for ix in t.nonzero().tolist()]
print(json.dumps({"lib": _lib.LIB_PATH, "delayed_wave0_with_barrier_equals_product": bool(torch.equal(delayed, ref)),
                  "delayed_wave0_without_barrier_equals_product": bool(torch.equal(broken, ref)),
                  "tiles_wrong_without_barrier": len(units), "tiles": f.B * 16 * 8 * 4,
                  "max_abs_diff": float((broken - ref).abs().max())}))
ids, bad, first = [], [], []
for (b, ty, tx, nt), cnt in units[:6]:
    sl = (b, slice(16 * ty, 16 * ty + 16), slice(32 * tx, 32 * tx + 32), slice(64 * nt, 64 * nt + 64))
    ids.append([0, b, ty, tx, nt, cnt]); bad.append(broken[sl].cpu().numpy()); first.append(ref[sl].cpu().numpy())
np.savez_compressed(out, tile_ids=np.array(ids, dtype=np.int64).reshape(-1, 6), bad=np.array(bad, dtype=np.float32).reshape(-1, 16, 32, 64),
                    badk=np.array(bad, dtype=np.float32).reshape(-1, 16, 32, 64), first=np.array(first, dtype=np.float32).reshape(-1, 16, 32, 64),
                    events=json.dumps([]), launches=1, lib=_lib.LIB_PATH)
