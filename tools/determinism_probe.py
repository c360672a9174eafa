#!/usr/bin/env python3
"""Is every stage of the trunk bit-reproducible run to run? Batch 2 at 1024^2 (the shape of tests/test_gpu_fullsize.py's fixture),
N iterations in one process; every stage's output is compared with the first iteration's, bit for bit. One line per stage that
ever differed (with the iteration, how many elements and where), or 'all stages reproducible'."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskrcnn_amd import modules, ops
from maskrcnn_amd.config import InferenceConfig
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
cfg = InferenceConfig(image_height=1024, image_width=1024, backbone="resnet50")
sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
g0 = torch.Generator().manual_seed(0)
images = torch.randint(0, 256, (2, 1024, 1024, 3), generator=g0).float() - torch.tensor(cfg.mean_pixel)
images = images.permute(0, 3, 1, 2).contiguous().to(dev)
bb = modules.FusedBackbone(sd, "resnet50", dev, precision="f32")

def stages(x):
    outs = {}
    st = bb.stem
    y = ops.stem_pool_f32(x.contiguous(), st.w.w, st.scale, st.shift, st.algo_cin)
    outs["stem"] = y
    for si, blocks in enumerate(bb.stages):
        for bi, blk in enumerate(blocks):
            y = blk(y)
            outs[f"C{si+2}.{bi}"] = y
    return outs

first, firstp, bad, nfail = None, None, {}, 0
junk = []
for it in range(N):
    # vary the allocator's state between iterations: what an uninitialised / out-of-range read would pick up
    junk = [torch.full((1 + (it * 7919) % 5, 1 << 20), float(it + 1), device=dev) for _ in range(1 + it % 3)]
    s = stages(images)
    p = bb(images)
    torch.cuda.synchronize()
    if first is None:
        first, firstp = {k: v.clone() for k, v in s.items()}, [v.clone() for v in p]
        continue
    for k, v in s.items():
        d = (v != first[k])
        if bool(d.any()) and k not in bad:
            idx = d.nonzero()
            bad[k] = (it, int(d.sum()), idx[0].tolist(), idx[-1].tolist(), float((v - first[k]).abs().max()))
    nfail += int(any(not torch.equal(v, w) for v, w in zip(p, firstp)) or any(not torch.equal(v, first[k]) for k, v in s.items()))
    for i, (v, w) in enumerate(zip(p, firstp)):
        d = (v != w)
        if bool(d.any()) and f"P{i+2}" not in bad:
            idx = d.nonzero()
            bad[f"P{i+2}"] = (it, int(d.sum()), idx[0].tolist(), idx[-1].tolist(), float((v - w).abs().max()))
for k, v in bad.items():
    print("DIFFERS", k, "iteration %d, %d elements, first %s last %s, max |diff| %.3e" % v)
print("all stages reproducible over %d iterations" % N if not bad else "%d stage(s) differed; %d of %d iterations had a difference" % (len(bad), nfail, N))
