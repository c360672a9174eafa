#!/usr/bin/env python3
"""Run-to-run reproducibility of the WHOLE step in its own context (the kernels before and after each other, the step's clocks):
the benchmark workload (configs[2]: R50-FPN, batch 8, 1024^2, 1000 proposals; `c5`: configs[4]'s geometry in the fp16 mode), N
steps; every step's feature maps, RPN outputs, detections and masks are compared bit for bit with the first step's."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from maskrcnn_amd import modules  # noqa: E402
from maskrcnn_amd.config import InferenceConfig  # noqa: E402
from maskrcnn_amd.pipeline import MaskRCNNInference  # noqa: E402

c5 = len(sys.argv) > 1 and sys.argv[1] == "c5"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
arch, H, W, prec = ("resnet101", 832, 1344, "f16") if c5 else ("resnet50", 1024, 1024, "f32")
batch = 8
dev = torch.device("cuda:0")
cfg = InferenceConfig(image_height=H, image_width=W, backbone=arch, pre_nms_limit=1000, proposal_count=1000)
sd = modules.synthetic_state_dict(arch, seed=0, bn_seed=1)
mean = torch.tensor(cfg.mean_pixel)
g = torch.Generator().manual_seed(0)
images = (torch.randint(0, 256, (batch, H, W, 3), generator=g).float() - mean).permute(0, 3, 1, 2).contiguous().to(dev)
windows = torch.tensor([[0.0, 0.0, float(H), float(W)]] * batch, device=dev)
gc = torch.Generator().manual_seed(999)
cal = (torch.randint(0, 256, (batch, H, W, 3), generator=gc).float() - mean).permute(0, 3, 1, 2).contiguous()
net = bench.calibrate_heads_(sd, lambda s: MaskRCNNInference(s, cfg, dev, precision=prec, concurrent_sub_batches=1), cal.to(dev), windows)


def snapshot():
    det, mid = net.predict(images, windows, with_masks=True, return_intermediates=True)
    t = {f"P{i + 2}": f for i, f in enumerate(mid["feature_maps"])}
    t.update(rpn_scores=mid["rpn_scores"], rpn_deltas=mid["rpn_deltas"], rois=mid["rois"], logits=mid["logits"],
             detections=det.packed(), masks=det.masks)
    return t


first = {k: v.clone() for k, v in snapshot().items()}
bad = {}
t0 = time.time()
for it in range(1, n_steps):
    cur = snapshot()
    for k, v in cur.items():
        if not torch.equal(v, first[k]):
            e = bad.setdefault(k, {"steps": 0, "first_step": it, "max_abs_diff": 0.0})
            e["steps"] += 1
            e["max_abs_diff"] = max(e["max_abs_diff"], float((v.float() - first[k].float()).abs().max()))
torch.cuda.synchronize()
print(json.dumps({"workload": f"{arch} {H}x{W} {prec} batch {batch}", "steps": n_steps, "tensors_compared_per_step": len(first),
                  "tensors_that_ever_differed": bad, "seconds": round(time.time() - t0, 1)}))
