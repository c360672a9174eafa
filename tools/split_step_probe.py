#!/usr/bin/env python3
"""One step of the benchmark workload as SUB-BATCHES on concurrent HIP streams (8 = 4 + 4, 2 + 2 + 2 + 2 ...): does the tail of one
sub-batch's kernels fill with the other's? Results are identical either way (image i of a batch == image i alone). Prints images/s
of the whole batch per arrangement, same timing discipline as bench.py (K steps between synchronisations)."""
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

arch, H, W, batch, prec = "resnet50", 1024, 1024, 8, "f32"
if len(sys.argv) > 1 and sys.argv[1] == "c5":
    arch, H, W, prec = "resnet101", 832, 1344, "f16"
steps = 20
dev = torch.device("cuda:0")
cfg = InferenceConfig(image_height=H, image_width=W, backbone=arch, pre_nms_limit=1000, proposal_count=1000)
sd = modules.synthetic_state_dict(arch, seed=0, bn_seed=1)
mean = torch.tensor(cfg.mean_pixel)
g = torch.Generator().manual_seed(0)
images = (torch.randint(0, 256, (batch, H, W, 3), generator=g).float() - mean).permute(0, 3, 1, 2).contiguous().to(dev)
windows = torch.tensor([[0.0, 0.0, float(H), float(W)]] * batch, device=dev)
gc = torch.Generator().manual_seed(999)
cal = (torch.randint(0, 256, (batch, H, W, 3), generator=gc).float() - mean).permute(0, 3, 1, 2).contiguous()
net = bench.calibrate_heads_(sd, lambda s: MaskRCNNInference(s, cfg, dev, precision=prec), cal.to(dev), windows)


def step(parts, streams):
    if parts == 1:
        return [net.predict(images, windows, with_masks=True)]
    n = batch // parts
    cur = torch.cuda.current_stream()
    outs = []
    for i in range(parts):
        s = streams[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(net.predict(images[i * n:(i + 1) * n], windows[i * n:(i + 1) * n], with_masks=True))
    for s in streams[:parts]:
        cur.wait_stream(s)
    return outs


streams = [torch.cuda.Stream() for _ in range(4)]
ref = step(1, streams)[0]
if len(sys.argv) > 2 and sys.argv[2] == "soak":
    # the pipeline's own concurrent path (MaskRCNNInference._predict_concurrent), N steps, every result compared with one whole batch
    n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    net.sub_batches = 2
    bad = 0
    for i in range(n_steps):
        d = net.predict(images, windows, with_masks=True)
        if i % 7 == 0:
            torch.empty(64 << 20, dtype=torch.uint8, device=dev).fill_(i & 255)   # churn the allocator between steps
        if not (torch.equal(d.packed(), ref.packed()) and torch.equal(d.masks, ref.masks) and torch.equal(d.counts, ref.counts)):
            bad += 1
    torch.cuda.synchronize()
    print(json.dumps({"workload": f"{arch} {H}x{W} {prec}", "concurrent_sub_batches": 2, "steps": n_steps, "steps_that_differ_from_one_batch": bad}))
    sys.exit(0)
for parts in (1, 2, 4, 1, 2):
    for _ in range(5):
        step(parts, streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        outs = step(parts, streams)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    same = torch.equal(torch.cat([o.packed() for o in outs]), ref.packed()) and torch.equal(torch.cat([o.masks for o in outs]), ref.masks)
    print(json.dumps({"workload": f"{arch} {H}x{W} {prec}", "sub_batches": parts, "images_per_s": round(batch * steps / el, 2),
                      "ms_per_step": round(el / steps * 1e3, 3), "identical_to_one_batch": bool(same)}), flush=True)
