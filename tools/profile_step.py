#!/usr/bin/env python3
"""The benchmark's step, alone, for rocprofv3 counter passes (bench.py adds an alt-precision leg, the CPU baseline and
the per-op microbenchmarks, none of which belong in a PMC pass — counter collection serialises every dispatch).

    rocprofv3 --pmc SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o p -- \
        python3 tools/profile_step.py --steps 3 --meta DIR/meta.json

Same workload, weights, calibration and inputs as `bench.py` at N=1 (configs[2]: ResNet-50-FPN, batch 8, 1024^2, 1000
proposals). EVERY predict() of the run — the calibration passes too — has the workload's batch size, so each kernel's
launches all have the workload's shapes and per-step figures are totals / predict calls (written to --meta)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--proposals", type=int, default=1000)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--meta", default=None)
    args = ap.parse_args()
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    dev = torch.device("cuda:0")
    H, W = args.height, args.width
    cfg = InferenceConfig(image_height=H, image_width=W, backbone=args.arch, pre_nms_limit=args.proposals,
                          proposal_count=args.proposals)
    sd = modules.synthetic_state_dict(args.arch, seed=0, bn_seed=1)
    mean = torch.tensor(cfg.mean_pixel)
    g = torch.Generator().manual_seed(0)
    images = (torch.randint(0, 256, (args.batch, H, W, 3), generator=g).float() - mean).permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0.0, 0.0, float(H), float(W)]] * args.batch, device=dev)
    calls = [0]

    class Counting(MaskRCNNInference):
        def predict(self, images, windows, with_masks=True, **k):
            calls[0] += 1
            return super().predict(images, windows, with_masks=True, **k)   # every pass runs the mask head too

    make_net = lambda s: Counting(s, cfg, dev, precision=args.precision, concurrent_sub_batches=1)   # counters: whole-batch launches
    gc = torch.Generator().manual_seed(999)
    cal = (torch.randint(0, 256, (args.batch, H, W, 3), generator=gc).float() - mean).permute(0, 3, 1, 2).contiguous()
    net = bench.calibrate_heads_(sd, make_net, cal.to(dev), windows)
    for _ in range(args.steps):
        net.predict(images, windows, with_masks=True)
    torch.cuda.synchronize()
    meta = {"predict_calls": calls[0], "precision": args.precision, "batch": args.batch, "image": [H, W],
            "arch": args.arch, "proposals": args.proposals, "winograd": bool(modules.WINOGRAD),
            "stem_kernel": bool(modules.STEM_KERNEL),
            "fused_bottleneck": bool(getattr(modules, "FUSED_BOTTLENECK", False)),
            "rpn_fused_heads": bool(getattr(modules, "RPN_FUSED_HEADS", False)),
            "winograd4": bool(getattr(modules, "WINOGRAD4", False)),
            "winograd4_trunk": bool(getattr(modules, "WINOGRAD4_TRUNK", False)),
            # bench.py reports these byte counts only for a run of the SAME kernel sources
            "kernel_source_sha16": bench.kernel_source_sha16()}
    if args.meta:
        with open(args.meta, "w") as fh:
            json.dump(meta, fh)
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
