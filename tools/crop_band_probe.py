#!/usr/bin/env python3
"""crop_forward_nchw on BASELINE configs[1] (256 RoIs x 256 ch x 14x14 on the P2..P5 map sizes of a 1024^2 image), hipGraph-timed
like bench.py's roofline_ops. Run once per kernel choice (the switch is read once per process):
    MRCNN_CROP_BAND=0 python tools/crop_band_probe.py      # box-stationary (staged) kernel
    MRCNN_CROP_BAND=1 python tools/crop_band_probe.py      # map-stationary (band) kernel"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from maskrcnn_amd import ops
    dev = torch.device("cuda:0")

    def timeit(fn, iters=20, reps=3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graph.replay(); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        return best

    g = torch.Generator().manual_seed(1234)
    c = torch.rand(256, 2, generator=g)
    hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
    boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
    ind = torch.zeros(256, dtype=torch.int32, device=dev)
    for hl in (256, 128, 64, 32):
        fm = torch.randn(1, 256, hl, hl, generator=g).to(dev)
        us = timeit(lambda: ops.crop(fm, boxes, ind, 0.0, 14, 14))
        algo = 256 * 256 * 14 * 14 * 4 + fm.numel() * 4 + 256 * 20
        print(json.dumps({"band": os.environ.get("MRCNN_CROP_BAND", ""), "map": hl, "us": round(us, 2),
                          "frac_of_8TBs": round(algo / us / 1e3 / 8000.0, 4)}), flush=True)
    # the mask head's call shape on a batch: 8 images x 50 boxes x 14x14 on P2 (box_index over the batch)
    fm = torch.randn(8, 256, 256, 256, generator=g).to(dev)
    c = torch.rand(400, 2, generator=g)
    hw = torch.exp(torch.rand(400, 2, generator=g) * 2.0 - 3.5)
    b8 = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
    i8 = (torch.arange(400) // 50).to(torch.int32).to(dev)
    us = timeit(lambda: ops.crop(fm, b8, i8, 0.0, 14, 14), iters=5)
    print(json.dumps({"band": os.environ.get("MRCNN_CROP_BAND", ""), "map": "8x256x256x256, 400 boxes", "us": round(us, 2)}), flush=True)


if __name__ == "__main__":
    main()
