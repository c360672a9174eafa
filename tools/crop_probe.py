"""crop_forward_nchw on the BASELINE configs[1] shapes (256 RoIs x 256 ch x 14x14 on P2..P5), staged kernel vs gather kernel
(MRCNN_CROP_STAGED), channels per wave (MRCNN_CROP_CPW). Usage: python tools/crop_probe.py [cpw ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskrcnn_amd import ops  # noqa: E402


def timeit(fn, iters=20, reps=5):
    """Device time per call: the calls are captured in a hipGraph (the Python + ctypes cost of a call, ~20 us, would
    otherwise bound a 20 us kernel) and the graph is replayed."""
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
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    dev = torch.device("cuda:0")
    cpws = [int(a) for a in sys.argv[1:]] or [0]
    levels = [int(v) for v in os.environ.get("CROP_LEVELS", "256,128,64,32").split(",")]
    g = torch.Generator().manual_seed(1234)
    for hl in levels:
        fm = torch.randn(1, 256, hl, hl, generator=g).to(dev)
        c = torch.rand(256, 2, generator=g)
        hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
        boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
        ind = torch.zeros(256, dtype=torch.int32, device=dev)
        algo = 256 * 256 * 14 * 14 * 4 + fm.numel() * 4 + 256 * 20
        row = {"level_hw": hl, "algorithmic_MB": round(algo / 1e6, 2)}
        os.environ["MRCNN_CROP_STAGED"] = "0"
        us = timeit(lambda: ops.crop(fm, boxes, ind, 0.0, 14, 14))
        ref = ops.crop(fm, boxes, ind, 0.0, 14, 14)
        row["gather_us"] = round(us, 2)
        os.environ["MRCNN_CROP_STAGED"] = "1"
        for cpw in cpws:
            if cpw:
                os.environ["MRCNN_CROP_CPW"] = str(cpw)
            else:
                os.environ.pop("MRCNN_CROP_CPW", None)
            us = timeit(lambda: ops.crop(fm, boxes, ind, 0.0, 14, 14))
            got = ops.crop(fm, boxes, ind, 0.0, 14, 14)
            row[f"staged_cpw{cpw}_us"] = round(us, 2)
            row[f"staged_cpw{cpw}_frac_of_8TBps"] = round(algo / us / 1e3 / 8000.0, 4)
            row[f"staged_cpw{cpw}_equal"] = bool(torch.equal(got.view(torch.int32), ref.view(torch.int32)))
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
