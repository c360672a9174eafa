#!/usr/bin/env python3
"""What would skipping / compacting the classifier's empty RoI rows buy? The K = 12544 GEMM (model.py:782-786 as one GEMM) timed
alone (hipGraph replay, best of 3) for M = all 8000 slots, the 7168 rows of the tiles that run today, and the ~6304 valid rows
compacted, on the tiles the library can give it. One JSON line per case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from maskrcnn_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(1024, 1, 1, 12544, generator=g) * 0.01).to(dev)
    sc, sh = torch.rand(1024, generator=g).to(dev) + 0.5, torch.randn(1024, generator=g).to(dev)

    def timeit(fn, iters=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(iters):
                fn()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
        return best

    for m in (8000, 7168, 6400, 6304, 6144, 5120, 4096):
        x = torch.randn(m, 1, 1, 12544, generator=g).to(dev)
        out = torch.empty(m, 1, 1, 1024, device=dev)
        ms = timeit(lambda: ops.conv_bn_act(x, w, sc, sh, relu=True, out=out))
        tiles = -(-m // 128) * 8
        print(json.dumps({"M": m, "tiles_128x128": tiles, "ms": round(ms, 4), "tflops": round(2.0 * m * 12544 * 1024 / ms / 1e9, 1),
                          "tile_env": os.environ.get("MRCNN_CONV_TILE", "")}), flush=True)
        del x, out


if __name__ == "__main__":
    main()
