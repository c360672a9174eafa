#!/usr/bin/env python3
"""conv_f16p on the SMALL-M layers of configs[4] (C5 / P5 / P6 at 832 x 1344, batch 8: M = 8736 / 2184 rows; and the C4 layers,
M = 34944) for every legal tile: with 256-column tiles these launches are 34 - 138 workgroups on 256 CUs, each walking a long K
alone. One JSON line per (layer, tile): us per launch (hipGraph replay of 10, best of 3)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

LAYERS = [  # name, B, H, W, Cin, Cout, k, stride, residual
    ("C5.0 conv1 s2", 8, 52, 84, 1024, 512, 1, 2, False),
    ("C5.0 downsample s2", 8, 52, 84, 1024, 2048, 1, 2, False),
    ("C5 conv1", 8, 26, 42, 2048, 512, 1, 1, False),
    ("C5 conv2", 8, 26, 42, 512, 512, 3, 1, False),
    ("C5 conv3+res", 8, 26, 42, 512, 2048, 1, 1, True),
    ("P5 lateral", 8, 26, 42, 2048, 256, 1, 1, False),
    ("P5 smoothing", 8, 26, 42, 256, 256, 3, 1, False),
    ("RPN P5", 8, 26, 42, 256, 512, 3, 1, False),
    ("RPN P6", 8, 13, 21, 256, 512, 3, 1, False),
    ("C4 conv1", 8, 52, 84, 1024, 256, 1, 1, False),
    ("C4 conv2", 8, 52, 84, 256, 256, 3, 1, False),
    ("C4 conv3+res", 8, 52, 84, 256, 1024, 1, 1, True),
    ("P4 smoothing", 8, 52, 84, 256, 256, 3, 1, False),
    ("RPN P4", 8, 52, 84, 256, 512, 3, 1, False),
]
TILES = [(0, 0), (128, 256), (160, 256), (192, 256), (256, 256), (128, 128), (256, 128), (128, 64), (256, 64)]


def main():
    from maskrcnn_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)

    def timeit(fn, iters=10):
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
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        return best

    for name, b, h, w, cin, cout, k, stride, res in LAYERS:
        x = torch.randn(b, h, w, cin, generator=g).half().to(dev)
        wt = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).half().to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
        pad = (1, 1, 1, 1) if k == 3 else (0, 0, 0, 0)
        oh, ow = -(-h // stride), -(-w // stride)
        r = torch.randn(b, oh, ow, cout, generator=g).half().to(dev) if res else None
        for tr, tc in TILES:
            if tc and cout % tc:
                continue
            try:
                us = timeit(lambda: ops.conv_f16_pipelined(x, wt, sc, sh, pad, True, r, out_f16=True, tile_rows=tr, tile_cols=tc,
                                                           stride=stride))
            except Exception as e:
                print(json.dumps({"layer": name, "tile": [tr, tc], "error": str(e)[:100]}), flush=True)
                continue
            m = b * oh * ow
            print(json.dumps({"layer": name, "M": m, "N": cout, "K": k * k * cin, "tile": [tr, tc], "us": round(us, 2),
                              "tflops": round(2.0 * m * cout * k * k * cin / us / 1e6, 1)}), flush=True)
        del x, wt, r


if __name__ == "__main__":
    main()
