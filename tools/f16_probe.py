#!/usr/bin/env python3
"""Tuning aid: time the fp16-storage conv kernel (precision="f16") on the config-5 layer shapes — not part of the product.
Tile choice comes from the environment (read once per process): run once per setting and compare the lines."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from maskrcnn_amd import ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# (name, B, H, W, Cin, Cout, k, residual)
SHAPES = [
    ("rpn_P2 3x3 256->512", 8, 208, 336, 256, 512, 3, False),
    ("fpn_P2 3x3 256->256", 8, 208, 336, 256, 256, 3, False),
    ("rpn_P3 3x3 256->512", 8, 104, 168, 256, 512, 3, False),
    ("C3 conv2 3x3 128->128", 8, 104, 168, 128, 128, 3, False),
    ("C4 conv2 3x3 256->256", 8, 52, 84, 256, 256, 3, False),
    ("C5 conv2 3x3 512->512", 8, 26, 42, 512, 512, 3, False),
    ("C4 conv1 1x1 1024->256", 8, 52, 84, 1024, 256, 1, False),
    ("C4 conv3 1x1 256->1024 +res", 8, 52, 84, 256, 1024, 1, True),
    ("C3 conv3 1x1 128->512 +res", 8, 104, 168, 128, 512, 1, True),
    ("C2 conv3 1x1 64->256 +res", 8, 208, 336, 64, 256, 1, True),
    ("fc1 7x7 as 1x1 12544->1024", 1, 80, 100, 12544, 1024, 1, False),
    ("C3 conv1 1x1 512->128", 8, 104, 168, 512, 128, 1, False),
    ("C2 conv2 3x3 64->64", 8, 208, 336, 64, 64, 3, False),
    ("C2 conv1 1x1 256->64", 8, 208, 336, 256, 64, 1, False),
]


def main():
    dev = torch.device("cuda:0")
    tag = os.environ.get("MRCNN_F16_BIG", "0")
    ref_dir = os.environ.get("F16_PROBE_REF")
    tot = 0.0
    for name, b, h, w, cin, cout, k, res in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(sum(map(ord, name)))
        x = torch.randn(b, h, w, cin, generator=g).to(dev).to(torch.float16)
        wt = (torch.randn(cout, k, k, cin, generator=g) * (1.0 / (k * k * cin) ** 0.5)).to(dev).to(torch.float16)
        r = torch.randn(b, h, w, cout, generator=g).to(dev).to(torch.float16) if res else None
        pad = (1, 1, 1, 1) if k == 3 else (0, 0, 0, 0)
        fn = lambda: ops.conv_bn_act_f16mfma(x, wt, None, None, None, 1, pad, True, r, 1, 1, None, True)
        y = fn()
        torch.cuda.synchronize()
        if ref_dir:
            path = os.path.join(ref_dir, name.replace(" ", "_").replace(">", "").replace("+", "") + ".pt")
            if tag == "0":
                os.makedirs(ref_dir, exist_ok=True)
                torch.save(y.cpu(), path)
                same = "ref"
            elif os.path.exists(path):
                same = "bitwise==ref" if torch.equal(torch.load(path), y.cpu()) else "DIFFERS"
            else:
                same = "-"
        else:
            same = ""
        ms = timeit(fn)
        tot += ms
        fl = 2.0 * b * h * w * cin * cout * k * k
        print(f"big={tag} {name:30s} M={b*h*w:7d} N={cout:5d} K={cin*k*k:6d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  {same}",
              flush=True)
        if os.environ.get("F16_PROBE_P8") and ops.conv_f16_pipelined_supported(b, h, w, cin, cout, k, k, pad):
            for tile in os.environ["F16_PROBE_P8"].split(","):
                rows, cols = [int(v) for v in tile.split("x")]
                if cols and cout % cols:
                    continue
                fp = lambda: ops.conv_f16_pipelined(x, wt, None, None, pad, True, r, True, False, rows, tile_cols=cols)
                yp = fp()
                torch.cuda.synchronize()
                same = "bitwise ==" if torch.equal(yp, y) else "DIFFERS from"
                msp = timeit(fp)
                print(f"      pipelined {rows:3d}x{cols:3d}: {msp*1e3:8.1f} us  {fl/msp/1e9:7.1f} TFLOP/s  {same} the 128x128 kernel",
                      flush=True)
    print(f"big={tag} total {tot:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
