#!/usr/bin/env python3
"""Tuning aid (round 4): the 1x1 layers of the headline step on the tiled direct kernel (MRCNN_CONV_PIPE=0) against the
software-pipelined persistent kernel conv_pw_pipe_f32 in its tile shapes / drain spreads (MRCNN_CONV_PIPE values, see
conv.hip::run_conv_f32) — device time per call (hipGraph replay of 10 calls) and bit-equality of the outputs. One JSON line per layer."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from maskrcnn_amd import ops  # noqa: E402


def timeit(fn, iters=10, reps=3):
    for _ in range(2):
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
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


# (name, B, H, W, Cin, Cout, stride, relu, residual: 0 none / 1 same size / 2 half size, k-blocked output, k-blocked residual)
LAYERS = [
    ("C2 downsample 64->256", 8, 256, 256, 64, 256, 1, 0, 0, 0, 0),
    ("C3 conv3 128->512 +res", 8, 128, 128, 128, 512, 1, 1, 1, 0, 0),
    ("C3 downsample 256->512 s2", 8, 256, 256, 256, 512, 2, 0, 0, 0, 0),
    ("C3 conv1 512->128 (k-blocked out)", 8, 128, 128, 512, 128, 1, 1, 0, 1, 0),
    ("C4 conv3 256->1024 +res", 8, 64, 64, 256, 1024, 1, 1, 1, 0, 0),
    ("C4 downsample 512->1024 s2", 8, 128, 128, 512, 1024, 2, 0, 0, 0, 0),
    ("C4 conv1 1024->256 (k-blocked out)", 8, 64, 64, 1024, 256, 1, 1, 0, 1, 0),
    ("C5 conv3 512->2048 +res", 8, 32, 32, 512, 2048, 1, 1, 1, 0, 0),
    ("C5 conv1 2048->512", 8, 32, 32, 2048, 512, 1, 1, 0, 1, 0),
    ("P2 lateral 256->256 + half-size res (k-blocked)", 8, 256, 256, 256, 256, 1, 0, 2, 1, 1),
    ("P3 lateral 512->256 + half-size res (k-blocked)", 8, 128, 128, 512, 256, 1, 0, 2, 1, 1),
    ("P4 lateral 1024->256 + half-size res (k-blocked)", 8, 64, 64, 1024, 256, 1, 0, 2, 1, 1),
    ("classifier fc2 1024->1024", 1, 80, 100, 1024, 1024, 1, 1, 0, 0, 0),
]


def main():
    dev = torch.device("cuda:0")
    # "mode" or "mode:workgroups per CU" (MRCNN_PIPE_WGS)
    modes = sys.argv[1:] or ["0", "3", "5", "7", "2", "6", "6:3", "6:4"]
    g = torch.Generator().manual_seed(3)
    for (name, b, h, w, cin, cout, stride, relu, res, kb_out, kb_res) in LAYERS:
        x = torch.randn(b, h, w, cin, generator=g).to(dev)
        wt = (torch.randn(cout, 1, 1, cin, generator=g) * (2.0 / cin) ** 0.5).to(dev)
        scale = (torch.rand(cout, generator=g) + 0.5).to(dev)
        shift = (torch.randn(cout, generator=g) * 0.1).to(dev)
        oh, ow = (h + stride - 1) // stride, (w + stride - 1) // stride
        r = None
        if res:
            shape = (cout // 8, b, oh // res, ow // res, 8) if kb_res else (b, oh // res, ow // res, cout)
            r = torch.randn(*shape, generator=g).to(dev)
        fn = lambda: ops.conv_bn_act(x, wt, scale, shift, stride, (0, 0, 0, 0), relu, r, max(res, 1), out_kblocked=bool(kb_out))
        flops = 2.0 * b * oh * ow * cin * cout
        nbytes = 4.0 * (b * h * w * cin / (stride * stride) + b * oh * ow * cout * (1 + (1.0 / (res * res) if res else 0)) + cin * cout)
        row = {"layer": name, "M": b * oh * ow, "N": cout, "K": cin, "gflop": round(flops / 1e9, 2), "MB": round(nbytes / 1e6, 1)}
        ref = None
        for mode in modes:
            os.environ["MRCNN_CONV_PIPE"] = mode.split(":")[0]
            if ":" in mode:
                os.environ["MRCNN_PIPE_WGS"] = mode.split(":")[1]
            else:
                os.environ.pop("MRCNN_PIPE_WGS", None)
            y = fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = y
            ms = timeit(fn)
            row[f"mode{mode}_us"] = round(ms * 1e3, 1)
            row[f"mode{mode}_tflops"] = round(flops / ms / 1e9, 1)
            row[f"mode{mode}_equal"] = bool(torch.equal(y.view(torch.int32), ref.view(torch.int32)))
        os.environ.pop("MRCNN_CONV_PIPE", None)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
