#!/usr/bin/env python3
"""One ResNet C2 identity Bottleneck (model.py:190-211) at the benchmark's size (batch 8, 256 x 256 x 256 channels) in one of
three forms, for timing and for rocprofv3 counter passes (profiles/r03_bottleneck_traffic.json):
    native       ONE call of mrcnn_bottleneck_forward_f32 = conv1 + [conv2 + conv3 + residual fused]   (the default path)
    three        conv1 + F(4x4) conv2 + conv3-with-residual as three launches                          (MRCNN_FUSED_CONV3=0)
    whole_block  the one-launch whole-block kernel of csrc/bottleneck.hip                              (MRCNN_FUSED_BOTTLENECK=1)
usage: bottleneck_block.py <form> [--calls N] [--meta FILE]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from maskrcnn_amd import modules, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("form", choices=["native", "three", "whole_block"])
    ap.add_argument("--calls", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--meta", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = modules.synthetic_state_dict("resnet50")
    blk = modules.FusedBottleneck.from_state_dict(sd, "fpn.C2.1", 1, dev)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(args.batch, 256, 256, 256, generator=g).to(dev)
    modules.FUSED_BOTTLENECK = args.form == "whole_block"
    modules.FUSED_CONV3 = args.form != "three"
    fn = (lambda: blk(x)) if args.form != "three" else (lambda: blk.launch_by_launch(x))
    y = fn()
    torch.cuda.synchronize()
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            for _ in range(args.calls):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.calls
    m = args.batch * 256 * 256
    algo_bytes = 4 * (2 * m * 256)   # x read once, y written once: what a perfectly fused block moves
    meta = {"predict_calls": 2 + 2 * args.calls, "form": args.form, "shape": [args.batch, 256, 256, 256], "ms_per_block": round(ms, 4),
            "calls": 2 + 2 * args.calls, "block_min_bytes": algo_bytes, "checksum": float(y.double().sum().item()),
            "kernel_source_sha16": __import__("bench").kernel_source_sha16()}
    if args.meta:
        with open(args.meta, "w") as fh:
            json.dump(meta, fh)
    print(json.dumps(meta), flush=True)


if __name__ == "__main__":
    main()
