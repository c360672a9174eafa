#!/usr/bin/env python3
"""Tuning aid: the streaming 1x1 kernel (conv.hip: conv_pw_stream_f32) against the tiled kernel on ResNet C2's layer shapes
(batch 8, 256 x 256): time per call (hipGraph replay) and bitwise equality. MRCNN_CONV_NO_STREAM=1 times the tiled kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
res = {}
for name, cin, cout, relu, kb in (("c2_conv1", 256, 64, True, True), ("c2_conv1_b1", 64, 64, True, True), ("c2_ds", 64, 256, False, False)):
    x = torch.randn(8, 256, 256, cin, generator=g).to(dev)
    wt = (torch.randn(cout, 1, 1, cin, generator=g) * 0.05).to(dev)
    sc = (torch.rand(cout, generator=g) + 0.5).to(dev)
    sh = torch.randn(cout, generator=g).to(dev)
    fn = lambda: ops.conv_bn_act(x, wt, sc, sh, relu=relu, out_kblocked=kb)
    y = fn()
    # the tiled kernel: M below the streaming kernel's threshold (one image at a time)
    parts = [ops.conv_bn_act(x[i:i + 1].contiguous(), wt, sc, sh, relu=relu, out_kblocked=kb) for i in range(8)]
    ref = torch.cat(parts, dim=1 if kb else 0)
    same = bool(torch.equal(y, ref))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            for _ in range(20):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 60
    nbytes = (x.numel() + y.numel()) * 4
    res[name] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6), "bit_identical_to_tiled": same,
                 "max_abs_diff": float((y - ref).abs().max())}
print(json.dumps({"no_stream": os.environ.get("MRCNN_CONV_NO_STREAM", "0"), "layers": res}), flush=True)
