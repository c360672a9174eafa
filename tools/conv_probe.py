#!/usr/bin/env python3
"""Tuning aid: time one conv shape on random vs zero data (DVFS check) — not part of the product."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskrcnn_amd import ops

def run(x, w, iters=20, **kw):
    for _ in range(3):
        ops.conv_bn_act(x, w, None, None, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv_bn_act(x, w, None, None, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

dev = torch.device("cuda:0")
for (b, h, wd, cin, cout, k) in [(8, 256, 256, 256, 512, 3), (8, 64, 64, 256, 256, 3), (8, 256, 256, 64, 256, 1)]:
    pad = (1, 1, 1, 1) if k == 3 else (0, 0, 0, 0)
    flops = 2.0 * b * h * wd * cin * cout * k * k
    for name, fill in (("random", None), ("zeros", 0.0)):
        x = torch.randn(b, h, wd, cin, device=dev) if fill is None else torch.zeros(b, h, wd, cin, device=dev)
        w = torch.randn(cout, k, k, cin, device=dev) * 0.05 if fill is None else torch.zeros(cout, k, k, cin, device=dev)
        ms = run(x, w, pad=pad)
        print(f"M={b*h*wd} N={cout} K={cin*k*k} {name}: {ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s", flush=True)
        w_hi, w_lo = ops.split_f16(w)
        for products in (3, 1):
            fn = lambda: ops.conv_bn_act_f16mfma(x, w_hi, w_lo, None, None, 1, pad, False, None, 1, products)
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"    f16mfma products={products}: {ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s (algorithmic)", flush=True)
