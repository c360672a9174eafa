#!/usr/bin/env python3
"""Times the whole-block fused Bottleneck against the three-launch path on the C2 identity-block shape of the
benchmark (batch 8, 256 x 256 x 256). HIP events on the launch stream."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from maskrcnn_amd import modules, ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    sd = modules.synthetic_state_dict("resnet50")
    blk = modules.FusedBottleneck.from_state_dict(sd, "fpn.C2.1", 1, dev)
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    x = torch.randn(b, 256, 256, 256, device=dev)
    modules.FUSED_BOTTLENECK = True
    t_f = timeit(lambda: blk(x))
    yf = blk(x)
    modules.FUSED_BOTTLENECK = False
    t_u = timeit(lambda: blk(x))
    yu = blk(x)
    m = b * 256 * 256
    executed = 2.0 * m * (256 * 64 * 352 / 256 + 9 * 64 * 64 / 2.25 + 64 * 256)
    if len(sys.argv) > 2:  # meta for profiles/summarize_pmc.py traffic: every kernel of either path ran 24 times
        with open(sys.argv[2], "w") as fh:
            json.dump({"predict_calls": 24, "workload": f"one C2 identity Bottleneck, batch {b} x 256 x 256 x 256, fused kernel "
                                                        "and three-launch path, 24 calls each"}, fh)
    print(json.dumps({"shape": [b, 256, 256, 256], "fused_ms": round(t_f, 4), "three_launch_ms": round(t_u, 4),
                      "fused_executed_tflops": round(executed / t_f / 1e9, 1), "bit_identical": bool(torch.equal(yf, yu)),
                      "max_diff": (yf - yu).abs().max().item()}))


if __name__ == "__main__":
    main()
