"""One ResNet C2 block of the plain-fp16 path at configs[4]'s size (batch 8, 208 x 336): the one-launch kernel
(csrc/bottleneck_f16.hip) against the per-layer launches, device time from HIP events over `reps` back-to-back calls.
argv: [first|identity] [reps]. Also the driver of the rocprofv3 --pmc passes (tools/c2_f16_pmc.sh)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from maskrcnn_amd import modules  # noqa: E402
from test_gpu_conv import _block_sd  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "identity"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
first = kind == "first"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
cin = 64 if first else 256
m = modules.FusedBottleneck.from_state_dict(_block_sd(g, cin, 64, first), "", 1, dev, "f16")
x = torch.randn(8, 208, 336, cin, generator=g).half().to(dev)


def timed(fn):
    for _ in range(3):
        fn(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"block": kind, "shape": [8, 208, 336, cin], "one_launch_ms": round(timed(m), 4)}
if os.environ.get("C2_PROBE_PER_LAYER", "1") != "0":
    out["per_layer_ms"] = round(timed(m.launch_by_launch), 4)
px = 8 * 208 * 336
out["compulsory_MB"] = round(px * (cin + 256) * 2 / 1e6, 1)
out["one_launch_GBps_on_compulsory"] = round(px * (cin + 256) * 2 / out["one_launch_ms"] / 1e6, 0)
print(json.dumps(out))
