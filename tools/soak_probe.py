#!/usr/bin/env python3
"""Run-to-run reproducibility of the heavy kernels, one at a time: N launches on constant inputs, each output compared bit for bit
with the first launch's. Which kernels ever differ tells a kernel-specific race from a machine that computes wrong now and then."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskrcnn_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
which = sys.argv[2].split(",") if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 2
x = torch.randn(B, 256, 256, 256, generator=g).to(dev)
xk = ops.nhwc_to_kblocked(x)
w = (torch.randn(256, 3, 3, 256, generator=g) * 0.02).to(dev)
u4, u2 = ops.winograd4_weights(w), ops.winograd_weights(w)
w512 = (torch.randn(512, 3, 3, 256, generator=g) * 0.02).to(dev)
u4_512 = ops.winograd4_weights(w512)
wh = torch.zeros(32, 512); wh[:18] = torch.randn(18, 512, generator=g) * 0.05
wh = wh.to(dev)
shift = torch.randn(256, generator=g).to(dev)
shift512 = torch.randn(512, generator=g).to(dev)
w1 = (torch.randn(256, 1, 1, 256, generator=g) * 0.05).to(dev)
xg = torch.randn(8000, 1, 1, 12544, generator=g).to(dev)
wg = (torch.randn(1024, 1, 1, 12544, generator=g) * 0.01).to(dev)
x64 = torch.randn(B, 256, 256, 64, generator=g).to(dev)
x64k = ops.nhwc_to_kblocked(x64)
w64 = (torch.randn(64, 3, 3, 64, generator=g) * 0.05).to(dev)
u4_64 = ops.winograd4_weights(w64)
w3 = (torch.randn(256, 1, 1, 64, generator=g) * 0.1).to(dev)
res = torch.randn(B, 256, 256, 256, generator=g).to(dev)
s64, s256 = torch.randn(64, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
xm = torch.randn(100, 14, 14, 256, generator=g).to(dev)
xmk = ops.nhwc_to_kblocked(xm)
x16 = torch.randn(8, 52, 84, 256, generator=g).half().to(dev)
w16 = (torch.randn(256, 3, 3, 256, generator=g) * 0.02).half().to(dev)
x16b = torch.randn(2, 208, 336, 256, generator=g).half().to(dev)
w16b = (torch.randn(512, 3, 3, 256, generator=g) * 0.02).half().to(dev)
wh16 = torch.zeros(32, 512, dtype=torch.float16); wh16[:18] = (torch.randn(18, 512, generator=g) * 0.05).half()
wh16 = wh16.to(dev)
img16 = torch.randn(2, 3, 832, 1344, generator=g).to(dev)
wst = torch.zeros(64, 7, 7, 4); wst[..., :3] = torch.randn(64, 7, 7, 3, generator=g) * 0.05
wst = wst.to(dev)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from maskrcnn_amd import modules
from test_gpu_conv import _block_sd
blk_id = modules.FusedBottleneck.from_state_dict(_block_sd(g, 256, 64, False), "", 1, dev, "f16")
blk_first = modules.FusedBottleneck.from_state_dict(_block_sd(g, 64, 64, True), "", 1, dev, "f16")
x16c = torch.randn(4, 208, 336, 256, generator=g).half().to(dev)
x16d = torch.randn(4, 208, 336, 64, generator=g).half().to(dev)
xmt = torch.randn(400, 14, 14, 256, generator=g).half().to(dev)
wde16 = (torch.randn(1024, 1, 1, 256, generator=g) * 0.05).half().to(dev)
w5p16 = torch.zeros(96, 256, dtype=torch.float16); w5p16[:81] = (torch.randn(81, 256, generator=g) * 0.05).half()
fde, f5 = ops.pack_afrags_f16(wde16), ops.pack_afrags_f16(w5p16.to(dev))
bde4, b5 = torch.randn(1024, generator=g).to(dev), torch.randn(81, generator=g).to(dev)
cases = {
    "mask_tail_f16": lambda: ops.mask_tail_f16(xmt, fde, bde4, f5, b5),
    "c2_f16_block_identity": lambda: blk_id(x16c),
    "c2_f16_block_first": lambda: blk_first(x16d),
    "wino2_linear_mask_head": lambda: ops.conv3x3_winograd(xmk, u2, None, shift, True, None, "kblocked"),
    "f16p_c4_conv2": lambda: ops.conv_f16_pipelined(x16, w16, None, shift, (1, 1, 1, 1), True, None, out_f16=True),
    "f16p_rpn_heads_p2": lambda: ops.conv_f16_pipelined_heads(x16b, w16b, None, shift512, wh16, (1, 1, 1, 1), True).part,
    "stem_pool_f16": lambda: ops.stem_pool_f16(img16, wst, None, None),
    "stem_pool_f32": lambda: ops.stem_pool_f32(img16, wst, None, None),
    "wino4_plain_both": lambda: ops.conv3x3_winograd4(xk, u4, None, shift, False, None, "both")[0],
    "wino4_plain_relu_kblocked": lambda: ops.conv3x3_winograd4(xk, u4, None, shift, True, None, "kblocked"),
    "wino4_heads": lambda: ops.conv3x3_winograd4_heads(xk, u4_512, None, shift512, wh, True).part,
    "wino4_conv3": lambda: ops.conv3x3_winograd4_conv3(x64k, u4_64, None, s64, w3, None, s256, res),
    "wino2_spatial": lambda: ops.conv3x3_winograd(xk, u2, None, shift, False, None, "nhwc"),
    "direct_1x1_256": lambda: ops.conv_bn_act(x, w1, None, shift, relu=True),
    "direct_gemm_k12544": lambda: ops.conv_bn_act(xg, wg, None, None, relu=True),
}
for name, fn in cases.items():
    if which and name not in which:
        continue
    first, bad, t0 = None, [], time.time()
    n = N if "gemm" not in name and "heads" not in name else max(1000, N // 4)
    for it in range(n):
        y = fn()
        if first is None:
            first = y.clone()
            continue
        if not torch.equal(y, first):
            d = (y != first)
            bad.append({"launch": it, "elements": int(d.sum()), "max_abs_diff": float((y - first).abs().max())})
    torch.cuda.synchronize()
    print(json.dumps({"kernel": name, "launches": n, "differed": len(bad), "events": bad[:6], "seconds": round(time.time() - t0, 1)}), flush=True)
