"""Host-side mirror of the reference's nn graph on the hot path, executed with the fused HIP ops.

Two faces:
  * `reference_schema(arch)` — an nn.Module tree whose state_dict keys/shapes equal the reference
    MaskRCNN's (fpn.C1.0.weight … mask.conv5.bias; model.py:97-131,174-189,214-270,596-607,724-740,
    848-866), so a user's `mask_rcnn_coco.pth` loads unchanged. It carries parameters only.
  * `Fused*` — inference blocks built from such a state_dict: BatchNorm (eval) and conv bias are folded
    into an fp32 (scale, shift) epilogue, weights repacked OIHW → OHWI, activations channels-last. Every
    convolution/GEMM runs in torch.ops.maskrcnn.conv_bn_act (libmaskrcnn_hip.so); there is no PyTorch
    fallback.
"""
from __future__ import annotations

import math

import os

import torch
import torch.nn as nn

from . import ops

BN_EPS = 1e-3  # model.py:180,183,185,225,261,732,734,857-863
LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}  # model.py:219
NUM_CLASSES = 81


# --------------------------------------------------------------------------------------------------
# state-dict schema (parameters only)
# --------------------------------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride)
        self.bn1 = nn.BatchNorm2d(planes, eps=BN_EPS, momentum=0.01)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3)
        self.bn2 = nn.BatchNorm2d(planes, eps=BN_EPS, momentum=0.01)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1)
        self.bn3 = nn.BatchNorm2d(planes * 4, eps=BN_EPS, momentum=0.01)
        self.downsample = downsample
        self.stride = stride


def _stage(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, kernel_size=1, stride=stride),
                             nn.BatchNorm2d(planes * 4, eps=BN_EPS, momentum=0.01))
    layers = [_Bottleneck(inplanes, planes, stride, down)]
    layers += [_Bottleneck(planes * 4, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class _Holder(nn.Module):
    pass


def reference_schema(arch: str = "resnet101", num_classes: int = NUM_CLASSES) -> nn.Module:
    """Parameter tree with the reference MaskRCNN's state_dict layout (R50: 596 keys… R101: 800)."""
    l = LAYERS[arch]
    net = _Holder()
    fpn = _Holder()
    # nn.Identity placeholders keep the Sequential indices of ReLU / SamePad2d / MaxPool2d
    fpn.C1 = nn.Sequential(nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3),
                           nn.BatchNorm2d(64, eps=BN_EPS, momentum=0.01), nn.Identity(), nn.Identity(),
                           nn.Identity())
    fpn.C2 = _stage(64, 64, l[0], 1)
    fpn.C3 = _stage(256, 128, l[1], 2)
    fpn.C4 = _stage(512, 256, l[2], 2)
    fpn.C5 = _stage(1024, 512, l[3], 2)
    for name, cin in (("P5", 2048), ("P4", 1024), ("P3", 512), ("P2", 256)):
        setattr(fpn, name + "_conv1", nn.Conv2d(cin, 256, kernel_size=1))
        setattr(fpn, name + "_conv2", nn.Sequential(nn.Identity(), nn.Conv2d(256, 256, kernel_size=3)))
    net.fpn = fpn
    rpn = _Holder()
    rpn.conv_shared = nn.Conv2d(256, 512, kernel_size=3)
    rpn.conv_class = nn.Conv2d(512, 6, kernel_size=1)
    rpn.conv_bbox = nn.Conv2d(512, 12, kernel_size=1)
    net.rpn = rpn
    cl = _Holder()
    cl.conv1 = nn.Conv2d(256, 1024, kernel_size=7)
    cl.bn1 = nn.BatchNorm2d(1024, eps=BN_EPS, momentum=0.01)
    cl.conv2 = nn.Conv2d(1024, 1024, kernel_size=1)
    cl.bn2 = nn.BatchNorm2d(1024, eps=BN_EPS, momentum=0.01)
    cl.linear_class = nn.Linear(1024, num_classes)
    cl.linear_bbox = nn.Linear(1024, num_classes * 4)
    net.classifier = cl
    mk = _Holder()
    for i in (1, 2, 3, 4):
        setattr(mk, f"conv{i}", nn.Conv2d(256, 256, kernel_size=3))
        setattr(mk, f"bn{i}", nn.BatchNorm2d(256, eps=BN_EPS))
    mk.deconv = nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2)
    mk.conv5 = nn.Conv2d(256, num_classes, kernel_size=1)
    net.mask = mk
    return net


def synthetic_state_dict(arch: str = "resnet50", seed: int = 0, bn_seed: int = 1) -> dict:
    """Random weights of the reference architecture (no checkpoint is available offline):
    the reference's init (model.py:1021-1035: Xavier-uniform convs, zero bias, N(0,0.01) linears) under
    torch.manual_seed(seed), plus randomised BN statistics so the folded affine is non-trivial
    (gamma~U(.5,1.5), beta~N(0,.1), mean~N(0,.1), var~U(.5,1.5), seed bn_seed)."""
    rng_state = torch.get_rng_state()
    torch.manual_seed(seed)
    net = reference_schema(arch)
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0, 0.01)
            m.bias.data.zero_()
    g = torch.Generator().manual_seed(bn_seed)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    torch.set_rng_state(rng_state)
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


# --------------------------------------------------------------------------------------------------
# folding / packing
# --------------------------------------------------------------------------------------------------
def fold_bn(sd: dict, conv: str, bn: str | None, device):
    """(scale, shift) with y = scale * conv_nobias(x) + shift == BN_eval(conv(x) + bias).
    scale = gamma / sqrt(var + eps), shift = (bias - mean) * scale + beta; kept as an fp32 epilogue
    (not folded into the weights). Without BN: scale = None, shift = bias."""
    bias = sd.get(conv + ".bias")
    if bn is None:
        return None, (None if bias is None else bias.float().contiguous().to(device))
    gamma, beta = sd[bn + ".weight"].float(), sd[bn + ".bias"].float()
    mean, var = sd[bn + ".running_mean"].float(), sd[bn + ".running_var"].float()
    scale = gamma / torch.sqrt(var + BN_EPS)
    b = bias.float() if bias is not None else torch.zeros_like(mean)
    shift = (b - mean) * scale + beta
    return scale.contiguous().to(device), shift.contiguous().to(device)


def pack_weight(w_oihw: torch.Tensor, device, cin_pad: int | None = None) -> torch.Tensor:
    """OIHW → OHWI (k = (ky*KW + kx)*Cin + ci contiguous per output channel), optional Cin zero-pad."""
    w = w_oihw.float().permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad > w.size(3):
        w = torch.cat([w, w.new_zeros(*w.shape[:3], cin_pad - w.size(3))], dim=3)
    return w.contiguous().to(device)


PRECISIONS = ("f32", "f16x3", "f16")
# "f32+f16x3": every 3x3 stride-1 conv, the stem and every layer that writes a k-blocked tensor stay on the exact-fp32 MFMA
# kernels (Winograd / stem / direct) exactly as in "f32"; the GEMM-shaped layers with K >= MIX_MIN_K whose output is plain
# NHWC — Bottleneck conv3 of C4 / C5, the downsample convs, the classifier's three GEMMs, the mask head's transposed conv and
# 1x1 conv — take the error-compensated fp16x3 split (fp32-grade: it passes the same 1e-4 tests; 5x fewer MFMA cycles per
# multiply-add than v_mfma_f32_32x32x2_f32). The short-K expansion layers of C2 / C3 are HBM-bound and stay f32. Activations
# are fp32 in HBM everywhere. Reported by bench.py as an alt_precision, never as the headline.
MIXED = "f32+f16x3"
MIX_MIN_K = 256


def layer_precision(precision: str, k: int, gemm_out_nhwc: bool = True) -> str:
    """The contraction mode of ONE layer under a pipeline-level precision."""
    if precision != MIXED:
        return precision
    return "f16x3" if (gemm_out_nhwc and k >= MIX_MIN_K) else "f32"


def _tuning(name: str, default: str) -> str:
    """A/B switches of the routing below. Three are for users and always honoured (MRCNN_WINOGRAD, MRCNN_WINOGRAD4,
    MRCNN_SUB_BATCHES — a numerically different or differently scheduled mode each); every other one is a measurement aid and is
    read ONLY when MRCNN_TUNING=1 is set beside it (tools/ and bench.py's alt entries do that): a stray variable in a user's
    environment cannot re-route the product. Tests set the module attributes instead."""
    return os.environ.get(name, default) if os.environ.get("MRCNN_TUNING") == "1" else default


# f32 mode: 3x3 stride-1 SAME convs run the fused Winograd F(2x2,3x3) kernel (fp32 MFMA, 2.25x fewer multiply-adds;
# results differ from the direct kernel by the transforms' rounding, ~1e-5 abs on unit-scale activations).
# MRCNN_WINOGRAD=0 keeps every conv on the direct implicit-GEMM kernel (bitwise an fmaf chain).
WINOGRAD = os.environ.get("MRCNN_WINOGRAD", "1") != "0"
# f32 mode: the 7x7 stride-2 stem runs its own kernel (same fp32 MFMA arithmetic as the generic one, 3x faster);
# MRCNN_STEM_KERNEL=0 sends it through the generic implicit-GEMM kernel.
STEM_KERNEL = _tuning("MRCNN_STEM_KERNEL", "1") != "0"
# f32 Winograd mode: with MRCNN_FUSED_BOTTLENECK=1 the stride-1 identity Bottlenecks with planes = 64 (ResNet C2 blocks
# 1, 2) run as ONE launch of the whole-block kernel (csrc/bottleneck.hip; bit-identical to the three-launch path, 0.8x its
# fabric traffic). Off by default: at batch 8 x 256^2 x 256 it measures 0.735 ms per block against 0.704 ms for the three
# launches — one workgroup per CU runs its three GEMM phases back to back, so nothing overlaps the
# HBM-bound conv3 epilogue, while the per-layer kernels overlap five workgroups per CU there.
FUSED_BOTTLENECK = _tuning("MRCNN_FUSED_BOTTLENECK", "0") == "1"
# f32 Winograd mode: on the large pyramid levels the RPN's two 1x1 heads run inside the Winograd kernel of conv_shared
# (ops.conv3x3_winograd_heads): the 512-channel shared activation never reaches HBM. A workgroup then owns whole M tiles
# (64 tile positions each), so only levels with many of them qualify: at least RPN_HEADS_MIN_TILES PER IMAGE — P2 and P3
# of a 1024^2 or 832 x 1344 input. The rule looks at the image size only, never at the batch: the two forms add the 512
# channels up in different groupings, and image i of a batch must equal image i alone bit for bit.
# MRCNN_RPN_FUSED_HEADS=0 keeps the separate 18-channel head conv everywhere.
RPN_FUSED_HEADS = _tuning("MRCNN_RPN_FUSED_HEADS", "1") != "0"
RPN_HEADS_MIN_TILES = 64
# Winograd F(4x4,3x3) (ops.conv3x3_winograd4: 4x instead of 2.25x fewer multiply-adds, max |err| ~2e-5 at unit scale) for
# the layers that ask for it — the FPN smoothing convs and the RPN's shared conv, the big 3x3 layers at the END of the
# trunk, where its rounding does not travel through further layers — on maps of at least WINOGRAD4_MIN_TILES M tiles
# (16 x 32 output pixels) PER IMAGE (never a function of the batch: image i alone == slice i of the batch); the RPN heads
# are fused into it from WINOGRAD4_HEADS_MIN_TILES per image. MRCNN_WINOGRAD4=0 keeps F(2x2) everywhere.
WINOGRAD4 = os.environ.get("MRCNN_WINOGRAD4", "1") != "0"
WINOGRAD4_TRUNK = _tuning("MRCNN_WINOGRAD4_TRUNK", "1") != "0"   # also the Bottleneck conv2 layers (C2-C4 sizes)
WINOGRAD4_MIN_TILES = int(_tuning("MRCNN_W4_MIN_TILES", "8"))             # (the environment forms: tuning sweeps only)
WINOGRAD4_HEADS_MIN_TILES = int(_tuning("MRCNN_W4_HEADS_MIN_TILES", "32"))
# Bottlenecks with planes = 64 (ResNet C2) whose conv2 takes the F(4x4) kernel: conv3 (1x1 expansion + BN + residual + ReLU)
# runs inside that kernel's epilogue (ops.conv3x3_winograd4_conv3) — the 64-channel map between them never reaches HBM and
# one launch per block is gone; bit-identical to the two launches it replaces. MRCNN_FUSED_CONV3=0 keeps them apart.
FUSED_CONV3 = _tuning("MRCNN_FUSED_CONV3", "1") != "0"


def winograd4_tiles_per_image(h: int, w: int) -> int:
    return -(-(h // 4) // 4) * -(-(w // 4) // 8)
# "f16" mode (BASELINE config 5's fp16 MFMA path): activations are fp16 in HBM between convs — fp32 only at the boundary,
# at the RoIAlign inputs (the smoothed pyramid) and at the head outputs. A conv reading an fp16 tensor sees exactly the
# operand it would have rounded an fp32 tensor to; residual adds, the stem max-pool and the stores see fp16.
# MRCNN_F16_ACT=0 keeps every activation fp32 (round 1's form of the mode).
# The classifier head (RoIAlign 7x7 + three GEMMs) skips the RoI slots beyond each image's proposal count — the reference's rois
# tensor holds only the boxes NMS kept (model.py:1366-1374), this pipeline's has proposal_count slots per image with a count.
# MRCNN_SKIP_EMPTY_ROI_TILES=0 computes every slot (same valid rows bit for bit; A/B measurements).
SKIP_EMPTY_ROI_TILES = _tuning("MRCNN_SKIP_EMPTY_ROI_TILES", "1") != "0"
F16_ACT = _tuning("MRCNN_F16_ACT", "1") != "0"
# f32 mode: the stem's conv + BN + ReLU + SamePad + max-pool as ONE exact-fp32 launch on three real channels (csrc/stem.hip:
# stem7x7_s2_pool_f32, round 5) when H and W are multiples of 4; MRCNN_STEM_POOL=0 keeps the stem kernel + the separate max-pool.
STEM_POOL = _tuning("MRCNN_STEM_POOL", "1") != "0"
# The large stride-1 layers of the "f16" mode (fp16 NHWC input, Cin % 64 == 0, Cout % 256 == 0) run the pipelined kernel
# (csrc/conv_f16p.hip: eight waves, LDS-DMA in flight across barriers — 1.4-1.5x the 128x128-tile kernel on the 3x3 layers);
# MRCNN_F16_PIPELINED=0 keeps them on conv_igemm_f16.
F16_PIPELINED = _tuning("MRCNN_F16_PIPELINED", "1") != "0"
# "f16" mode: the stem's conv + BN + ReLU + max-pool as one fp16-MFMA launch (csrc/stem.hip: stem7x7_s2_pool_f16). 0 = the
# exact-fp32 stem with an fp16 store + the separate fp16 max-pool (rounds 2-3).
F16_STEM_POOL = _tuning("MRCNN_F16_STEM_POOL", "1") != "0"
# "f16" mode: a ResNet C2 block (planes 64, stride 1) as ONE launch (csrc/bottleneck_f16.hip: both 64-channel maps stay on chip,
# x is read once). MRCNN_F16_FUSED_C2=0 keeps the three / four per-layer launches.
F16_FUSED_C2 = _tuning("MRCNN_F16_FUSED_C2", "1") != "0"
# "f16" mode: the mask head's deconv + ReLU + conv5 + sigmoid as ONE launch (csrc/mask_tail_f16.hip: the deconv's fp16 map stays in
# registers). MRCNN_F16_FUSED_MASK_TAIL=0 keeps the two launches.
F16_FUSED_MASK_TAIL = _tuning("MRCNN_F16_FUSED_MASK_TAIL", "1") != "0"
# the RPN heads run inside that kernel on levels of at least this many pixels PER IMAGE (never a function of the batch: the two
# forms round differently, and image i of a batch must equal image i alone); below it the 18-channel conv is a launch of its own
F16_HEADS_MIN_PIXELS = int(_tuning("MRCNN_F16_HEADS_MIN_PIXELS", "4096"))


class ConvWeight:
    """A packed conv/GEMM weight [Cout,KH,KW,Cin] (OHWI) for one of the contraction modes:
       f32    exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)                      — the default, parity mode
       f16x3  fp16-operand MFMA, error-compensated 3-product split          — fp32-grade accuracy (~2^-21)
       f16    fp16-operand MFMA, plain (BASELINE config 5's fp16 MFMA path) — ~2^-11 per term"""

    def __init__(self, w_ohwi: torch.Tensor, precision: str = "f32", winograd4: bool = False):
        assert precision in PRECISIONS, precision   # one layer's mode: "f32+f16x3" is resolved per layer (layer_precision)
        self.precision = precision
        self.shape = tuple(w_ohwi.shape)
        self.u = None
        self.u4 = None
        if precision == "f32":
            self.w = w_ohwi.contiguous()
            if WINOGRAD and tuple(self.shape[1:3]) == (3, 3) and self.shape[3] % 8 == 0 and self.w.is_cuda:
                self.u = ops.winograd_weights(self.w)
                if winograd4 and WINOGRAD4 and self.shape[0] % 64 == 0:
                    self.u4 = ops.winograd4_weights(self.w)
        else:
            cin = w_ohwi.size(3)
            if cin % 8:  # fp16 kernel loads 8 halves at a time
                w_ohwi = torch.cat([w_ohwi, w_ohwi.new_zeros(*w_ohwi.shape[:3], 8 - cin % 8)], dim=3)
            self.w_hi, self.w_lo = ops.split_f16(w_ohwi.contiguous())
            if precision == "f16":
                self.w_lo = None
        self.cin = (self.w if precision == "f32" else self.w_hi).size(3)

    def deconv2x2(self, x, bias4, activation=0):
        """self holds the [4*Cout,1,1,Cin] repack of a 2x2 stride-2 transposed-conv weight."""
        if self.precision == "f32":
            return ops.deconv2x2(x, self.w, bias4, activation, 0)
        return ops.deconv2x2(x, (self.w_hi, self.w_lo), bias4, activation, 3 if self.precision == "f16x3" else 1)

    def takes_winograd(self, h, w, stride=1, pad=(1, 1, 1, 1), residual=None, relu=False):
        """Would conv() run the Winograd kernel for an [., h, w, .] input with these arguments?"""
        return (self.u is not None and stride == 1 and tuple(pad) == (1, 1, 1, 1) and residual is None
                and relu in (False, True, 0, 1) and h % 2 == 0 and w % 2 == 0)

    def takes_winograd4(self, h, w, stride=1, pad=(1, 1, 1, 1), residual=None, relu=False, batch=1):
        """Would conv() run the F(4x4) kernel for a K-BLOCKED [., batch, h, w, .] input with these arguments? (`batch` only
        enters through the kernel's 32-bit offset limit — past it the layer falls back to F(2x2) —, never through the
        tile-count rule: image i alone == slice i of a batch.)"""
        return (self.u4 is not None and self.takes_winograd(h, w, stride, pad, residual, relu)
                and ops.conv3x3_winograd4_supported(h, w, self.shape[3], self.shape[0], batch)
                and winograd4_tiles_per_image(h, w) >= WINOGRAD4_MIN_TILES)

    def takes_pipelined(self, x, stride=1, pad=(0, 0, 0, 0), relu=False, residual=None, res_div=1, out_f16=False,
                        out="nhwc"):
        """Would conv() run the pipelined fp16 kernel (ops.conv_f16_pipelined) for this fp16 NHWC input?"""
        if not (self.precision == "f16" and F16_PIPELINED and x.dim() == 4 and x.dtype == torch.float16
                and relu in (False, True, 0, 1) and out in ("nhwc", "f16+f32")):
            return False
        if residual is not None and not (res_div in (1, 2) and residual.dtype == torch.float16 and out_f16 and out == "nhwc"):
            return False
        cout, kh, kw, cin = self.w_hi.shape
        if cout < 128:   # the 64-channel layers (C2 conv1 / conv2) are faster on conv_igemm_f16's 256 x 64 tile (measured)
            return False
        return x.size(3) == cin and ops.conv_f16_pipelined_supported(x.size(0), x.size(1), x.size(2), cin, cout, kh, kw, pad,
                                                                     stride)

    def conv(self, x, scale, shift, stride=1, pad=(0, 0, 0, 0), relu=False, residual=None, res_div=1,
             algo_cin=None, out="nhwc", out_f16=False, row_counts=None, rows_per_group=0):
        """out: "nhwc" (default), or for the f32 mode "kblocked" / "both" (ops.conv3x3_winograd): the layout the next
        Winograd conv reads. x may itself be k-blocked (5-d) when this conv takes the Winograd kernel. "f16" mode only:
        out="f16+f32" returns the pair (fp16 copy, fp32 copy) of the same result."""
        if self.precision == "f32":
            hh, ww = (x.size(2), x.size(3)) if x.dim() == 5 else (x.size(1), x.size(2))
            if x.dim() == 5 and self.takes_winograd4(hh, ww, stride, pad, residual, relu, x.size(1)):
                return ops.conv3x3_winograd4(x, self.u4, scale, shift, bool(relu), algo_cin, out)
            if self.takes_winograd(hh, ww, stride, pad, residual, relu):
                return ops.conv3x3_winograd(x, self.u, scale, shift, bool(relu), algo_cin, out)
            assert x.dim() == 4 and out in ("nhwc", "kblocked")
            # (row groups — the heads' GEMMs skipping tiles of empty RoI slots — exist in the f32 kernel; the fp16 modes
            # compute every row)
            return ops.conv_bn_act(x, self.w, scale, shift, stride, pad, relu, residual, res_div, None,
                                   algo_cin, out_kblocked=(out == "kblocked"), row_counts=row_counts,
                                   rows_per_group=rows_per_group)
        if self.takes_pipelined(x, stride, pad, relu, residual, res_div, out_f16, out):
            return ops.conv_f16_pipelined(x, self.w_hi, scale, shift, pad, bool(relu), residual,
                                          out_f16=bool(out_f16) or out == "f16+f32", out_f32=(not out_f16) or out == "f16+f32",
                                          algo_cin=algo_cin, stride=stride, res_div=res_div)
        if out == "f16+f32":   # the two copies from one launch are the pipelined kernel's; otherwise one conv + a cast
            y = self.conv(x, scale, shift, stride, pad, relu, residual, res_div, algo_cin, "nhwc", False)
            return y.to(torch.float16), y
        assert out == "nhwc"
        return ops.conv_bn_act_f16mfma(x, self.w_hi, self.w_lo, scale, shift, stride, pad, relu, residual,
                                       res_div, 3 if self.precision == "f16x3" else 1, algo_cin,
                                       out_f16=bool(out_f16) and self.precision == "f16")


class FusedConv:
    """conv (+BN) (+ReLU) with SAME-style explicit padding, NHWC."""

    def __init__(self, sd, conv, bn, device, stride=1, relu=False, same_pad_kernel: int | None = None,
                 pad=(0, 0, 0, 0), cin_pad=None, precision="f32", out_f16: bool | None = None, winograd4: bool = False):
        # "f16" mode: the output is stored fp16 unless the layer says otherwise (out_f16=False: tensors that RoIAlign
        # or the caller reads)
        self.out_f16 = (precision == "f16" and F16_ACT) if out_f16 is None else (bool(out_f16) and precision == "f16" and F16_ACT)
        self.w = ConvWeight(pack_weight(sd[conv + ".weight"], device, cin_pad), precision, winograd4)
        self.scale, self.shift = fold_bn(sd, conv, bn, device)
        self.stride, self.relu, self.same_k, self.pad = stride, relu, same_pad_kernel, pad
        self.algo_cin = sd[conv + ".weight"].size(1)  # un-padded Cin for FLOP accounting

    def pad_for(self, h, w):
        return ops.same_pad(h, w, self.same_k, 1) if self.same_k else self.pad

    def takes_winograd(self, h, w):
        return self.w.takes_winograd(h, w, self.stride, self.pad_for(h, w), None, self.relu)

    def __call__(self, x, residual=None, res_div=1, out="nhwc", row_counts=None, rows_per_group=0):
        hh, ww = (x.size(2), x.size(3)) if x.dim() == 5 else (x.size(1), x.size(2))
        return self.w.conv(x, self.scale, self.shift, self.stride, self.pad_for(hh, ww), self.relu, residual, res_div,
                           self.algo_cin, out, self.out_f16, row_counts, rows_per_group)


class FusedBottleneck:
    """Bottleneck.forward (model.py:190-211): conv1 1x1 (stride) + BN + ReLU → SamePad(3,1) + conv2 3x3 + BN +
    ReLU → conv3 1x1 + BN, + residual (identity or 1x1-stride downsample + BN), ReLU. In f32 mode this is ONE call of the C ABI
    per block (ops.bottleneck_native, what torch.ops.maskrcnn.bottleneck_forward runs); the fp16-MFMA modes run the same
    launches as separate binding calls."""

    def __init__(self, p, precision="f32", convs=None):
        self.p, self.precision, self.convs = p, precision, convs
        self.f16_block = None
        if convs is not None and precision == "f16" and F16_ACT:
            c1, c2, c3, cd = convs
            if (c1.stride == 1 and c1.w.shape[0] == 64 and c3.w.shape[0] == 256 and c1.w.w_hi.is_cuda
                    and c1.w.shape[3] == (64 if cd is not None else 256) and all(c is None or c.w.precision == "f16" for c in convs)):
                # the block's weights as the A fragments of the one-launch kernel (ops.bottleneck_c2_f16)
                self.f16_block = tuple(ops.pack_afrags_f16(c.w.w_hi) if c is not None else None for c in convs)

    @classmethod
    def from_state_dict(cls, sd, prefix, stride, device, precision="f32"):
        pre = (prefix + ".") if prefix and not prefix.endswith(".") else prefix
        if precision != "f32" or WINOGRAD:
            p12 = layer_precision(precision, 0, False)      # conv1 writes k-blocked for conv2's Winograd kernel: f32 in "f32+f16x3"
            c1 = FusedConv(sd, pre + "conv1", pre + "bn1", device, stride=stride, relu=True, precision=p12)
            c2 = FusedConv(sd, pre + "conv2", pre + "bn2", device, relu=True, same_pad_kernel=3,
                           precision=p12, winograd4=WINOGRAD4_TRUNK)
            c3 = FusedConv(sd, pre + "conv3", pre + "bn3", device, relu=True,
                           precision=layer_precision(precision, sd[pre + "conv3.weight"].size(1)))
            cd = None
            if (pre + "downsample.0.weight") in sd:
                cd = FusedConv(sd, pre + "downsample.0", pre + "downsample.1", device, stride=stride,
                               precision=layer_precision(precision, sd[pre + "downsample.0.weight"].size(1)))
            return cls(None, "f32" if precision == MIXED else precision, (c1, c2, c3, cd))
        w1 = pack_weight(sd[pre + "conv1.weight"], device)
        s1, t1 = fold_bn(sd, pre + "conv1", pre + "bn1", device)
        w2 = pack_weight(sd[pre + "conv2.weight"], device)
        s2, t2 = fold_bn(sd, pre + "conv2", pre + "bn2", device)
        w3 = pack_weight(sd[pre + "conv3.weight"], device)
        s3, t3 = fold_bn(sd, pre + "conv3", pre + "bn3", device)
        wd = sdn = tdn = None
        if (pre + "downsample.0.weight") in sd:
            wd = pack_weight(sd[pre + "downsample.0.weight"], device)
            sdn, tdn = fold_bn(sd, pre + "downsample.0", pre + "downsample.1", device)
        return cls((w1, s1, t1, w2, s2, t2, w3, s3, t3, wd, sdn, tdn, int(stride)))

    def __call__(self, x):
        if self.convs is None:
            return torch.ops.maskrcnn.bottleneck_forward(x, *self.p)
        c1, c2, c3, cd = self.convs
        if (self.f16_block is not None and F16_FUSED_C2 and x.dim() == 4 and x.dtype == torch.float16
                and ops.bottleneck_c2_f16_supported(x.size(0), x.size(1), x.size(2), x.size(3), 64, cd is not None)):
            f1, f2, f3, fd = self.f16_block
            return ops.bottleneck_c2_f16(x, f1, c1.scale, c1.shift, f2, c2.scale, c2.shift, f3, c3.scale, c3.shift, fd,
                                         None if cd is None else cd.scale, None if cd is None else cd.shift)
        if (FUSED_BOTTLENECK and WINOGRAD and self.precision == "f32" and cd is None and c1.stride == 1
                and c2.w.u is not None and x.dim() == 4
                and ops.bottleneck_fused_supported(x.size(1), x.size(2), x.size(3), c1.w.shape[0])):
            return ops.bottleneck_fused(x, c1.w.w, c1.scale, c1.shift, c2.w.u, c2.scale, c2.shift,
                                        c3.w.w, c3.scale, c3.shift)
        if (self.precision == "f32" and x.dim() == 4 and x.dtype == torch.float32
                and all(c is None or c.w.precision == "f32" for c in self.convs)):
            # the whole block as ONE call of the C ABI (mrcnn_bottleneck_forward_f32 = torch.ops.maskrcnn.bottleneck_forward's
            # implementation): the plan below, made by the library — same launches, same results
            return ops.bottleneck_native(x, c1.w.w, c1.scale, c1.shift, c2.w.w, c2.w.u, c2.w.u4, c2.scale, c2.shift,
                                         c3.w.w, c3.scale, c3.shift, None if cd is None else cd.w.w,
                                         None if cd is None else cd.scale, None if cd is None else cd.shift, c1.stride,
                                         WINOGRAD4_MIN_TILES, FUSED_CONV3)
        return self.launch_by_launch(x)

    def launch_by_launch(self, x):
        """The block as separate binding calls (the fp16-MFMA modes and the mixed mode; tests compare the native call with it)."""
        c1, c2, c3, cd = self.convs
        res = x if cd is None else cd(x)
        oh, ow = -(-x.size(1) // c1.stride), -(-x.size(2) // c1.stride)
        # conv1's output only feeds conv2: written directly in the layout the Winograd kernel reads
        h = c1(x, out="kblocked" if (self.precision == "f32" and c2.takes_winograd(oh, ow)) else "nhwc")
        if (FUSED_CONV3 and h.dim() == 5 and c2.w.precision == "f32" and c3.w.precision == "f32" and c2.w.shape[0] == 64
                and c3.w.shape[1:3] == (1, 1) and c3.w.shape[0] % 32 == 0 and c2.relu and c3.relu
                and c2.w.takes_winograd4(oh, ow, 1, (1, 1, 1, 1), None, True, h.size(1))):
            return ops.conv3x3_winograd4_conv3(h, c2.w.u4, c2.scale, c2.shift, c3.w.w, c3.scale, c3.shift, res, c2.algo_cin)
        return c3(c2(h), residual=res)


class Bottleneck(nn.Module):
    """Drop-in for the reference's `Bottleneck` nn.Module (model.py:174-211): same constructor, same
    parameter/buffer names (conv{1,2,3}, bn{1,2,3}, downsample.{0,1}) so reference state dicts load unchanged,
    NCHW in / NCHW out. forward() runs the fused HIP path (inference, BN in eval mode); the folded (scale, shift)
    pairs and repacked weights are rebuilt whenever a parameter changes."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride)
        self.bn1 = nn.BatchNorm2d(planes, eps=BN_EPS, momentum=0.01)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3)
        self.bn2 = nn.BatchNorm2d(planes, eps=BN_EPS, momentum=0.01)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1)
        self.bn3 = nn.BatchNorm2d(planes * 4, eps=BN_EPS, momentum=0.01)
        self.downsample = downsample
        self.stride = stride
        self._fused, self._fused_key = None, None

    def _key(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def forward(self, x):
        if self.training:
            raise RuntimeError("maskrcnn_amd.Bottleneck is inference-only (call .eval(); BN statistics are frozen)")
        key = self._key()
        if self._fused is None or key != self._fused_key:
            self._fused = FusedBottleneck.from_state_dict(self.state_dict(), "", self.stride, x.device)
            self._fused_key = key
        return ops.nhwc_to_nchw(self._fused(ops.nchw_to_nhwc(x.contiguous())))


class FusedBackbone:
    """ResNet-FPN trunk (model.py:133-168, 223-270): NCHW molded image → [P2..P6] NHWC."""

    def __init__(self, sd, arch, device, prefix="fpn.", precision="f32"):
        l = LAYERS[arch]
        self.device = device
        block_precision = precision
        precision = layer_precision(precision, 0, False)   # "f32+f16x3": stem, laterals (k-blocked out) and smoothing are f32
        self.cin_pad = 4 if precision == "f32" else 8   # 16-byte pixels (fp32 path) / 8-half chunks (fp16 path)
        self.stem = FusedConv(sd, prefix + "C1.0", prefix + "C1.1", device, stride=2, relu=True,
                              pad=(3, 3, 3, 3), cin_pad=self.cin_pad, precision=precision)
        # "f16" mode: the stem runs the dedicated fp32 kernel too (exact fp32 products on the NCHW image, fp16 store): the
        # generic fp16 kernel on the 8-channel-padded image plus its layout pass cost more than the fp32 MFMAs do
        self.stem_w32 = pack_weight(sd[prefix + "C1.0.weight"], device, 4) if (precision == "f16" and F16_ACT) else None
        self.stages = []
        for name, n, stride in (("C2", l[0], 1), ("C3", l[1], 2), ("C4", l[2], 2), ("C5", l[3], 2)):
            self.stages.append([FusedBottleneck.from_state_dict(sd, f"{prefix}{name}.{i}",
                                                                stride if i == 0 else 1, device, block_precision)
                                for i in range(n)])
        self.lateral = {k: FusedConv(sd, f"{prefix}P{k}_conv1", None, device, precision=precision)
                        for k in (5, 4, 3, 2)}
        self.smooth = {k: FusedConv(sd, f"{prefix}P{k}_conv2.1", None, device, same_pad_kernel=3,
                                    precision=precision, out_f16=False, winograd4=True)
                       for k in (5, 4, 3, 2)}   # RoIAlign reads these

    def __call__(self, image_nchw):
        st = self.stem                                          # conv7x7 s2 p3 + BN + ReLU
        pooled = False
        if (st.w.precision == "f32" and st.w.shape == (64, 7, 7, 4) and image_nchw.size(2) % 2 == 0
                and image_nchw.size(3) % 2 == 0 and STEM_KERNEL):
            if STEM_POOL and image_nchw.size(2) % 4 == 0 and image_nchw.size(3) % 4 == 0:
                # round 5: conv + BN + ReLU + max-pool in one launch: the 64-channel full-resolution map never reaches memory
                x = ops.stem_pool_f32(image_nchw.contiguous(), st.w.w, st.scale, st.shift, st.algo_cin)
                pooled = True
            else:
                # dedicated kernel (csrc/stem.hip), reading the NCHW image itself: no layout pass over the image
                x = ops.stem_conv(image_nchw.contiguous(), st.w.w, st.scale, st.shift, True, st.algo_cin, nchw=True)
        elif (self.stem_w32 is not None and STEM_KERNEL and F16_PIPELINED and image_nchw.size(2) % 2 == 0
              and image_nchw.size(3) % 2 == 0):
            if F16_STEM_POOL and image_nchw.size(2) % 4 == 0 and image_nchw.size(3) % 4 == 0:
                # round 4: conv + BN + ReLU + max-pool in ONE launch on the fp16 MFMA: the 64-channel full-resolution map
                # never reaches memory (csrc/stem.hip: stem7x7_s2_pool_f16)
                x = ops.stem_pool_f16(image_nchw.contiguous(), self.stem_w32, st.scale, st.shift, st.algo_cin)
            else:
                x = ops.stem_conv(image_nchw.contiguous(), self.stem_w32, st.scale, st.shift, True, st.algo_cin, nchw=True,
                                  out_f16=True)
                x = ops.maxpool(x, 3, 2, ops.same_pad(x.size(1), x.size(2), 3, 2))
            pooled = True
        else:
            x = st(ops.nchw_to_nhwc(image_nchw.contiguous(), self.cin_pad))   # 3 → 4 (8) channels, zero-padded
        if not pooled:
            x = ops.maxpool(x, 3, 2, ops.same_pad(x.size(1), x.size(2), 3, 2))
        cs = []
        for blocks in self.stages:
            for blk in blocks:
                x = blk(x)
            cs.append(x)
        c2, c3, c4, c5 = cs
        # Lateral maps feed a Winograd conv (smoothing) and the next lateral's half-size residual read, nothing else:
        # in the f32 Winograd mode they are written k-blocked only, and read back k-blocked by both consumers.
        hw = {5: c5, 4: c4, 3: c3, 2: c2}
        kb = all(self.lateral[k].w.precision == "f32" and self.smooth[k].takes_winograd(hw[k].size(1), hw[k].size(2))
                 for k in (5, 4, 3, 2))
        lay = "kblocked" if kb else "nhwc"
        p5 = self.lateral[5](c5, out=lay)
        p4 = self.lateral[4](c4, residual=p5, res_div=2, out=lay)   # + nearest-upsampled P5 (model.py:150)
        p3 = self.lateral[3](c3, residual=p4, res_div=2, out=lay)
        p2 = self.lateral[2](c2, residual=p3, res_div=2, out=lay)
        outs, self.kblocked = [], []
        for k, p in ((2, p2), (3, p3), (4, p4), (5, p5)):
            sm = self.smooth[k]
            hh, ww = (p.size(2), p.size(3)) if p.dim() == 5 else (p.size(1), p.size(2))
            if sm.w.precision == "f32" and sm.takes_winograd(hh, ww) and sm.w.shape[0] % 8 == 0:
                # the smoothed map is read by RoIAlign (NHWC) and by the RPN's 3x3 conv (Winograd: k-blocked)
                y, yk = sm(p, out="both")
            elif sm.w.precision == "f16" and p.dtype == torch.float16 and F16_PIPELINED:
                # "f16" mode: RoIAlign reads fp32, the RPN's shared conv an fp16 copy (the operand it would round to anyway)
                yk, y = sm(p, out="f16+f32")
            else:
                y, yk = sm(p), None
            outs.append(y)
            self.kblocked.append(yk)
        outs.append(ops.maxpool(outs[3], 1, 2))                 # P6, model.py:109,161
        # the RPN's Winograd conv reads P6 k-blocked: subsample it a second time straight into that layout (64 KB per
        # image) rather than running a layout pass over the NHWC copy
        p6 = outs[4]
        p6k = None
        if (self.kblocked[3] is not None and self.kblocked[3].dim() == 5 and p6.size(1) % 2 == 0 and p6.size(2) % 2 == 0
                and p6.size(3) % 8 == 0):
            p6k = ops.maxpool(outs[3], 1, 2, out_kblocked=True)
        elif (self.kblocked[3] is not None and self.kblocked[3].dim() == 4 and self.kblocked[3].dtype == torch.float16):
            # "f16" mode (round 5): the RPN's shared conv reads an fp16 copy of every level; P6's is the same subsample of the
            # fp16 copy of P5 (the values the conv would round the fp32 map to), so the level runs the pipelined kernel like the
            # others instead of the round-1 kernel on the fp32 map (62 us for 2 184 pixels)
            p6k = ops.maxpool(self.kblocked[3], 1, 2)
        self.kblocked.append(p6k)
        return outs


class FusedRPN:
    """RPN.forward (model.py:609-649) per level; class (6) and bbox (12) 1x1 heads fused into one
    18-channel GEMM. Returns NHWC [B,H,W,18]: channels 0-5 = (bg,fg) logits x 3 anchors, 6-17 = deltas."""

    def __init__(self, sd, device, prefix="rpn.", precision="f32"):
        precision = layer_precision(precision, 0, False)   # "f32+f16x3": the RPN is Winograd + small head convs: f32
        self.precision = precision
        w = torch.cat([sd[prefix + "conv_class.weight"], sd[prefix + "conv_bbox.weight"]], 0)
        self.b_head = torch.cat([sd[prefix + "conv_class.bias"], sd[prefix + "conv_bbox.bias"]]).float() \
            .contiguous().to(device)
        # MRCNN_WINOGRAD=0 in an MRCNN_ABLATIONS build: the direct-kernel fused level; otherwise separate launches
        self.fused_level = precision == "f32" and not WINOGRAD and ops.HAVE_ABLATIONS
        if self.fused_level:
            # fused level kernel: the 512-channel shared activation never leaves the chip (ops.rpn_level_fused)
            self.w_shared = pack_weight(sd[prefix + "conv_shared.weight"], device)
            self.b_shared = sd[prefix + "conv_shared.bias"].float().contiguous().to(device)
            w32 = torch.zeros(32, w.size(1), dtype=torch.float32)
            w32[:w.size(0)] = w.float().view(w.size(0), -1)
            self.w_head32 = w32.contiguous().to(device)
            self.head_n = w.size(0)
        else:
            self.shared = FusedConv(sd, prefix + "conv_shared", None, device, relu=True, same_pad_kernel=3,
                                    precision=precision, winograd4=True)
            self.w_head = ConvWeight(pack_weight(w, device), precision)
            if precision == "f32" and self.shared.w.u is not None:   # heads inside the Winograd kernel (large levels)
                w32 = torch.zeros(32, w.size(1), dtype=torch.float32)
                w32[:w.size(0)] = w.float().view(w.size(0), -1)
                self.w_head32 = w32.contiguous().to(device)
            if precision == "f16" and F16_PIPELINED and RPN_FUSED_HEADS and w.size(1) == 512:
                # "f16" mode: heads inside the pipelined fp16 kernel's epilogue (levels whose map arrives as fp16)
                w16 = torch.zeros(32, w.size(1), dtype=torch.float16)
                w16[:w.size(0)] = w.float().view(w.size(0), -1).to(torch.float16)
                self.w_head16 = w16.contiguous().to(device)

    def __call__(self, p, p_kblocked=None):
        """p: a pyramid level NHWC; p_kblocked: the same map k-blocked, when the producer wrote it (f32 Winograd mode)."""
        if self.fused_level:
            return ops.rpn_level_fused(p, self.w_shared, self.b_shared, self.w_head32, self.b_head, self.head_n)
        if p_kblocked is not None and self.shared.takes_winograd(p.size(1), p.size(2)):
            b, h, w = p.size(0), p.size(1), p.size(2)
            sw = self.shared.w
            if (RPN_FUSED_HEADS and getattr(self, "w_head32", None) is not None
                    and sw.takes_winograd4(h, w, 1, (1, 1, 1, 1), None, True, b)
                    and winograd4_tiles_per_image(h, w) >= WINOGRAD4_HEADS_MIN_TILES):
                return ops.conv3x3_winograd4_heads(p_kblocked, sw.u4, self.shared.scale, self.shared.shift,
                                                   self.w_head32, True, self.shared.algo_cin)
            if (RPN_FUSED_HEADS and getattr(self, "w_head32", None) is not None
                    and not sw.takes_winograd4(h, w, 1, (1, 1, 1, 1), None, True, b)
                    and -(-((h // 2) * (w // 2)) // 64) >= RPN_HEADS_MIN_TILES and self.shared.w.shape[0] % 64 == 0
                    and (min(h, w) >= 16 or ops.HAVE_ABLATIONS)):
                return ops.conv3x3_winograd_heads(p_kblocked, self.shared.w.u, self.shared.scale, self.shared.shift,
                                                  self.w_head32, True, self.shared.algo_cin)
            p = p_kblocked
        if p_kblocked is not None and p_kblocked.dtype == torch.float16 and self.precision == "f16":
            p = p_kblocked   # "f16" mode: the producer's fp16 NHWC copy of the level
            sh = self.shared
            pad = sh.pad_for(p.size(1), p.size(2))
            if (getattr(self, "w_head16", None) is not None and sh.w.takes_pipelined(p, 1, pad, True, None, 1, True)
                    and p.size(1) * p.size(2) >= F16_HEADS_MIN_PIXELS):
                return ops.conv_f16_pipelined_heads(p, sh.w.w_hi, sh.scale, sh.shift, self.w_head16, pad, True,
                                                    algo_cin=sh.algo_cin)
        return self.w_head.conv(self.shared(p), None, self.b_head)   # fp32 out (the shared activation may be fp16)


class FusedClassifier:
    """Classifier.forward after roi_align (model.py:782-794). Input [R,7,7,256] NHWC; the 7x7 'valid'
    conv is one GEMM with K = 7*7*256 (OHWI weight flattening == NHWC crop flattening)."""

    def __init__(self, sd, device, prefix="classifier.", precision="f32"):
        precision = layer_precision(precision, 1024)           # "f32+f16x3": three GEMMs with K = 12544 / 1024 / 1024
        w1 = pack_weight(sd[prefix + "conv1.weight"], device)  # [1024,7,7,256]
        self.pool = w1.size(1)
        self.w1 = ConvWeight(w1.view(w1.size(0), 1, 1, -1), precision)
        self.s1, self.t1 = fold_bn(sd, prefix + "conv1", prefix + "bn1", device)
        self.conv2 = FusedConv(sd, prefix + "conv2", prefix + "bn2", device, relu=True, precision=precision)
        wl = torch.cat([sd[prefix + "linear_class.weight"], sd[prefix + "linear_bbox.weight"]], 0)
        self.w_fc = ConvWeight(wl.float().view(wl.size(0), 1, 1, wl.size(1)).contiguous().to(device), precision)
        self.b_fc = torch.cat([sd[prefix + "linear_class.bias"], sd[prefix + "linear_bbox.bias"]]) \
            .float().contiguous().to(device)
        self.num_classes = sd[prefix + "linear_class.weight"].size(0)

    def wants_f16(self) -> bool:
        """"f16" mode: RoIAlign may hand over fp16 crops (the rounding conv1 would apply to fp32 ones while staging them)."""
        return self.w1.precision == "f16" and F16_ACT and F16_PIPELINED

    def honours_row_counts(self) -> bool:
        """Do the three GEMMs skip row tiles of empty RoI slots (mrcnn_conv_bn_act_rows_f32)? Only the exact-fp32 kernel has the
        row-group form; a caller must not hand the other modes crops whose empty slots were left unwritten."""
        return self.w1.precision == "f32" and self.conv2.w.precision == "f32" and self.w_fc.precision == "f32"

    def __call__(self, pooled, roi_counts=None, rois_per_image=0):
        """roi_counts int32 [images] + rois_per_image: only the first roi_counts[i] of image i's slots hold a RoI (the reference
        runs the head on exactly those, model.py:1366-1374,1174); row tiles without one are skipped by the three GEMMs, and the
        rows of empty slots hold unspecified values (the detection stage never reads them: validity is slot < roi_counts)."""
        r = pooled.size(0)
        rc = dict(row_counts=roi_counts, rows_per_group=rois_per_image) \
            if (roi_counts is not None and SKIP_EMPTY_ROI_TILES and self.honours_row_counts()) else {}
        x = self.w1.conv(pooled.view(r, 1, 1, -1), self.s1, self.t1, relu=True, out_f16=self.conv2.out_f16, **rc)
        x = self.conv2(x, **rc)
        y = self.w_fc.conv(x, None, self.b_fc, **rc).view(r, -1)
        logits = y[:, :self.num_classes]
        bbox = y[:, self.num_classes:].reshape(r, self.num_classes, 4)
        return logits, bbox


class FusedMask:
    """Mask.forward after roi_align (model.py:894-914). Input [R,14,14,256] NHWC → [R,28,28,81] NHWC.
    The 2x2 stride-2 transposed conv is one GEMM to 4*256 channels whose epilogue scatters (dy,dx,co) straight into
    the up-sampled tensor and applies bias + ReLU; the final sigmoid is the 1x1 conv's epilogue activation."""

    def __init__(self, sd, device, prefix="mask.", precision="f32"):
        self.convs = [FusedConv(sd, f"{prefix}conv{i}", f"{prefix}bn{i}", device, relu=True,
                                same_pad_kernel=3, precision=layer_precision(precision, 0, False)) for i in (1, 2, 3, 4)]
        precision = layer_precision(precision, sd[prefix + "deconv.weight"].size(0))   # deconv and conv5: K = 256
        wt = sd[prefix + "deconv.weight"].float()  # [Cin, Cout, 2, 2]
        cin, cout = wt.size(0), wt.size(1)
        # GEMM output channel = (dy*2 + dx)*Cout + co
        self.w_de = ConvWeight(wt.permute(2, 3, 1, 0).reshape(4 * cout, 1, 1, cin).contiguous().to(device),
                               precision)
        self.b_de = sd[prefix + "deconv.bias"].float().repeat(4).contiguous().to(device)
        self.cout = cout
        self.conv5 = FusedConv(sd, prefix + "conv5", None, device, relu=2, precision=precision, out_f16=False)  # 2 = sigmoid
        self.tail16 = None
        if (precision == "f16" and F16_ACT and cin == 256 and cout == 256 and self.conv5.w.shape[0] <= 96 and self.conv5.w.shape[3] == 256
                and self.w_de.w_hi.is_cuda):
            w5 = self.conv5.w.w_hi.reshape(self.conv5.w.shape[0], 256)
            w5p = torch.cat([w5, w5.new_zeros(96 - w5.size(0), 256)], 0).contiguous()
            self.tail16 = (ops.pack_afrags_f16(self.w_de.w_hi), ops.pack_afrags_f16(w5p))

    def wants_f16(self) -> bool:
        """"f16" mode: RoIAlign may hand over fp16 crops (what conv1 would round fp32 ones to)."""
        return self.convs[0].w.precision == "f16" and F16_ACT and F16_PIPELINED

    def wants_kblocked(self, pool: int) -> bool:
        """Would the four 3x3 convs run as a k-blocked Winograd chain on [R, pool, pool, 256] crops? Then RoIAlign can
        write its output k-blocked and the first conv needs no layout pass."""
        return self.convs[0].w.precision == "f32" and all(c.takes_winograd(pool, pool) for c in self.convs)

    def __call__(self, pooled):
        x = pooled
        hh, ww = (x.size(2), x.size(3)) if x.dim() == 5 else (x.size(1), x.size(2))
        chain = self.convs[0].w.precision == "f32" and all(c.takes_winograd(hh, ww) for c in self.convs)
        assert x.dim() == 4 or chain
        for i, c in enumerate(self.convs):
            # Winograd -> Winograd: intermediate maps stay in the k-blocked layout, no transposition passes
            x = c(x, out="kblocked" if (chain and i + 1 < len(self.convs)) else "nhwc")
        if (self.tail16 is not None and F16_FUSED_MASK_TAIL and x.dim() == 4 and x.dtype == torch.float16
                and ops.mask_tail_f16_supported(x.size(0), x.size(1), x.size(2), x.size(3), 256, self.conv5.w.shape[0])):
            return ops.mask_tail_f16(x, self.tail16[0], self.b_de, self.tail16[1], self.conv5.shift)
        y = self.w_de.deconv2x2(x, self.b_de, activation=1)   # [R,2h,2w,C]: deconv + bias + ReLU, scattered in place
        return self.conv5(y)                                  # 1x1 conv + bias + sigmoid (model.py:913-914)


__all__ = ["reference_schema", "synthetic_state_dict", "fold_bn", "pack_weight", "ConvWeight", "PRECISIONS",
           "Bottleneck",
           "FusedConv",
           "FusedBottleneck", "FusedBackbone", "FusedRPN", "FusedClassifier", "FusedMask", "LAYERS",
           "BN_EPS"]
