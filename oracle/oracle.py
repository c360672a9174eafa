"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's inference hot path. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the
checker — the product (maskrcnn_amd/, maskrcnn/) never does and fails loudly without its HIP library.

Two layers:
  * native ops  — ctypes bindings to oracle/liboracle.so, the plain-C restatement of
                  c++ext/maskrcnn/csrc/cpu/{nms_cpu.cpp,crop_cpu.cpp} (nms_ref.c / crop_ref.c) and of Pillow's
                  8-bit bilinear resample (resample_ref.c: the third-party arithmetic behind utils.resize_image
                  and data.full_masks; pinned against Pillow itself).
  * graph level — torch-CPU fp32 functional restatement of the model.py / data.py / utils.py pieces on
                  the path (each function cites the reference lines it follows). torch is used here as
                  the floating-point reference for conv/BN/etc. (fp32, CPU).

Pinning: every function here is checked against (a) golden vectors in tests/golden/ produced by
tests/golden/make_golden.py from the reference itself (its compiled C++ sources via oracle/_ref and its
Python modules imported from /root/reference), (b) oracle/_ref live when present, (c) the pdb-comment
known answers in the reference (utils.py:148-151,248-289; data.py:464-471; model.py:165-166,1019).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build() -> str:
    """Compile liboracle.so (gcc, seconds). Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("nms_ref.c", "crop_ref.c", "resample_ref.c", "Makefile")]
    if (not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs)):
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        i64, i32, f32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p
        for name in ("oracle_nms_f32", "oracle_nms_f64"):
            fn = getattr(lib, name)
            fn.restype = i64
            fn.argtypes = [vp, i64, i64, i64, vp, f32, vp]
        lib.oracle_crop_forward_f32.restype = ctypes.c_int
        lib.oracle_crop_forward_f32.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, f32, i32, i32, vp]
        lib.oracle_crop_backward_f32.restype = ctypes.c_int
        lib.oracle_crop_backward_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        lib.oracle_resample_ksize.restype = ctypes.c_int
        lib.oracle_resample_ksize.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.oracle_resample_coeffs.restype = ctypes.c_int
        lib.oracle_resample_coeffs.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp]
        lib.oracle_resize_bilinear_u8.restype = ctypes.c_int
        lib.oracle_resize_bilinear_u8.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int,
                                                  ctypes.c_int]
        lib.oracle_f32_to_l8.restype = None
        lib.oracle_f32_to_l8.argtypes = [vp, i64, vp]
        _lib = lib
    return _lib


# --------------------------------------------------------------------------------------------------
# native ops (c++ext/maskrcnn)
# --------------------------------------------------------------------------------------------------
def nms(dets: torch.Tensor, threshold: float, class_ids: torch.Tensor | None = None) -> torch.Tensor:
    """nms(dets[N,5] rows (y1,x1,y2,x2,score), threshold) -> int64 [K] ascending input indices.

    Restates nms.h:15-30 + nms_cpu.cpp:11-79 (float32 / float64 dispatch at :75, any strides).
    `class_ids` (int32 [N], optional) restricts suppression to same-class pairs: the single-pass form
    of the per-class loop at model.py:1454-1475.
    """
    lib = _load()
    if dets.is_cuda:
        raise RuntimeError("oracle.nms is CPU only")
    if dets.dtype not in (torch.float32, torch.float64):
        raise NotImplementedError(f'"nms" not implemented for {dets.dtype}')  # nms_cpu.cpp:75
    n = dets.size(0) if dets.dim() > 0 else 0
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.int64)
    assert dets.dim() == 2 and dets.size(1) >= 5
    keep = torch.empty(n, dtype=torch.int64)
    cls_ptr = None
    if class_ids is not None:
        class_ids = class_ids.to(torch.int32).contiguous()
        assert class_ids.numel() == n
        cls_ptr = class_ids.data_ptr()
    fn = lib.oracle_nms_f32 if dets.dtype == torch.float32 else lib.oracle_nms_f64
    k = fn(dets.data_ptr(), n, dets.stride(0), dets.stride(1), cls_ptr, float(threshold),
           keep.data_ptr())
    return keep[:k].clone()


def crop_forward(image: torch.Tensor, boxes: torch.Tensor, box_index: torch.Tensor,
                 extrapolation_value: float, crop_height: int, crop_width: int) -> torch.Tensor:
    """crop_and_resize forward -> [N, C, crop_height, crop_width] (crop.h:14-34, crop_cpu.cpp:13-164).

    image [B,C,H,W] fp32 NCHW contiguous, boxes [N,4] fp32 normalised (y1,x1,y2,x2), box_index [N] int32.
    A box_index outside [0,B) raises (the reference printf()s and exit(-1)s, crop_cpu.cpp:47-50).
    """
    lib = _load()
    if image.dtype != torch.float32 or boxes.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float")  # .data<float>() crop_cpu.cpp:147-158
    if box_index.dtype != torch.int32:
        raise RuntimeError("expected scalar type Int")
    image = image.contiguous()
    boxes = boxes.contiguous()
    box_index = box_index.contiguous()
    b, c, h, w = image.shape
    n = boxes.size(0)
    crops = torch.empty(n, c, crop_height, crop_width, dtype=torch.float32)
    rc = lib.oracle_crop_forward_f32(image.data_ptr(), b, c, h, w, boxes.data_ptr(),
                                     box_index.data_ptr(), n, float(extrapolation_value),
                                     crop_height, crop_width, crops.data_ptr())
    if rc != 0:
        raise RuntimeError("batch_index out of range")
    return crops


def crop_backward(grads: torch.Tensor, boxes: torch.Tensor, box_index: torch.Tensor,
                  image_size) -> torch.Tensor:
    """crop_and_resize backward -> grads_image [B,C,H,W] (crop.h:36-53, crop_cpu.cpp:167-265)."""
    lib = _load()
    grads = grads.contiguous()
    boxes = boxes.contiguous()
    box_index = box_index.contiguous()
    b, c, h, w = [int(v) for v in image_size]
    n, _, ch, cw = grads.shape
    out = torch.empty(b, c, h, w, dtype=torch.float32)
    rc = lib.oracle_crop_backward_f32(grads.data_ptr(), boxes.data_ptr(), box_index.data_ptr(), n, b,
                                      c, h, w, ch, cw, out.data_ptr())
    if rc != 0:
        raise RuntimeError("batch_index out of range")
    return out


class CropFunction:
    """Call shape of c++ext/maskrcnn/__init__.py:25-45: CropFunction(h, w, extrap)(image, boxes, ind)."""

    def __init__(self, crop_height, crop_width, extrapolation_value=0):
        self.crop_height, self.crop_width = crop_height, crop_width
        self.extrapolation_value = extrapolation_value

    def __call__(self, image, boxes, box_ind):
        return crop_forward(image, boxes, box_ind, self.extrapolation_value, self.crop_height,
                            self.crop_width)


# --------------------------------------------------------------------------------------------------
# constants read by the path (config.py:54-126,199-204)
# --------------------------------------------------------------------------------------------------
class Cfg:
    BACKBONE_STRIDES = [4, 8, 16, 32, 64]
    RPN_ANCHOR_SCALES = (32, 64, 128, 256, 512)
    RPN_ANCHOR_RATIOS = [0.5, 1, 2]
    RPN_ANCHOR_STRIDE = 1
    RPN_NMS_THRESHOLD = 0.7
    RPN_NMS_MAX_ROIS_NUM = 500
    PRE_NMS_LIMIT = 500  # hard-coded at model.py:1345
    IMAGE_MIN_DIM, IMAGE_MAX_DIM = 800, 1024
    MEAN_PIXEL = [123.7, 116.8, 103.9]
    POOL_SIZE, MASK_POOL_SIZE = 7, 14
    RPN_BBOX_STD_DEV = [0.1, 0.1, 0.2, 0.2]
    DETECTION_MAX_INSTANCES = 50
    DETECTION_MIN_CONFIDENCE = 0  # CocoInferenceConfig, config.py:204 (falsy: no score filter)
    DETECTION_NMS_THRESHOLD = 0.3
    NUM_CLASSES = 81

    def __init__(self, height=1024, width=1024, **kw):
        self.IMAGE_SHAPE = np.array([height, width, 3])
        self.BACKBONE_SHAPES = np.array(
            [[int(math.ceil(height / s)), int(math.ceil(width / s))] for s in self.BACKBONE_STRIDES])
        for k, v in kw.items():
            setattr(self, k, v)


# --------------------------------------------------------------------------------------------------
# anchors (utils.py:116-291)
# --------------------------------------------------------------------------------------------------
def create_anchors(scale, ratios, shape, feature_stride, anchor_stride):
    """utils.py:116-220. float64 [H*W*len(ratios), 4]; anchor index = ((y*W + x)*R + ratio)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    heights = scale / np.sqrt(ratios)
    widths = scale * np.sqrt(ratios)
    ys = np.arange(0, shape[0], anchor_stride) * feature_stride  # cell ORIGIN, not centre (:154)
    xs = np.arange(0, shape[1], anchor_stride) * feature_stride
    cy = np.repeat(ys, len(xs))[:, None].repeat(len(ratios), 1)  # [H*W, R]
    cx = np.tile(xs, len(ys))[:, None].repeat(len(ratios), 1)
    hh = np.broadcast_to(heights, cy.shape)
    ww = np.broadcast_to(widths, cx.shape)
    centers = np.stack([cy, cx], axis=2).reshape(-1, 2).astype(np.float64)
    sizes = np.stack([hh, ww], axis=2).reshape(-1, 2)
    return np.concatenate([centers - 0.5 * sizes, centers + 0.5 * sizes], axis=1)


def create_pyramid_anchors(scales, ratios, feature_shapes, feature_strides, anchor_stride):
    """utils.py:223-291: concat over levels, scale[i] ↔ level i."""
    return np.concatenate([
        create_anchors(scales[i], ratios, feature_shapes[i], feature_strides[i], anchor_stride)
        for i in range(len(scales))], axis=0)


def anchors_for(cfg: Cfg) -> torch.Tensor:
    """model.py:991-995: float64 → .float()."""
    return torch.from_numpy(create_pyramid_anchors(
        cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, cfg.BACKBONE_SHAPES, cfg.BACKBONE_STRIDES,
        cfg.RPN_ANCHOR_STRIDE)).float()


# --------------------------------------------------------------------------------------------------
# box math (data.py:86-148)
# --------------------------------------------------------------------------------------------------
def boxes_refine(boxes: torch.Tensor, deltas: torch.Tensor) -> torch.Tensor:
    """data.py:124-148, same op order (each line one rounded fp32 tensor op)."""
    height = boxes[:, 2] - boxes[:, 0]
    width = boxes[:, 3] - boxes[:, 1]
    center_y = boxes[:, 0] + 0.5 * height
    center_x = boxes[:, 1] + 0.5 * width
    center_y = center_y + deltas[:, 0] * height
    center_x = center_x + deltas[:, 1] * width
    height = height * torch.exp(deltas[:, 2])
    width = width * torch.exp(deltas[:, 3])
    y1 = center_y - 0.5 * height
    x1 = center_x - 0.5 * width
    y2 = y1 + height
    x2 = x1 + width
    return torch.stack([y1, x1, y2, x2], dim=1)


def boxes_clamp(boxes: torch.Tensor, window) -> torch.Tensor:
    """data.py:86-92 (out-of-place). window = (y1, x1, y2, x2)."""
    wy1, wx1, wy2, wx2 = [float(v) for v in window]
    out = boxes.clone()
    out[:, 0].clamp_(wy1, wy2)
    out[:, 1].clamp_(wx1, wx2)
    out[:, 2].clamp_(wy1, wy2)
    out[:, 3].clamp_(wx1, wx2)
    return out


def boxes_scale(boxes: torch.Tensor, scale) -> torch.Tensor:
    """data.py:95-100."""
    return boxes * torch.tensor(scale, dtype=torch.float32)


# --------------------------------------------------------------------------------------------------
# RoIAlign pyramid dispatcher (model.py:276-393)
# --------------------------------------------------------------------------------------------------
def roi_levels(boxes: torch.Tensor, image_shape) -> torch.Tensor:
    """model.py:323-338: k = 4 + log2(sqrt(h*w) / (224/sqrt(H*W))), round-half-even, clamp [2,5]."""
    y1, x1, y2, x2 = boxes.chunk(4, dim=1)
    h = y2 - y1
    w = x2 - x1
    image_area = torch.tensor([float(image_shape[0] * image_shape[1])], dtype=torch.float32)
    lvl = 4 + torch.log2(torch.sqrt(h * w) / (224.0 / torch.sqrt(image_area)))
    return lvl.round().int().clamp(2, 5)[:, 0]


def roi_align(rois: torch.Tensor, feature_maps, pool_size: int, image_shape) -> torch.Tensor:
    """model.py:276-393. rois [N,4] (or [1,N,4]) normalised; feature_maps [P2..P5] each [1,C,H,W].
    Returns [N, C, pool, pool] in the original roi order."""
    boxes = rois.squeeze(0) if rois.dim() == 3 else rois
    levels = roi_levels(boxes, image_shape)
    pooled, box_to_level = [], []
    for i, level in enumerate(range(2, 6)):
        ix = torch.nonzero(levels == level)[:, 0]
        if ix.numel() == 0:
            continue
        fm = feature_maps[i]
        fm = fm.unsqueeze(0) if fm.dim() == 3 else fm
        ind = torch.zeros(ix.numel(), dtype=torch.int32)
        pooled.append(crop_forward(fm, boxes[ix].contiguous(), ind, 0.0, pool_size, pool_size))
        box_to_level.append(ix)
    pooled = torch.cat(pooled, dim=0)
    order = torch.sort(torch.cat(box_to_level))[1]
    return pooled[order]


# --------------------------------------------------------------------------------------------------
# nn graph (model.py:64-270, 582-649, 724-800, 848-920), functional over a flat state_dict whose keys
# equal the reference's (fpn.C1.0.weight, fpn.C2.0.conv1.weight, fpn.C2.0.downsample.1.running_var, …)
# --------------------------------------------------------------------------------------------------
BN_EPS = 1e-3  # model.py:180 etc.
LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}


def same_pad(x: torch.Tensor, kernel_size: int, stride: int) -> torch.Tensor:
    """model.py:64-87, formula reproduced as written (the width/height names are swapped there; F.pad's
    first pair pads the LAST dim)."""
    in_a, in_b = x.size(2), x.size(3)
    out_a = math.ceil(float(in_a) / float(stride))
    out_b = math.ceil(float(in_b) / float(stride))
    pad_a = (out_a - 1) * stride + kernel_size - in_a
    pad_b = (out_b - 1) * stride + kernel_size - in_b
    a_lo, b_lo = math.floor(pad_a / 2), math.floor(pad_b / 2)
    return F.pad(x, (a_lo, pad_a - a_lo, b_lo, pad_b - b_lo), "constant", 0)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], training=False, eps=BN_EPS)


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def bottleneck(x, sd, p, stride):
    """model.py:190-211: stride on the 1x1 conv1; SamePad(3,1) before conv2; residual add then ReLU."""
    out = F.relu(_bn(_conv(x, sd, p + ".conv1", stride=stride), sd, p + ".bn1"))
    out = F.relu(_bn(_conv(same_pad(out, 3, 1), sd, p + ".conv2"), sd, p + ".bn2"))
    out = _bn(_conv(out, sd, p + ".conv3"), sd, p + ".bn3")
    if (p + ".downsample.0.weight") in sd:
        residual = _bn(_conv(x, sd, p + ".downsample.0", stride=stride), sd, p + ".downsample.1")
    else:
        residual = x
    return F.relu(out + residual)


def resnet_stage(x, sd, p, nblocks, stride):
    """model.py:251-270 make_layer: first block carries the stride (+downsample)."""
    for i in range(nblocks):
        x = bottleneck(x, sd, f"{p}.{i}", stride if i == 0 else 1)
    return x


def stem(x, sd, p="fpn.C1"):
    """model.py:223-229: 7x7 s2 p3 conv → BN → ReLU → SamePad(3,2) → MaxPool 3x3 s2."""
    x = F.relu(_bn(_conv(x, sd, p + ".0", stride=2, padding=3), sd, p + ".1"))
    return F.max_pool2d(same_pad(x, 3, 2), kernel_size=3, stride=2)


def fpn_forward(x, sd, arch="resnet101", p="fpn"):
    """model.py:133-168 → [P2, P3, P4, P5, P6]."""
    l = LAYERS[arch]
    x = stem(x, sd, p + ".C1")
    c2 = resnet_stage(x, sd, p + ".C2", l[0], 1)
    c3 = resnet_stage(c2, sd, p + ".C3", l[1], 2)
    c4 = resnet_stage(c3, sd, p + ".C4", l[2], 2)
    c5 = resnet_stage(c4, sd, p + ".C5", l[3], 2)
    p5 = _conv(c5, sd, p + ".P5_conv1")
    p4 = _conv(c4, sd, p + ".P4_conv1") + F.interpolate(p5, scale_factor=2)  # nearest (:150)
    p3 = _conv(c3, sd, p + ".P3_conv1") + F.interpolate(p4, scale_factor=2)
    p2 = _conv(c2, sd, p + ".P2_conv1") + F.interpolate(p3, scale_factor=2)
    p5 = _conv(same_pad(p5, 3, 1), sd, p + ".P5_conv2.1")
    p4 = _conv(same_pad(p4, 3, 1), sd, p + ".P4_conv2.1")
    p3 = _conv(same_pad(p3, 3, 1), sd, p + ".P3_conv2.1")
    p2 = _conv(same_pad(p2, 3, 1), sd, p + ".P2_conv2.1")
    p6 = F.max_pool2d(p5, kernel_size=1, stride=2)  # = p5[:, :, ::2, ::2] (:109,161)
    return [p2, p3, p4, p5, p6]


def rpn_forward(x, sd, p="rpn"):
    """model.py:609-649 on one level → (logits [B,A,2], probs [B,A,2], bbox [B,A,4])."""
    x = F.relu(_conv(same_pad(x, 3, 1), sd, p + ".conv_shared"))
    logits = _conv(x, sd, p + ".conv_class").permute(0, 2, 3, 1).contiguous().view(x.size(0), -1, 2)
    probs = F.softmax(logits, dim=2)
    bbox = _conv(x, sd, p + ".conv_bbox").permute(0, 2, 3, 1).contiguous().view(x.size(0), -1, 4)
    return logits, probs, bbox


def rpn_detect(feature_maps, sd):
    """model.py:1294-1304: per level then cat(dim=1)."""
    outs = [rpn_forward(fm, sd) for fm in feature_maps]
    return [torch.cat([o[i] for o in outs], dim=1) for i in range(3)]


def rpn_refine(rpn_class, rpn_bbox, anchors, cfg: Cfg, return_dets=False):
    """model.py:1307-1382 (batch 1). Returns normalised rois [1,R,4] (and the dets handed to nms)."""
    scores = rpn_class.squeeze(0)[:, 1]
    deltas = boxes_scale(rpn_bbox.squeeze(0), cfg.RPN_BBOX_STD_DEV)
    pre = min(cfg.PRE_NMS_LIMIT, anchors.size(0))
    scores, order = scores.sort(descending=True)
    order, scores = order[:pre], scores[:pre]
    boxes = boxes_refine(anchors[order], deltas[order])
    h, w = [int(v) for v in cfg.IMAGE_SHAPE[:2]]
    boxes = boxes_clamp(boxes, [0, 0, h, w])
    dets = torch.cat((boxes, scores.unsqueeze(1)), 1)
    keep = nms(dets, cfg.RPN_NMS_THRESHOLD)[:cfg.RPN_NMS_MAX_ROIS_NUM]
    rois = (boxes[keep] / torch.tensor([h, w, h, w], dtype=torch.float32)).unsqueeze(0)
    return (rois, dets) if return_dets else rois


def classifier_forward(feature_maps, rois, sd, cfg: Cfg, p="classifier"):
    """model.py:759-800 → (logits [N,81], probs [N,81], bbox [N,81,4])."""
    x = roi_align(rois, feature_maps, cfg.POOL_SIZE, cfg.IMAGE_SHAPE)
    x = F.relu(_bn(_conv(x, sd, p + ".conv1"), sd, p + ".bn1"))
    x = F.relu(_bn(_conv(x, sd, p + ".conv2"), sd, p + ".bn2"))
    x = x.view(-1, 1024)
    logits = F.linear(x, sd[p + ".linear_class.weight"], sd[p + ".linear_class.bias"])
    probs = F.softmax(logits, dim=1)
    bbox = F.linear(x, sd[p + ".linear_bbox.weight"], sd[p + ".linear_bbox.bias"])
    return logits, probs, bbox.view(bbox.size(0), -1, 4)


def mrn_refine(rois, probs, deltas, window, cfg: Cfg, return_dets=False):
    """model.py:1389-1487. Returns (class_ids [1,D] int64, scores [1,D], boxes [1,D,4]) or (None,)*3."""
    rois = rois.squeeze(0) if rois.dim() == 3 else rois
    class_ids = torch.max(probs, dim=1)[1]
    idx = torch.arange(class_ids.size(0))
    class_scores = probs[idx, class_ids]
    deltas_specific = deltas[idx, class_ids]
    std = torch.tensor(cfg.RPN_BBOX_STD_DEV, dtype=torch.float32).view(1, 4)  # :1418 (RPN_ std dev)
    refined = boxes_refine(rois, deltas_specific * std)
    h, w = [int(v) for v in cfg.IMAGE_SHAPE[:2]]
    boxes = boxes_scale(refined, [h, w, h, w])
    boxes = torch.round(boxes_clamp(boxes, window))
    keep_bool = class_ids > 0
    if cfg.DETECTION_MIN_CONFIDENCE:
        keep_bool = keep_bool & (class_scores >= cfg.DETECTION_MIN_CONFIDENCE)
    keep = torch.nonzero(keep_bool)[:, 0]
    if keep.numel() < 1:
        return (None, None, None, None) if return_dets else (None, None, None)
    pre_ids, pre_scores, pre_rois = class_ids[keep], class_scores[keep], boxes[keep]
    nms_keep, det_log = [], []
    for class_id in torch.unique(pre_ids):
        ixs = torch.nonzero(pre_ids == class_id)[:, 0]
        ix_scores, order = pre_scores[ixs].sort(descending=True)
        ix_rois = pre_rois[ixs][order]
        d = torch.cat((ix_rois, ix_scores.unsqueeze(1)), dim=1)
        ck = nms(d, cfg.DETECTION_NMS_THRESHOLD)
        det_log.append((int(class_id), d, ck))
        nms_keep.append(keep[ixs[order[ck]]])
    nms_keep = torch.unique(torch.cat(nms_keep))  # sorted ascending; ∩ keep is a no-op subset
    top = class_scores[nms_keep].sort(descending=True)[1][:cfg.DETECTION_MAX_INSTANCES]
    kept = nms_keep[top]
    out = (class_ids[kept].unsqueeze(0), class_scores[kept].unsqueeze(0), boxes[kept].unsqueeze(0))
    return out + (det_log,) if return_dets else out


def mask_forward(feature_maps, rois, sd, cfg: Cfg, p="mask"):
    """model.py:875-920 → [N, 81, 28, 28] sigmoid masks."""
    x = roi_align(rois, feature_maps, cfg.MASK_POOL_SIZE, cfg.IMAGE_SHAPE)
    for i in (1, 2, 3, 4):
        x = F.relu(_bn(_conv(same_pad(x, 3, 1), sd, f"{p}.conv{i}"), sd, f"{p}.bn{i}"))
    x = F.relu(F.conv_transpose2d(x, sd[p + ".deconv.weight"], sd[p + ".deconv.bias"], stride=2))
    return torch.sigmoid(_conv(x, sd, p + ".conv5"))


def predict(image, window, sd, cfg: Cfg, arch="resnet101", anchors=None):
    """model.py:1140-1203 up to (and including) the mask head; excludes datalib.full_masks (PIL paste,
    out of scope). image [1,3,H,W] molded. Returns dict of every intermediate the tests compare."""
    anchors = anchors_for(cfg) if anchors is None else anchors
    fms = fpn_forward(image, sd, arch)
    _, rpn_class, rpn_bbox = rpn_detect(fms, sd)
    rois = rpn_refine(rpn_class, rpn_bbox, anchors, cfg)
    _, probs, bbox = classifier_forward(fms[:4], rois, sd, cfg)
    class_ids, scores, boxes = mrn_refine(rois, probs, bbox, window, cfg)
    out = dict(feature_maps=fms, rpn_class=rpn_class, rpn_bbox=rpn_bbox, rois=rois, probs=probs,
               bbox=bbox, class_ids=class_ids, scores=scores, boxes=boxes, masks=None)
    if class_ids is not None:
        h = int(cfg.IMAGE_SHAPE[0])
        out["masks"] = mask_forward(fms[:4], boxes.float() * 1.0 / h, sd, cfg)  # :1188 (÷h only)
    return out


# ------------------------------------------------------------------------------------------------------
# image pre-/post-processing: utils.resize_image, model.mold_image, data.full_masks
# ------------------------------------------------------------------------------------------------------
def resample_coeffs(in_size: int, out_size: int):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc (BILINEAR) → bounds [out,2], kk [out,ksize] int32."""
    lib = _load()
    ks = lib.oracle_resample_ksize(in_size, out_size)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ks), dtype=np.int32)
    got = lib.oracle_resample_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data)
    assert got == ks
    return bounds, kk


def pil_resize_u8(image: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Image.fromarray(image).resize((out_w, out_h), Image.BILINEAR) for uint8 [H,W] or [H,W,C] arrays."""
    a = np.ascontiguousarray(image, dtype=np.uint8)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    h, w, c = a.shape
    out = np.empty((out_h, out_w, c), dtype=np.uint8)
    rc = _load().oracle_resize_bilinear_u8(a.ctypes.data, h, w, c, out.ctypes.data, out_h, out_w)
    if rc != 0:
        raise ValueError("height and width must be > 0")      # PIL's error for an empty size
    return out[:, :, 0] if squeeze else out


def f32_to_l8(a: np.ndarray) -> np.ndarray:
    """Image.fromarray(float32 array).convert('L'): clamp to [0,255], truncate (Pillow Convert.c f2l)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint8)
    _load().oracle_f32_to_l8(a.ctypes.data, a.size, out.ctypes.data)
    return out


def resize_image(image: np.ndarray, min_dim=None, max_dim=None, padding=False):
    """utils.py:42-90, with scipy.misc.imresize(image, (h, w)) restated as the PIL bilinear resize it wraps.
    image uint8 [h,w,3] → (image, window (y1,x1,y2,x2), scale, padding)."""
    h, w = image.shape[:2]
    window = (0, 0, h, w)
    scale = 1
    if min_dim:
        scale = max(1, min_dim / min(h, w))                    # :62-64 scale up but not down
    if max_dim:
        image_max = max(h, w)
        if round(image_max * scale) > max_dim:                 # :67-69
            scale = max_dim / image_max
    if scale != 1:
        image = pil_resize_u8(image, round(h * scale), round(w * scale))   # :72-74 (Python round: half-to-even)
    if padding:
        h, w = image.shape[:2]
        top_pad = (max_dim - h) // 2                           # :79-84
        bottom_pad = max_dim - h - top_pad
        left_pad = (max_dim - w) // 2
        right_pad = max_dim - w - left_pad
        padding = [(top_pad, bottom_pad), (left_pad, right_pad), (0, 0)]
        image = np.pad(image, padding, mode="constant", constant_values=0)
        window = (top_pad, left_pad, h + top_pad, w + left_pad)
    return image, window, scale, padding


def mold_image(image: np.ndarray, mean_pixel) -> torch.Tensor:
    """model.py:1750-1754 + :1108-1110: float32(image) - MEAN_PIXEL (a float64 array, so the subtraction is in
    double), HWC → [1,3,H,W], .float()."""
    molded = image.astype(np.float32) - np.asarray(mean_pixel, dtype=np.float64)
    return torch.from_numpy(molded.transpose(2, 0, 1)).float().unsqueeze(0)


def full_masks(class_id: torch.Tensor, boxes: torch.Tensor, masks: torch.Tensor, height: int, width: int):
    """data.py:287-314. class_id [N], boxes [N,4] pixel (y1,x1,y2,x2), masks [N,C,mh,mw] → bool [N,height,width].
    transform.Resize((bh, bw)) / transform.Pad restated as the PIL calls they wrap."""
    out = []
    for i in range(class_id.size(0)):
        mask = (masks[i][int(class_id[i].item())] * 255.0).cpu().numpy()      # :291
        y1, x1, y2, x2 = boxes[i].tolist()
        img = f32_to_l8(mask)                                                  # :294 fromarray(F).convert('L')
        img = pil_resize_u8(img, int(y2 - y1), int(x2 - x1))                   # :295
        top, left = int(y1), int(x1)                                           # :298-303
        full = np.zeros((height, width), dtype=np.uint8)
        full[top:top + img.shape[0], left:left + img.shape[1]] = img           # :305 transform.Pad, fill 0
        out.append(torch.from_numpy(full) > 127)                               # :307-308
    return torch.stack(out, dim=0)


def decode_boxes(boxes: torch.Tensor, scale, window) -> torch.Tensor:
    """data.py:331-343."""
    if scale == 1:
        return boxes
    boxes = boxes.clone()
    boxes[:, 0] -= window[0]
    boxes[:, 1] -= window[1]
    boxes[:, 2] -= window[0]
    boxes[:, 3] -= window[1]
    s = 1.0 / (scale + 1e-5)
    return boxes * torch.Tensor([s, s, s, s])


def decode_masks(masks: torch.Tensor, scale, window):
    """data.py:264-284. masks bool [N,H,W] → uint8 [N,nh,nw] grey levels (bool array → PIL mode '1' → 'L' 0/255,
    torchvision CenterCrop to the window's size, Resize by 1/scale)."""
    if scale == 1:
        return masks
    out = []
    th, tw = window[2] - window[0], window[3] - window[1]
    for i in range(masks.size(0)):
        img = masks[i].cpu().numpy().astype(np.uint8) * 255                    # :271
        h, w = img.shape
        i0, j0 = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))        # :272 CenterCrop
        img = img[i0:i0 + th, j0:j0 + tw]
        nh, nw = round(img.shape[0] * 1.0 / scale), round(img.shape[1] * 1.0 / scale)   # :275-276
        out.append(torch.from_numpy(pil_resize_u8(img, nh, nw)))               # :277
    return torch.stack(out, dim=0)
