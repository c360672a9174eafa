"""torch.ops.maskrcnn.* — thin bindings of the C ABI (include/maskrcnn_hip.h) for torch tensors.

Signatures mirror the reference's pybind module c++ext/maskrcnn/csrc/vision.cpp:11-15:
    nms(Tensor dets, float threshold) -> Tensor                                   (nms.h:15)
    crop_forward(image, boxes, box_index, extrapolation_value, crop_height, crop_width, crops) -> ()
                                                                                    (crop.h:14-22)
    crop_backward(grads, boxes, box_index, grads_image) -> ()                      (crop.h:36-41)
plus functional / batched forms used by the sync-free pipeline.

PyTorch is plumbing here: device memory, the current HIP stream, and the dispatcher. All arithmetic
happens in libmaskrcnn_hip.so. The three reference entry points also take CPU tensors, as the reference's dispatch
does (nms.h:15-30, crop.h:14-53: CPU in -> CPU out): they are staged host -> device, run the SAME HIP kernels and
come back as CPU tensors — there is no CPU arithmetic in this build, and without a visible GPU such a call raises
(the mirror of the reference's "Not compiled with GPU support", nms.h:24). The ops without a reference counterpart
(conv_bn_act, bottleneck_forward, the batched forms) take GPU tensors only.
"""
from __future__ import annotations

import ctypes
import functools
import os

import torch

from ._lib import MaskrcnnHipError, c_f32, c_i32, c_vp, check, lib

_LIB = torch.library.Library("maskrcnn", "DEF")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif hasattr(obj, "part") and isinstance(getattr(obj, "part"), torch.Tensor):   # HeadSums
        yield obj.part
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)


def _on_device(fn):
    """Device guard for a binding: every GPU tensor argument must live on ONE device, and the call — output
    allocation, torch.cuda.current_stream() and the kernel launch — runs with that device current, whatever the
    caller's current device is (a launch on cuda:0's stream against cuda:1's pointers is a memory fault, or silent
    peer traffic over xGMI)."""
    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        dev = None
        for t in _tensors(args + tuple(kwargs.values())):
            if t.is_cuda:
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise RuntimeError(f"maskrcnn_amd.{fn.__name__}: tensor arguments on different devices "
                                       f"({dev} and {t.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return guarded


def _staging_device() -> torch.device:
    """Where a CPU-tensor call of a reference entry point is computed: the current GPU. No GPU → the call fails loudly; there
    is no CPU kernel to fall back to."""
    if not torch.cuda.is_available():
        raise RuntimeError("maskrcnn_amd: Not compiled with CPU support — CPU tensors are staged to the GPU and run the HIP "
                           "kernels, and no GPU is visible")
    return torch.device("cuda", torch.cuda.current_device())


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("maskrcnn_amd: Not compiled with CPU support (tensor is on %s); "
                               "move tensors to the GPU" % t.device)


# --------------------------------------------------------------------------------------------------
# NMS
# --------------------------------------------------------------------------------------------------
@_on_device
def nms_batched(dets: torch.Tensor, threshold: float, seg_counts: torch.Tensor | None = None,
                class_ids: torch.Tensor | None = None, use_workspace: bool = True):
    """dets [S, N, 5] fp32 (any strides) → (keep int64 [S, N] ascending indices padded with -1,
    counts int32 [S]). No host synchronisation. use_workspace=False forces the single-launch LDS path."""
    _need_gpu(dets, seg_counts, class_ids)
    if dets.dtype != torch.float32:
        raise RuntimeError(f'"nms" GPU path implemented for Float only, got {dets.dtype}')
    assert dets.dim() == 3 and dets.size(2) >= 5
    s, n = dets.size(0), dets.size(1)
    keep = torch.empty(s, n, dtype=torch.int64, device=dets.device)
    counts = torch.empty(s, dtype=torch.int32, device=dets.device)
    if seg_counts is not None:
        assert seg_counts.dtype == torch.int32 and seg_counts.is_contiguous() and seg_counts.numel() == s
    if class_ids is not None:
        assert class_ids.dtype == torch.int32 and class_ids.is_contiguous() and class_ids.numel() == s * n
    ws, ws_bytes = None, 0
    if use_workspace and n > 128:
        ws_bytes = int(lib.mrcnn_nms_workspace_bytes(s, n))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dets.device)
    check(lib.mrcnn_nms_batched_f32(dets.data_ptr(), s, n, dets.stride(0), dets.stride(1),
                                    dets.stride(2), _ptr(seg_counts), _ptr(class_ids),
                                    float(threshold), keep.data_ptr(), counts.data_ptr(), _ptr(ws), ws_bytes,
                                    _stream()))
    return keep, counts


NMS_MAX_BOXES = int(lib.mrcnn_nms_max_boxes())   # per segment on the fp32 LDS / pair-mask paths (csrc/nms.hip)


@_on_device
def nms_general(dets: torch.Tensor, threshold: float):
    """dets [N, >=5] float32 or float64, any strides, any N → (keep int64 [N] ascending input indices padded with -1, count
    int64 [1]); arithmetic in dets' own type (cpu/nms_cpu.cpp:73-79). csrc/nms_general.hip; no host synchronisation."""
    _need_gpu(dets)
    if dets.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f'"nms" not implemented for {dets.dtype}')   # AT_DISPATCH_FLOATING_TYPES
    assert dets.dim() == 2 and dets.size(1) >= 5
    n = dets.size(0)
    dt = 0 if dets.dtype == torch.float32 else 1
    keep = torch.empty(n, dtype=torch.int64, device=dets.device)
    count = torch.empty(1, dtype=torch.int64, device=dets.device)
    nbytes = int(lib.mrcnn_nms_general_workspace_bytes(n, dt))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dets.device)
    check(lib.mrcnn_nms_general(dets.data_ptr(), dt, n, dets.stride(0), dets.stride(1), float(threshold), keep.data_ptr(),
                                count.data_ptr(), ws.data_ptr(), nbytes, _stream()))
    return keep, count


def _nms(dets: torch.Tensor, threshold: float) -> torch.Tensor:
    """Reference call shape: [N,5] → int64 [K] ascending input indices on dets.device; float32 or float64 boxes, any N
    (nms.h:15-30, cpu/nms_cpu.cpp:73-79). fp32 with N <= 16384 — everything model.py asks for — runs csrc/nms.hip, the
    rest csrc/nms_general.hip. (The data-dependent output length costs one D2H read; use nms_batched to stay async.)"""
    _need_gpu(dets)
    if dets.numel() == 0:  # nms.h:20-21
        return torch.empty(0, dtype=torch.int64, device=dets.device)
    if dets.dim() != 2 or dets.size(1) < 5:
        raise RuntimeError("nms: dets must be [N, 5]")
    if dets.dtype == torch.float32 and dets.size(0) <= NMS_MAX_BOXES:
        keep, counts = nms_batched(dets.unsqueeze(0), threshold)
        return keep[0, :int(counts.item())]
    keep, count = nms_general(dets, threshold)
    return keep[:int(count.item())]


_LIB.define("nms(Tensor dets, float threshold) -> Tensor")
_LIB.impl("nms", _nms, "CUDA")


def _nms_cpu(dets: torch.Tensor, threshold: float) -> torch.Tensor:
    """CPU tensor in → CPU int64 out (nms.h:26-29 → nms_cpu, cpu/nms_cpu.cpp:69: ascending input indices): H2D, the HIP kernel,
    D2H. float32 / float64 in their own arithmetic like the reference's AT_DISPATCH_FLOATING_TYPES (cpu/nms_cpu.cpp:73-79)."""
    dev = _staging_device()
    if dets.numel() == 0:   # nms_cpu.cpp:16-18
        return torch.empty(0, dtype=torch.int64)
    return _nms(dets.to(dev), threshold).cpu()


_LIB.impl("nms", _nms_cpu, "CPU")


# --------------------------------------------------------------------------------------------------
# crop_and_resize
# --------------------------------------------------------------------------------------------------
def _check_crop_inputs(image, boxes, box_index, gpu: bool = True):
    if gpu:
        _need_gpu(image, boxes, box_index)
    if image.dtype != torch.float32 or boxes.dtype != torch.float32:
        raise RuntimeError("crop: expected scalar type Float for image and boxes")
    if box_index.dtype != torch.int32:
        raise RuntimeError("crop: expected scalar type Int for box_index")
    if image.dim() != 4 or boxes.dim() != 2 or boxes.size(1) != 4 or box_index.numel() != boxes.size(0):
        raise RuntimeError("crop: image [B,C,H,W], boxes [N,4], box_index [N] expected")


@_on_device
def crop(image: torch.Tensor, boxes: torch.Tensor, box_index: torch.Tensor,
         extrapolation_value: float, crop_height: int, crop_width: int) -> torch.Tensor:
    """Functional form: returns a fresh [N, C, crop_height, crop_width] tensor."""
    _check_crop_inputs(image, boxes, box_index)
    image, boxes, box_index = image.contiguous(), boxes.contiguous(), box_index.contiguous()
    b, c, h, w = image.shape
    n = boxes.size(0)
    crops = torch.empty(n, c, crop_height, crop_width, dtype=torch.float32, device=image.device)
    check(lib.mrcnn_crop_forward_f32(image.data_ptr(), b, c, h, w, boxes.data_ptr(),
                                     box_index.data_ptr(), n, float(extrapolation_value),
                                     int(crop_height), int(crop_width), crops.data_ptr(), _stream()))
    return crops


@_on_device
def _crop_forward(image, boxes, box_index, extrapolation_value, crop_height, crop_width, crops):
    """Out-param form of crop.h:14-22: `crops` is resized to [N,C,h,w] and overwritten
    (crop_cpu.cpp:141-143)."""
    _check_crop_inputs(image, boxes, box_index)
    _need_gpu(crops)
    if crops.dtype != torch.float32:
        raise RuntimeError("crop_forward: expected scalar type Float for crops")
    image, boxes, box_index = image.contiguous(), boxes.contiguous(), box_index.contiguous()
    b, c, h, w = image.shape
    n = boxes.size(0)
    crops.resize_(n, c, crop_height, crop_width)
    check(lib.mrcnn_crop_forward_f32(image.data_ptr(), b, c, h, w, boxes.data_ptr(),
                                     box_index.data_ptr(), n, float(extrapolation_value),
                                     int(crop_height), int(crop_width), crops.data_ptr(), _stream()))


@_on_device
def _crop_backward(grads, boxes, box_index, grads_image):
    _need_gpu(grads, boxes, box_index, grads_image)
    if grads.dtype != torch.float32 or boxes.dtype != torch.float32 or grads_image.dtype != torch.float32:
        raise RuntimeError("crop_backward: expected scalar type Float")
    if box_index.dtype != torch.int32:
        raise RuntimeError("crop_backward: expected scalar type Int for box_index")
    if not grads_image.is_contiguous() or grads_image.dim() != 4:
        raise RuntimeError("crop_backward: grads_image must be a contiguous [B,C,H,W] tensor")
    grads, boxes, box_index = grads.contiguous(), boxes.contiguous(), box_index.contiguous()
    b, c, h, w = grads_image.shape
    n, gc, ch, cw = grads.shape
    if gc != c:
        raise RuntimeError("crop_backward: channel mismatch")
    check(lib.mrcnn_crop_backward_f32(grads.data_ptr(), boxes.data_ptr(), box_index.data_ptr(), n, b,
                                      c, h, w, ch, cw, grads_image.data_ptr(), _stream()))


_LIB.define("crop_forward(Tensor image, Tensor boxes, Tensor box_index, float extrapolation_value, "
            "int crop_height, int crop_width, Tensor(a!) crops) -> ()")
_LIB.impl("crop_forward", _crop_forward, "CUDA")


def _crop_forward_cpu(image, boxes, box_index, extrapolation_value, crop_height, crop_width, crops):
    """crop.h:24-33 with CPU tensors: staged through the GPU; `crops` (a CPU float tensor of any shape) is resized in place
    to [N,C,h,w] and overwritten, as crop_cpu.cpp:141-143 does. (A box_index outside [0, B) makes the reference exit(-1),
    crop_cpu.cpp:47-50; here that box's crop is extrapolation_value, as on the GPU path.)"""
    _check_crop_inputs(image, boxes, box_index, gpu=False)
    if crops.dtype != torch.float32:
        raise RuntimeError("crop_forward: expected scalar type Float for crops")
    dev = _staging_device()
    out = crop(image.to(dev), boxes.to(dev), box_index.to(dev), extrapolation_value, crop_height, crop_width)
    crops.resize_(out.shape)
    crops.copy_(out)


_LIB.impl("crop_forward", _crop_forward_cpu, "CPU")
_LIB.define("crop_backward(Tensor grads, Tensor boxes, Tensor box_index, Tensor(a!) grads_image) -> ()")
_LIB.impl("crop_backward", _crop_backward, "CUDA")


def _crop_backward_cpu(grads, boxes, box_index, grads_image):
    """crop.h:43-52 with CPU tensors: grads_image [B,C,H,W] is zeroed and accumulated on the GPU, then copied back in place."""
    dev = _staging_device()
    if not grads_image.is_contiguous() or grads_image.dim() != 4:
        raise RuntimeError("crop_backward: grads_image must be a contiguous [B,C,H,W] tensor")
    gi = torch.empty(grads_image.shape, dtype=grads_image.dtype, device=dev)
    _crop_backward(grads.to(dev), boxes.to(dev), box_index.to(dev), gi)
    grads_image.copy_(gi)


_LIB.impl("crop_backward", _crop_backward_cpu, "CPU")
_LIB.define("crop(Tensor image, Tensor boxes, Tensor box_index, float extrapolation_value, "
            "int crop_height, int crop_width) -> Tensor")
_LIB.impl("crop", crop, "CUDA")


def _crop_cpu(image, boxes, box_index, extrapolation_value, crop_height, crop_width):
    _check_crop_inputs(image, boxes, box_index, gpu=False)
    dev = _staging_device()
    return crop(image.to(dev), boxes.to(dev), box_index.to(dev), extrapolation_value, crop_height, crop_width).cpu()


_LIB.impl("crop", _crop_cpu, "CPU")


@_on_device
def roi_align_pyramid(feature_maps, rois: torch.Tensor, pool: int, image_area: float,
                      rois_per_image: int | None = None, roi_batch: torch.Tensor | None = None,
                      return_levels: bool = False, out_kblocked: bool = False, out_f16: bool = False,
                      roi_counts: torch.Tensor | None = None):
    """model.py:276-393 in one launch on channels-last maps.

    feature_maps: [P2,P3,P4,P5], each a contiguous fp32 [B, H_l, W_l, C] (NHWC) tensor.
    rois [R,4] normalised. Returns [R, pool, pool, C] (NHWC) in roi order (+ int32 levels); out_kblocked=True returns
    [C/8, R, pool, pool, 8] instead (the layout conv3x3_winograd reads: no layout pass before the mask head); out_f16=True
    returns the NHWC result rounded to fp16 (the "f16" mode's heads: what their first conv would round the values to).
    roi_counts int32 [images] (with rois_per_image): only the first roi_counts[i] slots of image i hold a RoI; the others are
    skipped and their rows of the output (and of `levels`) are left UNWRITTEN — unspecified values, possibly NaN bit patterns:
    hand such a tensor only to consumers that skip the same rows (conv_bn_act(row_counts=...))."""
    assert len(feature_maps) == 4
    _need_gpu(rois, roi_batch, *feature_maps)
    rois = rois.contiguous()
    b, _, _, c = feature_maps[0].shape
    for fm in feature_maps:
        assert fm.is_contiguous() and fm.dtype == torch.float32 and fm.size(0) == b and fm.size(3) == c
    r = rois.size(0)
    if out_kblocked:
        assert c % 8 == 0 and not out_f16
        out = torch.empty(c // 8, r, pool, pool, 8, dtype=torch.float32, device=rois.device)
    elif out_f16:
        out = torch.empty(r, pool, pool, c, dtype=torch.float16, device=rois.device)
    else:
        out = torch.empty(r, pool, pool, c, dtype=torch.float32, device=rois.device)
    levels = torch.empty(r, dtype=torch.int32, device=rois.device) if return_levels else None
    ptrs = (c_vp * 4)(*[fm.data_ptr() for fm in feature_maps])
    hs = (c_i32 * 4)(*[fm.size(1) for fm in feature_maps])
    ws = (c_i32 * 4)(*[fm.size(2) for fm in feature_maps])
    if roi_batch is not None:
        assert roi_batch.dtype == torch.int32 and roi_batch.is_contiguous()
    if roi_counts is not None:   # only the first roi_counts[image] slots of an image hold a RoI: the others are skipped
        assert roi_counts.dtype == torch.int32 and roi_counts.is_contiguous() and roi_batch is None and rois_per_image
        assert roi_counts.numel() * rois_per_image == r
    check(lib.mrcnn_roi_align_pyramid_counted_f32(ptrs, hs, ws, b, c, rois.data_ptr(), _ptr(roi_batch), r,
                                                  int(rois_per_image or 0), _ptr(roi_counts), int(pool), float(image_area),
                                                  out.data_ptr(), 1 if out_kblocked else 2 if out_f16 else 0, _ptr(levels),
                                                  _stream()))
    return (out, levels) if return_levels else out


__all__ = ["nms_batched", "nms_general", "crop", "roi_align_pyramid", "MaskrcnnHipError"]


# --------------------------------------------------------------------------------------------------
# conv + BN + ReLU (+ residual), channels-last
# --------------------------------------------------------------------------------------------------
# When set to a list, every conv launch appends (start_event, end_event, algorithmic_flops, (M, N, K),
# algorithmic_bytes = each operand/result tensor once, kernel tag) —
# HIP events recorded on the launch stream; used by bench.py's roofline pass, never in the timed region.
CONV_PROFILE: list | None = None


@_on_device
def conv_bn_act(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor | None,
                shift: torch.Tensor | None, stride: int = 1, pad=(0, 0, 0, 0), relu: bool = False,
                residual: torch.Tensor | None = None, res_div: int = 1,
                out: torch.Tensor | None = None, algo_cin: int | None = None,
                out_kblocked: bool = False, row_counts: torch.Tensor | None = None,
                rows_per_group: int = 0) -> torch.Tensor:
    """y = act(scale * conv(x, w) + shift + residual).

    row_counts / rows_per_group: the output rows come in groups of rows_per_group RoI slots of which the first row_counts[g]
    are valid (int32 device tensor); 128-row tiles without a valid row are skipped and their output rows left untouched.

    out_kblocked: write y as [Cout/8,B,OH,OW,8] (what conv3x3_winograd reads) instead of NHWC.

    relu: False/0 none, True/1 ReLU, 2 sigmoid.
    x [B,H,W,Cin] NHWC fp32 contiguous; w [Cout,KH,KW,Cin] (OHWI) contiguous; scale/shift [Cout] or
    None; pad = (top, left, bottom, right) zero padding applied on the fly; residual [B,OH/res_div,
    OW/res_div,Cout] (or the same tensor k-blocked, 5-d). Returns y [B,OH,OW,Cout] NHWC."""
    _need_gpu(x, w, scale, shift, residual)
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    assert x.is_contiguous() and w.is_contiguous() and x.dim() == 4 and w.dim() == 4
    b, h, wd, cin = x.shape
    cout, kh, kw, wcin = w.shape
    if wcin != cin:
        raise RuntimeError(f"conv_bn_act: weight Cin {wcin} != input Cin {cin}")
    pt, pl, pb, pr = [int(v) for v in pad]
    oh = (h + pt + pb - kh) // stride + 1
    ow = (wd + pl + pr - kw) // stride + 1
    if out_kblocked:
        assert out is None and cout % 8 == 0
        out = torch.empty(cout // 8, b, oh, ow, 8, dtype=torch.float32, device=x.device)
    elif out is None:
        out = torch.empty(b, oh, ow, cout, dtype=torch.float32, device=x.device)
    else:
        assert out.is_contiguous() and tuple(out.shape) == (b, oh, ow, cout)
    for t in (scale, shift):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == cout)
    res_kblocked = residual is not None and residual.dim() == 5
    if residual is not None:
        assert residual.is_contiguous() and residual.dtype == torch.float32
        want = (cout // 8, b, oh // res_div, ow // res_div, 8) if res_kblocked else (b, oh // res_div, ow // res_div, cout)
        assert tuple(residual.shape) == want, \
            f"residual {tuple(residual.shape)} vs output {(b, oh, ow, cout)} / {res_div}"
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if row_counts is not None:
        assert residual is None and not out_kblocked and row_counts.dtype == torch.int32 and row_counts.is_contiguous()
        assert rows_per_group >= 1 and row_counts.numel() * rows_per_group == b * oh * ow
        check(lib.mrcnn_conv_bn_act_rows_f32(x.data_ptr(), b, h, wd, cin, w.data_ptr(), cout, kh, kw, int(stride), pt, pl, pb,
                                             pr, _ptr(scale), _ptr(shift), int(relu), out.data_ptr(), row_counts.data_ptr(),
                                             int(rows_per_group), _stream()))
    else:
        check(lib.mrcnn_conv_bn_act_f32(x.data_ptr(), b, h, wd, cin, w.data_ptr(), cout, kh, kw,
                                        int(stride), pt, pl, pb, pr, _ptr(scale), _ptr(shift),
                                        _ptr(residual), int(res_div), 1 if res_kblocked else 0, int(relu),
                                        out.data_ptr(), 1 if out_kblocked else 0, _stream()))
    if prof is not None:
        e1.record()
        m, k = b * oh * ow, kh * kw * (algo_cin or cin)  # algorithmic: 2*MACs of the un-padded conv
        if row_counts is not None:
            # Book what RAN: the kernel returns early for M tiles without a valid row (conv.hip), so the MFMA work is the
            # executed tiles' rows and the problem the reference poses is the valid rows only (model.py:1366-1374). The host
            # read of row_counts synchronises — profiling pass only, after the end event is on the stream.
            bm = int(lib.mrcnn_conv_rows_tile_m(cout))
            rows_exec, rows_valid = rows_executed(row_counts.tolist(), int(rows_per_group), m, bm)
            kk = kh * kw * cin
            nbytes = 4 * (rows_exec * (kk + cout) + w.numel())
            prof.append((e0, e1, 2.0 * rows_valid * k * cout, (rows_valid, cout, k), nbytes, "direct",
                         2.0 * rows_exec * k * cout, {"rows_slots": m, "rows_executed": rows_exec, "rows_valid": rows_valid}))
            return out
        nbytes = 4 * (x.numel() + w.numel() + out.numel() + (residual.numel() if residual is not None else 0))
        prof.append((e0, e1, 2.0 * m * k * cout, (m, cout, k), nbytes, "direct"))
    return out


def rows_executed(counts, rows_per_group: int, m: int, bm: int):
    """Host restatement of conv_common.hpp::tile_has_rows for the profiling pass: output rows come in groups of rows_per_group
    slots of which the first counts[g] are valid; an M tile [t*bm, (t+1)*bm) runs iff it holds a valid row. → (rows of the
    tiles that run — what the MFMA pipe computes, capped at m —, valid rows — the problem the reference poses)."""
    assert rows_per_group >= 1 and len(counts) * rows_per_group == m and bm >= 1
    rows = 0
    for t in range(-(-m // bm)):
        m0, last = t * bm, min((t + 1) * bm, m) - 1
        g0, g1 = m0 // rows_per_group, last // rows_per_group
        live = (m0 - g0 * rows_per_group) < counts[g0] or any(counts[g] > 0 for g in range(g0 + 1, g1 + 1))
        if live:
            rows += last + 1 - m0
    return rows, int(sum(min(max(int(c), 0), rows_per_group) for c in counts))


def split_f16(w: torch.Tensor):
    """fp32 → (hi, lo) fp16 planes: hi = fp16(w), lo = fp16(w - fp32(hi)); hi + lo carries 22 mantissa bits."""
    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float32)).to(torch.float16)
    return hi.contiguous(), lo.contiguous()


@_on_device
def conv_bn_act_f16mfma(x: torch.Tensor, w_hi: torch.Tensor, w_lo: torch.Tensor | None,
                        scale: torch.Tensor | None, shift: torch.Tensor | None, stride: int = 1,
                        pad=(0, 0, 0, 0), relu: bool = False, residual: torch.Tensor | None = None,
                        res_div: int = 1, products: int = 3, algo_cin: int | None = None,
                        out_f16: bool = False) -> torch.Tensor:
    """conv_bn_act on fp16-operand MFMA. w_hi/w_lo fp16 OHWI planes (split_f16).
    products = 3: error-compensated split (fp32-grade accuracy); x / residual / y fp32 NHWC.
    products = 1: plain fp16 operands; x may be fp32 or fp16 NHWC, y is fp16 when out_f16 (the residual has y's type):
    the fp16-activation form of BASELINE config 5's "fp16 MFMA path"."""
    _need_gpu(x, w_hi, w_lo, scale, shift, residual)
    x_f16 = x.dtype == torch.float16
    assert (x_f16 or x.dtype == torch.float32) and w_hi.dtype == torch.float16 and x.is_contiguous() and w_hi.is_contiguous()
    assert w_lo is None or (w_lo.dtype == torch.float16 and w_lo.is_contiguous() and w_lo.shape == w_hi.shape)
    b, h, wd, cin = x.shape
    cout, kh, kw, wcin = w_hi.shape
    if wcin != cin:
        raise RuntimeError(f"conv_bn_act_f16mfma: weight Cin {wcin} != input Cin {cin}")
    pt, pl, pb, pr = [int(v) for v in pad]
    oh = (h + pt + pb - kh) // stride + 1
    ow = (wd + pl + pr - kw) // stride + 1
    io16 = x_f16 or out_f16
    if io16 and products != 1:
        raise RuntimeError("conv_bn_act_f16mfma: fp16 activations belong to the plain-fp16 mode (products = 1)")
    out = torch.empty(b, oh, ow, cout, dtype=torch.float16 if out_f16 else torch.float32, device=x.device)
    if residual is not None:
        assert residual.is_contiguous() and tuple(residual.shape) == (b, oh // res_div, ow // res_div, cout)
        assert residual.dtype == out.dtype, "the residual has the output's storage type"
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if io16:
        check(lib.mrcnn_conv_bn_act_nhwc_f16io(x.data_ptr(), 1 if x_f16 else 0, b, h, wd, cin, w_hi.data_ptr(), cout, kh,
                                               kw, int(stride), pt, pl, pb, pr, _ptr(scale), _ptr(shift), _ptr(residual),
                                               int(res_div), int(relu), out.data_ptr(), 1 if out_f16 else 0, _stream()))
    else:
        check(lib.mrcnn_conv_bn_act_nhwc_f16mfma(x.data_ptr(), b, h, wd, cin, w_hi.data_ptr(), _ptr(w_lo), cout,
                                                 kh, kw, int(stride), pt, pl, pb, pr, _ptr(scale), _ptr(shift),
                                                 _ptr(residual), int(res_div), int(relu), int(products),
                                                 out.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m, k = b * oh * ow, kh * kw * (algo_cin or cin)
        nbytes = x.numel() * x.element_size() + out.numel() * out.element_size() + \
            (residual.numel() * residual.element_size() if residual is not None else 0) + \
            2 * w_hi.numel() * (2 if products == 3 else 1)
        prof.append((e0, e1, 2.0 * m * k * cout, (m, cout, k), nbytes, "f16"))
    return out


def conv_f16_pipelined_supported(b: int, h: int, w: int, cin: int, cout: int, kh: int, kw: int, pad=(0, 0, 0, 0),
                                 stride: int = 1) -> bool:
    """Shape gate of conv_f16_pipelined (Cin % 64 == 0, Cout % 64 == 0, <= 25 taps, 32-bit byte offsets)."""
    pt, pl, pb, pr = [int(v) for v in pad]
    return bool(lib.mrcnn_conv_f16_pipelined_supported(int(b), int(h), int(w), int(cin), int(cout), int(kh), int(kw),
                                                       int(stride), pt, pl, pb, pr))


@_on_device
def conv_f16_pipelined(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor | None, shift: torch.Tensor | None,
                       pad=(0, 0, 0, 0), relu: bool = False, residual: torch.Tensor | None = None,
                       out_f16: bool = True, out_f32: bool = False, tile_rows: int = 0, algo_cin: int | None = None,
                       stride: int = 1, res_div: int = 1, tile_cols: int = 0):
    """The pipelined plain-fp16 conv (csrc/conv_f16p.hip): x fp16 NHWC, w fp16 OHWI, fp16 residual of the output's size
    (res_div 1) or half of it (res_div 2). Returns the fp16 output, the fp32 output, or the pair (fp16, fp32)."""
    _need_gpu(x, w, scale, shift, residual)
    assert x.dtype == torch.float16 and w.dtype == torch.float16 and x.is_contiguous() and w.is_contiguous()
    assert out_f16 or out_f32
    b, h, wd, cin = x.shape
    cout, kh, kw, wcin = w.shape
    if wcin != cin:
        raise RuntimeError(f"conv_f16_pipelined: weight Cin {wcin} != input Cin {cin}")
    pt, pl, pb, pr = [int(v) for v in pad]
    oh, ow = (h + pt + pb - kh) // stride + 1, (wd + pl + pr - kw) // stride + 1
    y16 = torch.empty(b, oh, ow, cout, dtype=torch.float16, device=x.device) if out_f16 else None
    y32 = torch.empty(b, oh, ow, cout, dtype=torch.float32, device=x.device) if out_f32 else None
    if residual is not None:
        assert residual.dtype == torch.float16 and residual.is_contiguous()
        assert tuple(residual.shape) == (b, oh // res_div, ow // res_div, cout)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_conv_f16_pipelined(x.data_ptr(), b, h, wd, cin, w.data_ptr(), cout, kh, kw, int(stride), pt, pl, pb, pr,
                                       _ptr(scale), _ptr(shift), _ptr(residual), int(res_div), int(relu), _ptr(y16),
                                       _ptr(y32), int(tile_rows), int(tile_cols), _stream()))
    if prof is not None:
        e1.record()
        m, k = b * oh * ow, kh * kw * (algo_cin or cin)
        nbytes = x.numel() * 2 + (y16.numel() * 2 if out_f16 else 0) + (y32.numel() * 4 if out_f32 else 0) + \
            (residual.numel() * 2 if residual is not None else 0) + 2 * w.numel()
        prof.append((e0, e1, 2.0 * m * k * cout, (m, cout, k), nbytes, "f16p"))
    if out_f16 and out_f32:
        return y16, y32
    return y16 if out_f16 else y32


def pack_afrags_f16(w: torch.Tensor) -> torch.Tensor:
    """fp16 weights [Cout, ...] (OHWI, flattened to [Cout][K], Cout % 32 == 0, K % 32 == 0) -> the A fragments
    csrc/bottleneck_f16.hip reads: [K/32][Cout/16][64 lanes][8]."""
    _need_gpu(w)
    assert w.dtype == torch.float16 and w.is_contiguous()
    cout, k = w.size(0), w.numel() // w.size(0)
    out = torch.empty(k // 32, cout // 16, 64, 8, dtype=torch.float16, device=w.device)
    with torch.cuda.device(w.device):
        check(lib.mrcnn_pack_afrags_f16(w.data_ptr(), cout, k, out.data_ptr(), _stream()))
    return out


def bottleneck_c2_f16_supported(b: int, h: int, w: int, cin: int, planes: int, has_downsample: bool) -> bool:
    """Shape gate of bottleneck_c2_f16 (planes 64, stride 1, Cin 256 identity / 64 with the downsample branch)."""
    return bool(lib.mrcnn_bottleneck_c2_f16_supported(int(b), int(h), int(w), int(cin), int(planes), int(bool(has_downsample))))


@_on_device
def bottleneck_c2_f16(x: torch.Tensor, w1f, s1, t1, w2f, s2, t2, w3f, s3, t3, wdf=None, sd=None, td=None) -> torch.Tensor:
    """Bottleneck.forward (model.py:190-211) of a ResNet C2 block in ONE launch, plain-fp16 path (csrc/bottleneck_f16.hip):
    x fp16 NHWC [B,H,W,256] (identity block) or [B,H,W,64] (the stage's first block: wdf / sd / td = its downsample branch);
    w*f: pack_afrags_f16 of the fp16 OHWI weights; s* / t*: folded BN (fp32). → fp16 NHWC [B,H,W,256]."""
    _need_gpu(x, w1f, w2f, w3f, wdf, s1, t1, s2, t2, s3, t3, sd, td)
    assert x.dtype == torch.float16 and x.is_contiguous() and x.dim() == 4
    b, h, w, cin = x.shape
    if w1f.numel() != 64 * cin or w2f.numel() != 64 * 576 or w3f.numel() != 256 * 64 or (wdf is not None and wdf.numel() != 256 * 64):
        raise RuntimeError("bottleneck_c2_f16: weight fragments do not belong to a planes-64 block of this Cin")
    for v, n in ((s1, 64), (t1, 64), (s2, 64), (t2, 64), (s3, 256), (t3, 256), (sd, 256), (td, 256)):
        if v is not None and not (v.dtype == torch.float32 and v.numel() == n and v.is_contiguous()):
            raise RuntimeError("bottleneck_c2_f16: scale / shift vectors must be contiguous fp32 of the layer's Cout")
    y = torch.empty(b, h, w, 256, dtype=torch.float16, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_bottleneck_c2_f16(x.data_ptr(), b, h, w, cin, w1f.data_ptr(), _ptr(s1), _ptr(t1), w2f.data_ptr(), _ptr(s2),
                                      _ptr(t2), w3f.data_ptr(), _ptr(s3), _ptr(t3), _ptr(wdf), _ptr(sd), _ptr(td),
                                      y.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m = b * h * w
        kk = cin * 64 + 576 * 64 + 64 * 256 + (cin * 256 if wdf is not None else 0)       # multiply-adds per pixel
        nbytes = x.numel() * 2 + y.numel() * 2 + 2 * kk
        # the reference's layers inside this launch, each as the per-layer path books it (FLOPs, bytes: input + output (+ residual)
        # + weights) — bench.py prices a fused launch against ITS bytes and, beside that, against the layers' own floors
        layers = [(2.0 * m * cin * 64, 2 * (m * cin + m * 64 + cin * 64)), (2.0 * m * 576 * 64, 2 * (2 * m * 64 + 576 * 64)),
                  (2.0 * m * 64 * 256, 2 * (m * 64 + 2 * m * 256 + 64 * 256))]
        if wdf is not None:
            layers.append((2.0 * m * cin * 256, 2 * (m * cin + m * 256 + cin * 256)))
        prof.append((e0, e1, 2.0 * m * kk, (m, 256, kk // 256), nbytes, "f16blk", 2.0 * m * kk, {"layers": layers}))
    return y


def mask_tail_f16_supported(rois: int, h: int, w: int, cin: int, cout_deconv: int, classes: int) -> bool:
    """Shape gate of mask_tail_f16 (Cin 256, deconv Cout 256, classes <= 96, 32-bit byte offsets)."""
    return bool(lib.mrcnn_mask_tail_f16_supported(int(rois), int(h), int(w), int(cin), int(cout_deconv), int(classes)))


@_on_device
def mask_tail_f16(x: torch.Tensor, wde_frags: torch.Tensor, bias_de4: torch.Tensor, w5_frags: torch.Tensor, bias5: torch.Tensor) -> torch.Tensor:
    """The tail of Mask.forward (model.py:906-914) in ONE launch, plain-fp16 path (csrc/mask_tail_f16.hip): deconv 2x2 stride 2
    + bias + ReLU (fp16, in registers) -> conv5 1x1 + bias -> sigmoid. x fp16 NHWC [R,h,w,256]; wde_frags: pack_afrags_f16 of the
    deconv's GEMM weight [4*256,1,1,256]; w5_frags: pack_afrags_f16 of conv5's weight zero-padded to 96 rows; → fp32 [R,2h,2w,classes]."""
    _need_gpu(x, wde_frags, bias_de4, w5_frags, bias5)
    assert x.dtype == torch.float16 and x.is_contiguous() and x.dim() == 4
    r, h, w, cin = x.shape
    classes = bias5.numel()
    if wde_frags.numel() != 1024 * cin or w5_frags.numel() != 96 * cin or bias_de4.numel() != 1024:
        raise RuntimeError("mask_tail_f16: weight fragments / biases do not belong to a 256 -> 256 deconv and a <= 96-class conv5")
    assert bias_de4.dtype == torch.float32 and bias5.dtype == torch.float32 and bias_de4.is_contiguous() and bias5.is_contiguous()
    y = torch.empty(r, 2 * h, 2 * w, classes, dtype=torch.float32, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_mask_tail_f16(x.data_ptr(), r, h, w, cin, wde_frags.data_ptr(), bias_de4.data_ptr(), 256, w5_frags.data_ptr(),
                                  bias5.data_ptr(), classes, y.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m = r * h * w
        kk = cin * 1024 + 4 * 256 * classes                                   # multiply-adds per input pixel
        layers = [(2.0 * m * cin * 1024, 2 * (m * cin + m * 1024 + cin * 1024)),          # deconv as the per-layer path books it
                  (2.0 * m * 4 * 256 * classes, 2 * m * 1024 + 4 * m * 4 * classes + 2 * 256 * classes)]
        prof.append((e0, e1, 2.0 * m * kk, (m, 4 * classes, kk // (4 * classes)), x.numel() * 2 + y.numel() * 4 + 2 * (1024 + 96) * cin,
                     "f16tail", 2.0 * m * kk, {"layers": layers}))
    return y


@_on_device
def conv_f16_pipelined_heads(x: torch.Tensor, w: torch.Tensor, scale, shift, w_head32: torch.Tensor, pad=(1, 1, 1, 1),
                             relu: bool = True, tile_rows: int = 0, algo_cin: int | None = None) -> "HeadSums":
    """RPN conv_shared + both 1x1 heads in one launch on the pipelined fp16 kernel ("f16" mode; model.py:605-607,624-641):
    x fp16 NHWC, w fp16 OHWI with Cout = 512, w_head32 fp16 [32, 512] (rows 0-17: conv_class then conv_bbox). → HeadSums
    (form 4: two planes in pixel order, without the bias)."""
    _need_gpu(x, w, scale, shift, w_head32)
    assert x.dtype == torch.float16 and w.dtype == torch.float16 and x.is_contiguous() and w.is_contiguous()
    b, h, wd, cin = x.shape
    cout, kh, kw, wcin = w.shape
    assert wcin == cin and cout == 512, "the consumer (rpn_scores_deltas form 4) adds exactly two 256-channel planes"
    assert w_head32.dtype == torch.float16 and w_head32.is_contiguous() and tuple(w_head32.shape) == (32, cout)
    pt, pl, pb, pr = [int(v) for v in pad]
    oh, ow = h + pt + pb - kh + 1, wd + pl + pr - kw + 1
    part = torch.empty(cout // 256, b * oh * ow, 32, dtype=torch.float32, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_conv_f16_pipelined_heads(x.data_ptr(), b, h, wd, cin, w.data_ptr(), cout, kh, kw, pt, pl, pb, pr,
                                             _ptr(scale), _ptr(shift), 1 if relu else 0, w_head32.data_ptr(),
                                             part.data_ptr(), int(tile_rows), _stream()))
    if prof is not None:
        e1.record()
        m, k = b * oh * ow, kh * kw * (algo_cin or cin)
        prof.append((e0, e1, 2.0 * m * cout * (k + 18), (m, cout, k), x.numel() * 2 + part.numel() * 4 + 2 * w.numel(), "f16p"))
    return HeadSums(part, b, oh, ow, 4)


def _conv_bn_act_op(x, w, scale, shift, stride, pad, relu, residual, res_div):
    return conv_bn_act(x, w, scale, shift, stride, pad, relu, residual, res_div)


_LIB.define("conv_bn_act(Tensor x, Tensor w, Tensor? scale, Tensor? shift, int stride, int[] pad, "
            "bool relu, Tensor? residual, int res_div) -> Tensor")
_LIB.impl("conv_bn_act", _conv_bn_act_op, "CUDA")
_LIB.impl("conv_bn_act", lambda x, *a: _need_gpu(x), "CPU")


def same_pad(size_a: int, size_b: int, kernel: int, stride: int):
    """SamePad2d (model.py:64-87) as (top, left, bottom, right) for a [.., size_a, size_b] (H, W) input.
    The reference computes the LAST-dim pad from size(2) and vice versa (its width/height names are
    swapped); reproduced as written — it only matters when H and W need different pad amounts."""
    import math
    out_a = math.ceil(float(size_a) / float(stride))
    out_b = math.ceil(float(size_b) / float(stride))
    pad_a = max((out_a - 1) * stride + kernel - size_a, 0)
    pad_b = max((out_b - 1) * stride + kernel - size_b, 0)
    a_lo, b_lo = pad_a // 2, pad_b // 2
    # F.pad(input, (a_lo, a_hi, b_lo, b_hi)): first pair → last dim (W), second pair → H
    return (b_lo, a_lo, pad_b - b_lo, pad_a - a_lo)


@_on_device
def maxpool(x: torch.Tensor, kernel: int, stride: int, pad=(0, 0, 0, 0), out_kblocked: bool = False) -> torch.Tensor:
    """Zero-padded max-pool on NHWC fp32 (stem pool: kernel 3, stride 2, pad = same_pad(H, W, 3, 2);
    P6: kernel 1, stride 2). out_kblocked=True writes [C/8,B,OH,OW,8] (P6 for the RPN's Winograd conv)."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype in (torch.float32, torch.float16) and x.dim() == 4
    b, h, w, c = x.shape
    pt, pl, pb, pr = [int(v) for v in pad]
    oh = (h + pt + pb - kernel) // stride + 1
    ow = (w + pl + pr - kernel) // stride + 1
    if x.dtype == torch.float16:
        assert not out_kblocked
        y = torch.empty(b, oh, ow, c, dtype=torch.float16, device=x.device)
        check(lib.mrcnn_maxpool_nhwc_f16(x.data_ptr(), b, h, w, c, int(kernel), int(stride), pt, pl, pb, pr, y.data_ptr(),
                                         _stream()))
        return y
    if out_kblocked:
        assert c % 8 == 0
        y = torch.empty(c // 8, b, oh, ow, 8, dtype=torch.float32, device=x.device)
    else:
        y = torch.empty(b, oh, ow, c, dtype=torch.float32, device=x.device)
    check(lib.mrcnn_maxpool_f32(x.data_ptr(), b, h, w, c, int(kernel), int(stride), pt, pl, pb, pr, y.data_ptr(),
                                1 if out_kblocked else 0, _stream()))
    return y


@_on_device
def nchw_to_nhwc(x: torch.Tensor, channels_padded: int | None = None) -> torch.Tensor:
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 4
    b, c, h, w = x.shape
    cp = int(channels_padded or c)
    y = torch.empty(b, h, w, cp, dtype=torch.float32, device=x.device)
    check(lib.mrcnn_nchw_to_nhwc_f32(x.data_ptr(), b, c, h, w, cp, y.data_ptr(), _stream()))
    return y


@_on_device
def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 4
    b, h, w, c = x.shape
    y = torch.empty(b, c, h, w, dtype=torch.float32, device=x.device)
    check(lib.mrcnn_nhwc_to_nchw_f32(x.data_ptr(), b, c, h, w, y.data_ptr(), _stream()))
    return y


def bottleneck_fused_supported(h: int, w: int, cin: int, planes: int) -> bool:
    """Shapes the whole-block kernel covers: stride-1 identity blocks with planes = 64, Cin = 256, H and W % 16 == 0."""
    return bool(lib.mrcnn_bottleneck_fused_supported(int(h), int(w), int(cin), int(planes)))


@_on_device
def bottleneck_fused(x, w1, s1, t1, u2, s2, t2, w3, s3, t3) -> torch.Tensor:
    """Bottleneck.forward (model.py:190-211) of a stride-1 identity block in ONE launch (csrc/bottleneck.hip): both
    planes-channel intermediates stay in LDS. x [B,H,W,Cin] NHWC; w1 [P,1,1,Cin]; u2 = winograd_weights(conv2 weight
    [P,3,3,P]); w3 [4P,1,1,P]; (s, t) the folded BN/bias affine of each conv. Returns [B,H,W,4P]. Equals the
    three-launch path bit for bit."""
    _need_gpu(x, w1, s1, t1, u2, s2, t2, w3, s3, t3)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    b, h, w, cin = x.shape
    planes = w1.size(0)
    assert w1.is_contiguous() and w1.numel() == planes * cin and w3.is_contiguous() and w3.numel() == 4 * planes * planes
    assert u2.is_contiguous() and u2.numel() == 16 * planes * planes
    for t, n in ((s1, planes), (t1, planes), (s2, planes), (t2, planes), (s3, 4 * planes), (t3, 4 * planes)):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n)
    y = torch.empty(b, h, w, 4 * planes, dtype=torch.float32, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_bottleneck_fused_f32(x.data_ptr(), b, h, w, cin, w1.data_ptr(), _ptr(s1), _ptr(t1), u2.data_ptr(),
                                         _ptr(s2), _ptr(t2), w3.data_ptr(), _ptr(s3), _ptr(t3), planes, y.data_ptr(),
                                         _stream()))
    if prof is not None:
        e1.record()
        m = b * h * w
        algo = 2.0 * m * (cin * planes + 9 * planes * planes + planes * 4 * planes)
        # what the MFMA pipe executes: conv1 on 352 GEMM rows per 256 output pixels (halo + row-tile padding),
        # conv2 as Winograd (1/2.25), conv3 as is
        executed = 2.0 * m * (cin * planes * 352.0 / 256.0 + 9 * planes * planes / 2.25 + planes * 4 * planes)
        prof.append((e0, e1, algo, (m, 4 * planes, cin + 9 * planes + planes),
                     4.0 * (2 * x.numel() + y.numel() + w1.numel() + u2.numel() + w3.numel()), "bottleneck", executed))
    return y


# torch.ops.maskrcnn.bottleneck_forward sends eligible blocks to the whole-block kernel (whose conv2 is Winograd F(2x2)) only
# when BOTH switches say so — MRCNN_FUSED_BOTTLENECK=1 and Winograd not disabled (MRCNN_WINOGRAD=0 promises "every conv on the
# direct kernel, bitwise an fmaf chain"); otherwise the op is the exact three/four-launch composite. modules.py keeps this in
# step with its own flags.
BOTTLENECK_OP_FUSED = (os.environ.get("MRCNN_TUNING") == "1" and os.environ.get("MRCNN_FUSED_BOTTLENECK", "0") == "1"
                       and os.environ.get("MRCNN_WINOGRAD", "1") != "0")
_U2_CACHE: dict = {}   # data_ptr -> (weakref to the conv2 weight tensor, its version, its Winograd transform)


def _cached_winograd_weights(w2: torch.Tensor) -> torch.Tensor:
    """The transform of a conv2 weight is computed once per weight tensor, not per call. An entry is valid only for the SAME
    tensor object at the same version: an address alone is reused by the allocator for other weights."""
    import weakref
    e = _U2_CACHE.get(w2.data_ptr())
    if e is not None and e[0]() is w2 and e[1] == w2._version:
        return e[2]
    if len(_U2_CACHE) > 256:
        _U2_CACHE.clear()
    u = winograd_weights(w2)
    _U2_CACHE[w2.data_ptr()] = (weakref.ref(w2), w2._version, u)
    return u


_U4_CACHE: dict = {}   # the same for the F(4x4) transform


def _cached_winograd4_weights(w2: torch.Tensor) -> torch.Tensor:
    import weakref
    e = _U4_CACHE.get(w2.data_ptr())
    if e is not None and e[0]() is w2 and e[1] == w2._version:
        return e[2]
    if len(_U4_CACHE) > 256:
        _U4_CACHE.clear()
    u = winograd4_weights(w2)
    _U4_CACHE[w2.data_ptr()] = (weakref.ref(w2), w2._version, u)
    return u


def bottleneck_plan(batch, h, w, cin, planes, stride, have_u2, have_u4, min_tiles4, fuse_conv3) -> int:
    """Bits of the plan mrcnn_bottleneck_forward_f32 follows: 1 = conv2 on F(4x4), 2 = conv2 on F(2x2), 4 = conv2 + conv3 fused."""
    return int(lib.mrcnn_bottleneck_plan(batch, h, w, cin, planes, int(stride), int(bool(have_u2)), int(bool(have_u4)),
                                         int(min_tiles4), int(bool(fuse_conv3))))


@_on_device
def bottleneck_native(x, w1, s1, t1, w2, u2, u4, s2, t2, w3, s3, t3, wd, sd, td, stride: int, min_tiles4: int = 8,
                      fuse_conv3: bool = True) -> torch.Tensor:
    """Bottleneck.forward (model.py:190-211) as ONE call of the C ABI (mrcnn_bottleneck_forward_f32, csrc/bottleneck_op.hip):
    the library plans the block — conv1 (+ downsample), then conv2 + conv3 + residual in one launch for ResNet C2, or conv2
    (F(4x4) / F(2x2) Winograd when u4 / u2 are given, else the exact direct kernel) and conv3 + residual — and enqueues the
    launches on the current stream. x [B,H,W,Cin] NHWC fp32; w* OHWI; u2 / u4 = winograd_weights / winograd4_weights of w2 or
    None. While ops.CONV_PROFILE is collecting per-launch events, the same plan runs launch by launch from Python
    (identical launches, identical result) so that every launch can be timed."""
    _need_gpu(x, w1, s1, t1, w2, u2, u4, s2, t2, w3, s3, t3, wd, sd, td)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    b, h, w, cin = x.shape
    planes, stride = w1.size(0), int(stride)
    assert tuple(w1.shape) == (planes, 1, 1, cin) and tuple(w2.shape) == (planes, 3, 3, planes)
    assert tuple(w3.shape) == (4 * planes, 1, 1, planes) and (wd is None or tuple(wd.shape) == (4 * planes, 1, 1, cin))
    for t in (w1, w2, w3, wd, u2, u4, s1, t1, s2, t2, s3, t3, sd, td):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32)
    if CONV_PROFILE is not None:
        return _bottleneck_launch_by_launch(x, w1, s1, t1, w2, u2, u4, s2, t2, w3, s3, t3, wd, sd, td, stride, min_tiles4,
                                            fuse_conv3)
    oh, ow = -(-h // stride), -(-w // stride)
    y = torch.empty(b, oh, ow, 4 * planes, dtype=torch.float32, device=x.device)
    nbytes = int(lib.mrcnn_bottleneck_workspace_bytes(b, h, w, cin, planes, stride, int(wd is not None)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    assert ws.data_ptr() % 256 == 0
    ptrs = (ctypes.c_void_p * 14)(*[_ptr(t) for t in (w1, s1, t1, w2, u2, u4, s2, t2, w3, s3, t3, wd, sd, td)])
    check(lib.mrcnn_bottleneck_forward_f32(x.data_ptr(), b, h, w, cin, planes, stride, ptrs, int(min_tiles4), int(bool(fuse_conv3)),
                                           ws.data_ptr(), nbytes, y.data_ptr(), _stream()))
    return y


def _bottleneck_launch_by_launch(x, w1, s1, t1, w2, u2, u4, s2, t2, w3, s3, t3, wd, sd, td, stride, min_tiles4, fuse_conv3):
    """The plan of mrcnn_bottleneck_forward_f32 (asked from the library), one binding call per launch."""
    b, h, w, cin = x.shape
    planes = w1.size(0)
    plan = bottleneck_plan(b, h, w, cin, planes, stride, u2 is not None, u4 is not None, min_tiles4, fuse_conv3)
    res = x if wd is None else conv_bn_act(x, wd, sd, td, stride=stride, relu=False)
    hmid = conv_bn_act(x, w1, s1, t1, stride=stride, relu=True, out_kblocked=bool(plan & 3))
    if plan & 4:
        return conv3x3_winograd4_conv3(hmid, u4, s2, t2, w3, s3, t3, res)
    if plan & 1:
        h2 = conv3x3_winograd4(hmid, u4, s2, t2, True)
    elif plan & 2:
        h2 = conv3x3_winograd(hmid, u2, s2, t2, True)
    else:
        h2 = conv_bn_act(hmid, w2, s2, t2, stride=1, pad=(1, 1, 1, 1), relu=True)
    return conv_bn_act(h2, w3, s3, t3, stride=1, relu=True, residual=res)


def bottleneck_forward(x, w1, s1, t1, w2, s2, t2, w3, s3, t3, wd, sd, td, stride: int):
    """torch.ops.maskrcnn.bottleneck_forward — Bottleneck.forward (model.py:190-211) on NHWC: conv1 1x1 (stride) + BN + ReLU,
    SamePad(3,1) + conv2 3x3 + BN + ReLU, conv3 1x1 + BN, + residual (identity or 1x1-stride downsample + BN), ReLU; BN / bias
    are (scale, shift) epilogues. ONE call of the C ABI per block (bottleneck_native): with Winograd enabled (the default) conv2
    runs the F(4x4) / F(2x2) kernels from cached transforms of w2 and the ResNet C2 blocks fuse conv2 + conv3 + residual into
    one launch — exactly what the inference pipeline launches; with MRCNN_WINOGRAD=0 every conv is the exact direct kernel
    (bitwise an fmaf chain). With BOTTLENECK_OP_FUSED (above) the stride-1 identity blocks with planes = 64 run the opt-in
    whole-block kernel instead."""
    from . import modules as _m
    if (BOTTLENECK_OP_FUSED and wd is None and int(stride) == 1 and x.is_cuda and x.dim() == 4
            and tuple(w2.shape[1:3]) == (3, 3)
            and bottleneck_fused_supported(x.size(1), x.size(2), x.size(3), w1.size(0)) and w3.size(0) == x.size(3)):
        return bottleneck_fused(x, w1, s1, t1, _cached_winograd_weights(w2), s2, t2, w3, s3, t3)
    wino = bool(_m.WINOGRAD) and x.is_cuda and tuple(w2.shape[1:3]) == (3, 3) and w2.size(3) % 8 == 0
    u2 = _cached_winograd_weights(w2) if wino else None
    u4 = _cached_winograd4_weights(w2) if (wino and _m.WINOGRAD4 and _m.WINOGRAD4_TRUNK and w2.size(0) % 64 == 0) else None
    return bottleneck_native(x, w1, s1, t1, w2, u2, u4, s2, t2, w3, s3, t3, wd, sd, td, int(stride), _m.WINOGRAD4_MIN_TILES,
                             _m.FUSED_CONV3)


_LIB.define("bottleneck_forward(Tensor x, Tensor w1, Tensor s1, Tensor t1, Tensor w2, Tensor s2, "
            "Tensor t2, Tensor w3, Tensor s3, Tensor t3, Tensor? wd, Tensor? sd, Tensor? td, "
            "int stride) -> Tensor")
_LIB.impl("bottleneck_forward", bottleneck_forward, "CUDA")
_LIB.impl("bottleneck_forward", lambda x, *a: _need_gpu(x), "CPU")

__all__ += ["conv_bn_act", "conv_bn_act_f16mfma", "split_f16", "same_pad", "maxpool", "nchw_to_nhwc", "nhwc_to_nchw", "bottleneck_forward",
            "bottleneck_fused", "bottleneck_fused_supported", "bottleneck_native", "bottleneck_plan"]


class HeadSums:
    """Output of conv3x3_winograd_heads for one pyramid level: the two k halves of the 1x1-head sums (without bias) in
    position-major pixel order, [2, rows, 32] fp32, for a [B, H, W] level. Consumed by rpn_scores_deltas."""

    def __init__(self, part: torch.Tensor, batch: int, height: int, width: int, tile_mode: int = 1):
        self.part, self.batch, self.height, self.width, self.tile_mode = part, batch, height, width, tile_mode

    def to_nhwc(self, bias: torch.Tensor) -> torch.Tensor:
        """[B,H,W,18] = (half 0 + half 1) + bias, in image order (tests / debugging; the pipeline never needs it)."""
        b, h, w = self.batch, self.height, self.width
        if self.tile_mode == 4:                                            # two planes in pixel order (conv_f16_pipelined_heads)
            return ((self.part[0, :, :18] + self.part[1, :, :18]) + bias).view(b, h, w, 18).contiguous()
        th, tw = h // 2, w // 2
        if self.tile_mode == 1:
            t = b * th * tw
            v = (self.part[0, :t * 4, :18] + self.part[1, :t * 4, :18]) + bias
            return v.view(b, th, tw, 2, 2, 18).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, 18).contiguous()
        if self.tile_mode == 3:                                            # F(4x4): rows (b, tyb, txb, py4, px8, i4, j4)
            tyb, txb = -(-(h // 4) // 4), -(-(w // 4) // 8)
            v = self.part[:, :18] + bias
            v = v.view(b, tyb, txb, 4, 8, 4, 4, 18).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(b, tyb * 16, txb * 32, 18)
            return v[:, :h, :w].contiguous()
        tyb, txb = -(-th // 8), -(-tw // 8)
        v = (self.part[0, :, :18] + self.part[1, :, :18]) + bias          # rows: (b, tyb, txb, py, px, a, c)
        v = v.view(b, tyb, txb, 8, 8, 2, 2, 18).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(b, tyb * 16, txb * 16, 18)
        return v[:, :h, :w].contiguous()


@_on_device
def conv3x3_winograd_heads(x_kblocked: torch.Tensor, u: torch.Tensor, scale, shift, w_head32: torch.Tensor,
                           relu: bool = True, algo_cin=None, tile_mode: int | None = None) -> HeadSums:
    """RPN conv_shared + both 1x1 heads in one launch (model.py:605-607,624-641): relu(conv3x3_same(x)*scale + shift)
    stays on chip and is multiplied by w_head32 [32, Cout] (rows 0-17: conv_class then conv_bbox weights).
    x_kblocked [Cin/8,B,H,W,8]; u from winograd_weights. → HeadSums."""
    _need_gpu(x_kblocked, u, scale, shift, w_head32)
    assert x_kblocked.dim() == 5 and x_kblocked.is_contiguous() and x_kblocked.dtype == torch.float32
    g, b, h, w, _ = x_kblocked.shape
    cin, cout = g * 8, u.size(1)
    assert u.is_contiguous() and u.size(2) == cin and w_head32.is_contiguous() and tuple(w_head32.shape) == (32, cout)
    mode = int(lib.mrcnn_conv3x3_winograd_heads_tile_mode(h, w)) if tile_mode is None else int(tile_mode)
    if mode != 2 and not HAVE_ABLATIONS:
        raise RuntimeError("conv3x3_winograd_heads: maps of at least 16 x 16 pixels (8 x 8 tile positions) and the spatial-tile "
                           "kernel are required; the linear-tile heads variant is an MRCNN_ABLATIONS build")
    rows = int(lib.mrcnn_conv3x3_winograd_heads_rows(b, h, w, mode))
    part = torch.empty(2, rows, 32, dtype=torch.float32, device=x_kblocked.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_conv3x3_winograd_heads_f32(x_kblocked.data_ptr(), b, h, w, cin, u.data_ptr(), cout, _ptr(scale),
                                               _ptr(shift), 1 if relu else 0, w_head32.data_ptr(), mode, part.data_ptr(),
                                               _stream()))
    if prof is not None:
        e1.record()
        m, k = b * h * w, 9 * (algo_cin or cin)
        algo = 2.0 * m * cout * (k + 18)                      # the 3x3 conv + both 1x1 heads (18 channels)
        executed = 2.0 * m * cout * (k / 2.25 + 32)           # Winograd multiplies + the head MFMAs on 32 padded columns
        prof.append((e0, e1, algo, (m, cout, k), 4.0 * (m * cin + 2 * part.numel() / 2 + cout * k),
                     "winograd_spatial" if mode == 2 else "winograd", executed))
    return HeadSums(part, b, h, w, mode)


@_on_device
def rpn_scores_deltas(heads, head_bias: torch.Tensor | None = None):
    """heads: 5 per-level entries (P2..P6), each a contiguous fp32 NHWC tensor [B,H_l,W_l,18] (fused RPN head outputs)
    or a HeadSums (conv3x3_winograd_heads; needs head_bias [18]) → (fg scores [B,A], deltas [B,A,4]) in the
    reference's anchor order. One launch."""
    assert len(heads) == 5
    tens = [h.part if isinstance(h, HeadSums) else h for h in heads]
    _need_gpu(*tens, head_bias)
    b = heads[0].batch if isinstance(heads[0], HeadSums) else heads[0].size(0)
    hs, ws, modes = [], [], []
    for h in heads:
        if isinstance(h, HeadSums):
            assert h.batch == b and h.part.is_contiguous() and head_bias is not None and head_bias.numel() == 18
            hs.append(h.height); ws.append(h.width); modes.append(h.tile_mode)
        else:
            assert h.is_contiguous() and h.dtype == torch.float32 and h.size(0) == b and h.size(3) == 18
            hs.append(h.size(1)); ws.append(h.size(2)); modes.append(0)
    a = 3 * sum(hh * ww for hh, ww in zip(hs, ws))
    scores = torch.empty(b, a, dtype=torch.float32, device=tens[0].device)
    deltas = torch.empty(b, a, 4, dtype=torch.float32, device=tens[0].device)
    ptrs = (c_vp * 5)(*[t.data_ptr() for t in tens])
    check(lib.mrcnn_rpn_scores_deltas_v2_f32(ptrs, (c_i32 * 5)(*hs), (c_i32 * 5)(*ws), (c_i32 * 5)(*modes),
                                             _ptr(head_bias), b, scores.data_ptr(), deltas.data_ptr(), _stream()))
    return scores, deltas


@_on_device
def proposal_decode(anchors, deltas, order, top_scores, std_dev, image_height, image_width):
    """anchors [A,4], deltas [B,A,4], order int64 [B,K], top_scores [B,K] → dets [B,K,5]: refined (data.py:124),
    clipped (data.py:86) boxes + score. One launch."""
    _need_gpu(anchors, deltas, order, top_scores)
    assert anchors.is_contiguous() and deltas.is_contiguous() and order.is_contiguous() and top_scores.is_contiguous()
    assert order.dtype == torch.int64 and deltas.dtype == torch.float32
    b, k = order.shape
    dets = torch.empty(b, k, 5, dtype=torch.float32, device=deltas.device)
    check(lib.mrcnn_proposal_decode_f32(anchors.data_ptr(), deltas.data_ptr(), order.data_ptr(),
                                        top_scores.data_ptr(), b, anchors.size(0), k,
                                        (c_f32 * 4)(*[float(v) for v in std_dev]), float(image_height),
                                        float(image_width), dets.data_ptr(), _stream()))
    return dets


__all__ += ["rpn_scores_deltas", "proposal_decode", "conv3x3_winograd_heads", "HeadSums"]


@_on_device
def detection_decode(logits, bbox, rois, roi_counts, windows, std_dev, image_height, image_width,
                     min_confidence: float = 0.0):
    """First half of mrn_refine (model.py:1405-1443), one launch. logits [B*P,C] (row-strided view allowed),
    bbox [B*P,C,4] (row-strided view allowed), rois [B,P,4], roi_counts int32 [B], windows [B,4] →
    (dets [B,P,5], nms_class_ids int32 [B,P], class_ids int64 [B,P])."""
    _need_gpu(logits, bbox, rois, roi_counts, windows)
    b, p, _ = rois.shape
    c = logits.size(1)
    assert logits.stride(1) == 1 and bbox.stride(2) == 1 and bbox.stride(1) == 4 and rois.is_contiguous()
    assert roi_counts.dtype == torch.int32 and windows.dtype == torch.float32 and windows.is_contiguous()
    dets = torch.empty(b, p, 5, dtype=torch.float32, device=rois.device)
    nms_cls = torch.empty(b, p, dtype=torch.int32, device=rois.device)
    cls = torch.empty(b, p, dtype=torch.int64, device=rois.device)
    check(lib.mrcnn_detection_decode_f32(logits.data_ptr(), logits.stride(0), bbox.data_ptr(), bbox.stride(0),
                                         rois.data_ptr(), roi_counts.data_ptr(), windows.data_ptr(), b, p, c,
                                         (c_f32 * 4)(*[float(v) for v in std_dev]), float(image_height),
                                         float(image_width), float(min_confidence), dets.data_ptr(),
                                         nms_cls.data_ptr(), cls.data_ptr(), _stream()))
    return dets, nms_cls, cls


__all__ += ["detection_decode"]


@_on_device
def topk_desc(scores: torch.Tensor, k: int):
    """scores fp32 [B,N] → (top [B,k], order int64 [B,k]): descending, ties by ascending index (model.py:1345-1350)."""
    _need_gpu(scores)
    assert scores.dtype == torch.float32 and scores.dim() == 2 and scores.is_contiguous()
    b, n = scores.shape
    top = torch.empty(b, k, dtype=torch.float32, device=scores.device)
    order = torch.empty(b, k, dtype=torch.int64, device=scores.device)
    nbytes = int(lib.mrcnn_topk_workspace_bytes(b))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=scores.device)
    check(lib.mrcnn_topk_desc_f32(scores.data_ptr(), b, n, k, top.data_ptr(), order.data_ptr(), ws.data_ptr(), nbytes,
                                  _stream()))
    return top, order


@_on_device
def proposal_select(dets, keep, keep_counts, proposal_count: int, image_height, image_width):
    """dets [B,K,5], NMS keep int64 [B,K] + counts int32 [B] → (rois [B,P,4] normalised, zero-padded; counts int32 [B])
    — keep[:proposal_count], gather, normalise (model.py:1366-1374) in one launch."""
    _need_gpu(dets, keep, keep_counts)
    assert dets.is_contiguous() and keep.is_contiguous() and keep.dtype == torch.int64
    assert keep_counts.dtype == torch.int32 and keep_counts.is_contiguous()
    b, k, _ = dets.shape
    rois = torch.empty(b, proposal_count, 4, dtype=torch.float32, device=dets.device)
    counts = torch.empty(b, dtype=torch.int32, device=dets.device)
    check(lib.mrcnn_proposal_select_f32(dets.data_ptr(), keep.data_ptr(), keep_counts.data_ptr(), b, k,
                                        proposal_count, float(image_height), float(image_width), rois.data_ptr(),
                                        counts.data_ptr(), _stream()))
    return rois, counts


@_on_device
def detection_select(dets, nms_class_ids, class_ids, keep, keep_counts, max_instances: int, image_height,
                     image_width):
    """Tail of mrn_refine (model.py:1475-1487) in one launch → (class_ids int64 [B,D], scores [B,D], boxes [B,D,4]
    pixels, rois [B,D,4] normalised for the mask head, counts int32 [B]); unused slots are zero."""
    _need_gpu(dets, nms_class_ids, class_ids, keep, keep_counts)
    assert dets.is_contiguous() and nms_class_ids.is_contiguous() and class_ids.is_contiguous()
    assert keep.is_contiguous() and keep.dtype == torch.int64 and class_ids.dtype == torch.int64
    assert nms_class_ids.dtype == torch.int32 and keep_counts.dtype == torch.int32
    b, p, _ = dets.shape
    d, dev = max_instances, dets.device
    ids = torch.empty(b, d, dtype=torch.int64, device=dev)
    scores = torch.empty(b, d, dtype=torch.float32, device=dev)
    boxes = torch.empty(b, d, 4, dtype=torch.float32, device=dev)
    rois = torch.empty(b, d, 4, dtype=torch.float32, device=dev)
    counts = torch.empty(b, dtype=torch.int32, device=dev)
    check(lib.mrcnn_detection_select_f32(dets.data_ptr(), nms_class_ids.data_ptr(), class_ids.data_ptr(),
                                         keep.data_ptr(), keep_counts.data_ptr(), b, p, d, float(image_height),
                                         float(image_width), ids.data_ptr(), scores.data_ptr(), boxes.data_ptr(),
                                         rois.data_ptr(), counts.data_ptr(), _stream()))
    return ids, scores, boxes, rois, counts


__all__ += ["topk_desc", "proposal_select", "detection_select"]


@_on_device
def deconv2x2(x: torch.Tensor, w, bias4: torch.Tensor, activation: int = 0, products: int = 0) -> torch.Tensor:
    """2x2 stride-2 transposed conv + bias + activation (Mask.forward's deconv, model.py:864,906-912) as one GEMM
    scattering into [B,2H,2W,Cout]. w: fp32 [4*Cout,1,1,Cin] (products = 0) or the (w_hi, w_lo) fp16 planes of it
    (products = 1 or 3); bias4 [4*Cout]."""
    _need_gpu(x, bias4)
    assert x.is_contiguous() and x.dtype in (torch.float32, torch.float16) and bias4.is_contiguous()
    b, h, wd, cin = x.shape
    w0 = w if products == 0 else w[0]
    cout = w0.size(0) // 4
    assert w0.size(3) == cin and bias4.numel() == 4 * cout
    y = torch.empty(b, 2 * h, 2 * wd, cout, dtype=x.dtype, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if products == 0:
        assert w.dtype == torch.float32 and w.is_contiguous()
        check(lib.mrcnn_deconv2x2_bias_act_nhwc_f32(x.data_ptr(), b, h, wd, cin, w.data_ptr(), cout,
                                                    bias4.data_ptr(), int(activation), y.data_ptr(), _stream()))
    elif x.dtype == torch.float16:   # fp16 activations: plain-fp16 mode only
        assert products == 1
        check(lib.mrcnn_deconv2x2_bias_act_nhwc_f16io(x.data_ptr(), b, h, wd, cin, w[0].data_ptr(), cout, bias4.data_ptr(),
                                                      int(activation), y.data_ptr(), _stream()))
    else:
        w_hi, w_lo = w
        check(lib.mrcnn_deconv2x2_bias_act_nhwc_f16mfma(x.data_ptr(), b, h, wd, cin, w_hi.data_ptr(), _ptr(w_lo),
                                                        cout, bias4.data_ptr(), int(activation), int(products),
                                                        y.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m = b * h * wd
        prof.append((e0, e1, 2.0 * m * cin * 4 * cout, (m, 4 * cout, cin),
                     x.element_size() * (x.numel() + y.numel()) + (4 if products == 0 else 2 * (2 if products == 3 else 1)) * w0.numel(),
                     "direct" if products == 0 else "f16"))
    return y


__all__ += ["deconv2x2"]


HAVE_ABLATIONS = hasattr(lib, "mrcnn_rpn_level_fused_f32")   # an MRCNN_ABLATIONS build of the library is loaded


@_on_device
def rpn_level_fused(x, w_shared, b_shared, w_head32, b_head, head_n: int = 18) -> torch.Tensor:
    """RPN.forward on one level (model.py:609-649) with the shared 512-channel activation kept on chip:
    x [B,H,W,Cin] NHWC → [B,H,W,head_n] (class logits then box deltas). fp32 MFMA path, direct kernel. MRCNN_ABLATIONS
    builds only (the default library fuses the heads into the Winograd kernels instead)."""
    if not HAVE_ABLATIONS:
        raise RuntimeError("rpn_level_fused: built only with MRCNN_ABLATIONS=1 python maskrcnn_amd/build.py")
    _need_gpu(x, w_shared, b_shared, w_head32, b_head)
    assert x.is_contiguous() and w_shared.is_contiguous() and w_head32.is_contiguous()
    b, h, wd, cin = x.shape
    cout = w_shared.size(0)
    assert tuple(w_shared.shape[1:]) == (3, 3, cin) and tuple(w_head32.shape) == (32, cout)
    ws_bytes = int(lib.mrcnn_rpn_level_workspace_bytes(b, h, wd, cout, head_n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    y = torch.empty(b, h, wd, head_n, dtype=torch.float32, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_rpn_level_fused_f32(x.data_ptr(), b, h, wd, cin, w_shared.data_ptr(), cout, _ptr(b_shared),
                                        w_head32.data_ptr(), _ptr(b_head), int(head_n), ws.data_ptr(), ws_bytes,
                                        y.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m = b * h * wd
        flops = 2.0 * m * (9 * cin * cout + cout * head_n)  # shared 3x3 conv + both 1x1 heads
        prof.append((e0, e1, flops, (m, cout, 9 * cin),
                     4 * (x.numel() + w_shared.numel() + cout * head_n + y.numel()), "rpn_fused"))
    return y


__all__ += ["rpn_level_fused"]


# --------------------------------------------------------------------------------------------------
# image pre-/post-processing (Pillow-exact 8-bit bilinear resample)
# --------------------------------------------------------------------------------------------------
@_on_device
def resize_bilinear_u8(image: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """uint8 [H,W], [H,W,C] (C <= 4) or a batch [N,H,W] of single-channel images (rows contiguous; image and row
    strides free, so a cropped view needs no copy) → uint8 of the same rank at out_h x out_w; every image
    == Image.fromarray(a).resize((out_w, out_h), Image.BILINEAR)."""
    _need_gpu(image)
    if image.dtype != torch.uint8 or image.dim() not in (2, 3):
        raise RuntimeError(f"resize_bilinear_u8: expected a uint8 2-d or 3-d tensor, got {image.dtype} {tuple(image.shape)}")
    a = image
    if a.dim() == 3 and a.size(2) <= 4 and a.stride(2) == 1 and a.stride(1) == a.size(2):   # [H,W,C] interleaved
        n, h, w, c = 1, a.size(0), a.size(1), a.size(2)
        image_stride, row_stride = 0, a.stride(0)
        out_shape = (out_h, out_w, c)
    elif a.dim() == 2:
        if a.stride(1) != 1:
            a = a.contiguous()
        n, h, w, c = 1, a.size(0), a.size(1), 1
        image_stride, row_stride = 0, a.stride(0)
        out_shape = (out_h, out_w)
    else:                                                                                     # [N,H,W]
        if a.stride(2) != 1:
            a = a.contiguous()
        n, h, w, c = a.size(0), a.size(1), a.size(2), 1
        image_stride, row_stride = a.stride(0), a.stride(1)
        out_shape = (n, out_h, out_w)
    if out_h < 1 or out_w < 1:
        raise ValueError("height and width must be > 0")       # PIL's message for an empty size
    out = torch.empty(out_shape, dtype=torch.uint8, device=a.device)
    if n == 0:
        return out
    nbytes = int(lib.mrcnn_resize_u8_workspace_bytes(n, h, w, c, out_h, out_w))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=a.device)
    check(lib.mrcnn_resize_bilinear_u8(a.data_ptr(), n, h, w, c, image_stride, row_stride, out.data_ptr(), out_h,
                                       out_w, ws.data_ptr(), nbytes, _stream()))
    return out


@_on_device
def mold_image_u8(image: torch.Tensor, new_h: int, new_w: int, top: int, left: int, out: torch.Tensor,
                  mean_pixel) -> None:
    """uint8 RGB [h,w,3] → out fp32 [3,H,W] (one slot of the batch): resize to new_h x new_w, paste at (top,left) of a
    zero canvas, subtract mean_pixel in double, HWC → CHW. One call replaces utils.py:72-88 + model.py:1754,1108."""
    _need_gpu(image, out)
    if image.dtype != torch.uint8 or image.dim() != 3 or image.size(2) != 3:
        raise RuntimeError(f"mold_image_u8: expected uint8 [h,w,3], got {image.dtype} {tuple(image.shape)}")
    assert out.dtype == torch.float32 and out.dim() == 3 and out.size(0) == 3 and out.is_contiguous()
    a = image.contiguous()
    h, w = a.shape[:2]
    nbytes, ws = 0, None
    if (new_h, new_w) != (h, w):
        nbytes = int(lib.mrcnn_resize_u8_workspace_bytes(1, h, w, 3, new_h, new_w))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=a.device)
    mean = (ctypes.c_double * 3)(*[float(m) for m in mean_pixel])
    check(lib.mrcnn_mold_image_u8(a.data_ptr(), h, w, new_h, new_w, top, left, out.size(1), out.size(2), mean,
                                  out.data_ptr(), _ptr(ws), nbytes, _stream()))


@_on_device
def mold_images_u8(images: torch.Tensor, new_h: int, new_w: int, top: int, left: int, out: torch.Tensor, mean_pixel) -> None:
    """n RGB images of ONE size, uint8 [n,h,w,3] contiguous → out fp32 [n,3,H,W]: mold_image_u8 for a whole batch in one set of
    launches (coefficient tables, horizontal pass, vertical + mold pass)."""
    _need_gpu(images, out)
    if images.dtype != torch.uint8 or images.dim() != 4 or images.size(3) != 3 or not images.is_contiguous():
        raise RuntimeError(f"mold_images_u8: expected contiguous uint8 [n,h,w,3], got {images.dtype} {tuple(images.shape)}")
    n, h, w = images.shape[:3]
    assert out.dtype == torch.float32 and out.dim() == 4 and out.size(0) == n and out.size(1) == 3 and out.is_contiguous()
    nbytes, ws = 0, None
    if (new_h, new_w) != (h, w):
        nbytes = int(lib.mrcnn_resize_u8_workspace_bytes(n, h, w, 3, new_h, new_w))
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=images.device)
    mean = (ctypes.c_double * 3)(*[float(m) for m in mean_pixel])
    check(lib.mrcnn_mold_images_u8(images.data_ptr(), n, h * w * 3, h, w, new_h, new_w, top, left, out.size(2), out.size(3), mean,
                                   out.data_ptr(), _ptr(ws), nbytes, _stream()))


@_on_device
def paste_masks(masks: torch.Tensor, class_ids: torch.Tensor, boxes: torch.Tensor, height: int, width: int,
                channels_last: bool, as_l8: bool = False) -> torch.Tensor:
    """datalib.full_masks (data.py:287-314) for N detections in one launch → bool [N,height,width] (as_l8: the same
    mask as a uint8 0/255 'L' image, what decode_masks converts it to before resizing, data.py:271).
    masks fp32 [N,C,mh,mw] (channels_last=False, the reference layout) or [N,mh,mw,C] (True), any strides;
    class_ids int64 [N]; boxes fp32 [N,4] pixel (y1,x1,y2,x2)."""
    _need_gpu(masks, class_ids, boxes)
    if masks.dtype != torch.float32 or masks.dim() != 4:
        raise RuntimeError(f"paste_masks: expected fp32 4-d masks, got {masks.dtype} {tuple(masks.shape)}")
    n = masks.size(0)
    if channels_last:
        mh, mw, c = masks.shape[1:]
        sn, sy, sx, sc = masks.stride()
    else:
        c, mh, mw = masks.shape[1:]
        sn, sc, sy, sx = masks.stride()
    class_ids = class_ids.to(torch.int64).contiguous()
    boxes = boxes.to(torch.float32).contiguous()
    assert class_ids.numel() == n and tuple(boxes.shape) == (n, 4)
    out = torch.empty(n, height, width, dtype=torch.uint8, device=masks.device)
    check(lib.mrcnn_paste_masks_u8(masks.data_ptr(), sn, sy, sx, sc, n, mh, mw, c, class_ids.data_ptr(),
                                   boxes.data_ptr(), height, width, 255 if as_l8 else 1, out.data_ptr(), _stream()))
    return out if as_l8 else out.view(torch.bool)


# --------------------------------------------------------------------------------------------------
# Winograd F(2x2,3x3) 3x3 stride-1 SAME conv (csrc/conv_wino.hip)
# --------------------------------------------------------------------------------------------------
def winograd_set_spatial(on: int) -> None:
    """Process-wide tuning switch: 1 = spatial-tile Winograd kernel on large maps (default), 0 = linear-tile kernel
    everywhere, -1 = default / MRCNN_WINO_SPATIAL. Results are bit-identical either way."""
    check(lib.mrcnn_winograd_set_spatial(int(on)))


@_on_device
def winograd_weights(w_ohwi: torch.Tensor) -> torch.Tensor:
    """[Cout,3,3,Cin] fp32 → the transformed filter G g G^T (evaluated in double), 16*Cout*Cin floats in the
    kernel's k-blocked order [Cin/8][16][Cout][8] (returned with the logical shape [16,Cout,Cin])."""
    _need_gpu(w_ohwi)
    assert w_ohwi.dtype == torch.float32 and w_ohwi.is_contiguous() and tuple(w_ohwi.shape[1:3]) == (3, 3)
    cout, cin = w_ohwi.size(0), w_ohwi.size(3)
    u = torch.empty(16, cout, cin, dtype=torch.float32, device=w_ohwi.device)
    check(lib.mrcnn_winograd_weights_f32(w_ohwi.data_ptr(), cout, cin, u.data_ptr(), _stream()))
    return u


@_on_device
def nhwc_to_kblocked(x: torch.Tensor) -> torch.Tensor:
    """[B,H,W,C] → [C/8,B,H,W,8]."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.size(3) % 8 == 0
    b, h, w, c = x.shape
    y = torch.empty(c // 8, b, h, w, 8, dtype=torch.float32, device=x.device)
    check(lib.mrcnn_nhwc_to_kblocked_f32(x.data_ptr(), b * h * w, c, y.data_ptr(), _stream()))
    return y


@_on_device
def conv3x3_winograd(x: torch.Tensor, u: torch.Tensor, scale, shift, relu: bool = False, algo_cin=None,
                     out: str = "nhwc"):
    """relu(conv3x3_same(x) * scale + shift) with u from winograd_weights.
    x: NHWC fp32 [B,H,W,Cin] or k-blocked [Cin/8,B,H,W,8] (H, W even; Cin % 8 == 0).
    out: "nhwc" → [B,H,W,Cout]; "kblocked" → [Cout/8,B,H,W,8] (feeds the next Winograd conv without a transposition
    pass); "both" → (nhwc, kblocked)."""
    _need_gpu(x, u, scale, shift)
    assert x.dtype == torch.float32 and x.is_contiguous() and u.is_contiguous() and out in ("nhwc", "kblocked", "both")
    if x.dim() == 5:
        layout = 1
        g, b, h, w, _ = x.shape
        cin = g * 8
    else:
        layout = 0
        b, h, w, cin = x.shape
    cout = u.size(1)
    assert u.size(2) == cin
    y = torch.empty(b, h, w, cout, dtype=torch.float32, device=x.device) if out != "kblocked" else None
    yk = torch.empty(cout // 8, b, h, w, 8, dtype=torch.float32, device=x.device) if out != "nhwc" else None
    nbytes, ws = 0, None
    if layout == 0:
        nbytes = int(lib.mrcnn_conv3x3_winograd_workspace_bytes(b, h, w, cin))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_conv3x3_winograd_f32(x.data_ptr(), layout, b, h, w, cin, u.data_ptr(), cout, _ptr(scale),
                                         _ptr(shift), 1 if relu else 0, _ptr(y), _ptr(yk), _ptr(ws), nbytes, _stream()))
    if prof is not None:
        e1.record()
        m, k = b * h * w, 9 * (algo_cin or cin)
        # FLOPs are the algorithmic ones of the convolution (2*M*N*K), as for the direct kernel — not the reduced
        # multiply count Winograd actually executes
        prof.append((e0, e1, 2.0 * m * cout * k, (m, cout, k),
                     4.0 * (m * cin + m * cout * (2 if out == "both" else 1) + cout * k),
                     "winograd_spatial" if int(lib.mrcnn_conv3x3_winograd_heads_tile_mode(h, w)) == 2 else "winograd"))
    return y if out == "nhwc" else yk if out == "kblocked" else (y, yk)


# --------------------------------------------------------------------------------------------------
# Winograd F(4x4,3x3) 3x3 stride-1 SAME conv (csrc/conv_wino4.hip): maps with H % 4 == W % 4 == 0, Cout % 64 == 0
# --------------------------------------------------------------------------------------------------
def conv3x3_winograd4_supported(h: int, w: int, cin: int, cout: int, batch: int = 1) -> bool:
    """Shapes the F(4x4) kernel takes — its 32-bit offset limits (batch * h * w * channels < 2^30) included."""
    return bool(lib.mrcnn_conv3x3_winograd4_supported(int(batch), int(h), int(w), int(cin), int(cout)))


@_on_device
def winograd4_weights(w_ohwi: torch.Tensor) -> torch.Tensor:
    """[Cout,3,3,Cin] fp32 → G g G^T (6 x 6, evaluated in double) in the kernel's order [Cin/4,36,2,Cout,2]."""
    _need_gpu(w_ohwi)
    assert w_ohwi.dtype == torch.float32 and w_ohwi.is_contiguous() and tuple(w_ohwi.shape[1:3]) == (3, 3)
    cout, cin = w_ohwi.size(0), w_ohwi.size(3)
    assert cin % 4 == 0
    u = torch.empty(cin // 4, 36, 2, cout, 2, dtype=torch.float32, device=w_ohwi.device)
    check(lib.mrcnn_winograd4_weights_f32(w_ohwi.data_ptr(), cout, cin, u.data_ptr(), _stream()))
    return u


@_on_device
def conv3x3_winograd4(x_kblocked: torch.Tensor, u4: torch.Tensor, scale, shift, relu: bool = False, algo_cin=None,
                      out: str = "nhwc"):
    """relu(conv3x3_same(x) * scale + shift) with u4 from winograd4_weights; x k-blocked [Cin/8,B,H,W,8].
    out: "nhwc" → [B,H,W,Cout]; "kblocked" → [Cout/8,B,H,W,8]; "both" → (nhwc, kblocked)."""
    _need_gpu(x_kblocked, u4, scale, shift)
    assert x_kblocked.dim() == 5 and x_kblocked.dtype == torch.float32 and x_kblocked.is_contiguous()
    assert u4.is_contiguous() and out in ("nhwc", "kblocked", "both")
    g, b, h, w, _ = x_kblocked.shape
    cin, cout = g * 8, u4.size(3)
    assert u4.size(0) * 4 == cin and conv3x3_winograd4_supported(h, w, cin, cout, b), (b, h, w, cin, cout)
    y = torch.empty(b, h, w, cout, dtype=torch.float32, device=x_kblocked.device) if out != "kblocked" else None
    yk = torch.empty(cout // 8, b, h, w, 8, dtype=torch.float32, device=x_kblocked.device) if out != "nhwc" else None
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_conv3x3_winograd4_f32(x_kblocked.data_ptr(), b, h, w, cin, u4.data_ptr(), cout, _ptr(scale),
                                          _ptr(shift), 1 if relu else 0, _ptr(y), _ptr(yk), _stream()))
    if prof is not None:
        e1.record()
        m, k = b * h * w, 9 * (algo_cin or cin)
        prof.append((e0, e1, 2.0 * m * cout * k, (m, cout, k),
                     4.0 * (m * cin + m * cout * (2 if out == "both" else 1) + 4 * cout * k), "winograd4",
                     2.0 * m * cout * k / 4.0))
    return y if out == "nhwc" else yk if out == "kblocked" else (y, yk)


@_on_device
def conv3x3_winograd4_conv3(x_kblocked: torch.Tensor, u4: torch.Tensor, scale, shift, w3: torch.Tensor, scale3, shift3,
                            residual: torch.Tensor, algo_cin=None) -> torch.Tensor:
    """conv2 + conv3 of a Bottleneck in one launch (model.py:197-209): relu(conv3x3_same(x) * scale + shift) — 64 channels,
    never written — times the 1x1 expansion w3 [C3,1,1,64], affine (scale3, shift3), + residual [B,H,W,C3], ReLU.
    x k-blocked [Cin/8,B,H,W,8]; u4 = winograd4_weights of the [64,3,3,Cin] conv2 weight. Equals conv3x3_winograd4 followed
    by conv_bn_act(..., residual=...) bit for bit."""
    _need_gpu(x_kblocked, u4, scale, shift, w3, scale3, shift3, residual)
    assert x_kblocked.dim() == 5 and x_kblocked.is_contiguous() and x_kblocked.dtype == torch.float32
    g, b, h, w, _ = x_kblocked.shape
    cin, c3 = g * 8, w3.size(0)
    assert u4.is_contiguous() and u4.size(0) * 4 == cin and u4.size(3) == 64 and conv3x3_winograd4_supported(h, w, cin, 64, b)
    assert w3.is_contiguous() and w3.numel() == c3 * 64 and c3 % 32 == 0
    assert residual.is_contiguous() and tuple(residual.shape) == (b, h, w, c3) and residual.dtype == torch.float32
    y = torch.empty(b, h, w, c3, dtype=torch.float32, device=x_kblocked.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_conv3x3_winograd4_conv3_f32(x_kblocked.data_ptr(), b, h, w, cin, u4.data_ptr(), _ptr(scale), _ptr(shift),
                                                w3.data_ptr(), c3, _ptr(scale3), _ptr(shift3), residual.data_ptr(),
                                                y.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m, k = b * h * w, 9 * (algo_cin or cin)
        algo = 2.0 * m * (64 * k + 64 * c3)
        prof.append((e0, e1, algo, (m, c3, k + 64), 4.0 * (m * cin + 2 * m * c3 + 4 * 64 * k + 64 * c3), "winograd4",
                     2.0 * m * (64 * k / 4.0 + 64 * c3)))
    return y


@_on_device
def conv3x3_winograd4_heads(x_kblocked: torch.Tensor, u4: torch.Tensor, scale, shift, w_head32: torch.Tensor,
                            relu: bool = True, algo_cin=None) -> HeadSums:
    """The RPN level in one launch on the F(4x4) kernel: relu(conv3x3_same(x)*scale+shift) and its two 1x1 heads
    (w_head32 [32, Cout]: rows 0..17 = conv_class then conv_bbox) → HeadSums with tile_mode 3."""
    _need_gpu(x_kblocked, u4, scale, shift, w_head32)
    assert x_kblocked.dim() == 5 and x_kblocked.is_contiguous() and x_kblocked.dtype == torch.float32
    g, b, h, w, _ = x_kblocked.shape
    cin, cout = g * 8, u4.size(3)
    assert u4.is_contiguous() and u4.size(0) * 4 == cin and conv3x3_winograd4_supported(h, w, cin, cout, b)
    assert w_head32.is_contiguous() and tuple(w_head32.shape) == (32, cout)
    rows = int(lib.mrcnn_conv3x3_winograd4_heads_rows(b, h, w))
    part = torch.empty(rows, 32, dtype=torch.float32, device=x_kblocked.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_conv3x3_winograd4_heads_f32(x_kblocked.data_ptr(), b, h, w, cin, u4.data_ptr(), cout, _ptr(scale),
                                                _ptr(shift), 1 if relu else 0, w_head32.data_ptr(), part.data_ptr(),
                                                _stream()))
    if prof is not None:
        e1.record()
        m, k = b * h * w, 9 * (algo_cin or cin)
        prof.append((e0, e1, 2.0 * m * cout * (k + 18), (m, cout, k), 4.0 * (m * cin + part.numel() + 4 * cout * k),
                     "winograd4", 2.0 * m * cout * (k / 4.0 + 32)))
    return HeadSums(part, b, h, w, 3)


@_on_device
def stem_conv(x: torch.Tensor, w: torch.Tensor, scale, shift, relu: bool = True, algo_cin: int | None = None,
              nchw: bool = False, out_f16: bool = False):
    """The ResNet stem: conv 7x7 stride 2 pad 3 + affine + ReLU. x NHWC [B,H,W,4] (RGB + zero channel) — or, with nchw=True,
    the molded image itself [B,3,H,W] (no layout pass beforehand; same result bit for bit) —, w OHWI [64,7,7,4] → [B,H/2,W/2,64]."""
    _need_gpu(x, w, scale, shift)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.size(1 if nchw else 3) == (3 if nchw else 4)
    assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (64, 7, 7, 4)
    if nchw:
        b, _, h, wd = x.shape
    else:
        b, h, wd, _ = x.shape
    assert nchw or not out_f16, "the fp16-output form reads the NCHW image"
    y = torch.empty(b, h // 2, wd // 2, 64, dtype=torch.float16 if out_f16 else torch.float32, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = lib.mrcnn_stem_conv7x7_s2_nchw_f16out if out_f16 else \
        lib.mrcnn_stem_conv7x7_s2_nchw_f32 if nchw else lib.mrcnn_stem_conv7x7_s2_nhwc_f32
    check(fn(x.data_ptr(), b, h, wd, w.data_ptr(), _ptr(scale), _ptr(shift), 1 if relu else 0, y.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m, k = y.numel() // 64, 49 * (algo_cin or 4)
        prof.append((e0, e1, 2.0 * m * 64 * k, (m, 64, k), 4.0 * (x.numel() + w.numel()) + y.numel() * y.element_size(), "stem"))
    return y


@_on_device
def stem_pool_f16(x: torch.Tensor, w: torch.Tensor, scale, shift, algo_cin: int | None = None) -> torch.Tensor:
    """The "f16" mode's stem + max-pool in one launch on the fp16 MFMA (model.py:223-229): x the molded NCHW image [B,3,H,W]
    fp32, w OHWI [64,7,7,4] fp32 → fp16 NHWC [B, ceil(H/4), ceil(W/4), 64]."""
    _need_gpu(x, w, scale, shift)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.size(1) == 3
    assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (64, 7, 7, 4)
    b, _, h, wd = x.shape
    assert h % 4 == 0 and wd % 4 == 0
    poh, pow_ = h // 4, wd // 4
    y = torch.empty(b, poh, pow_, 64, dtype=torch.float16, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_stem_conv7x7_s2_pool_f16(x.data_ptr(), b, h, wd, w.data_ptr(), _ptr(scale), _ptr(shift), y.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m, k = b * (h // 2) * (wd // 2), 49 * (algo_cin or 4)
        prof.append((e0, e1, 2.0 * m * 64 * k, (m, 64, k), 4.0 * (x.numel() + w.numel()) + y.numel() * 2, "stem"))
    return y


@_on_device
def stem_pool_f32(x: torch.Tensor, w: torch.Tensor, scale, shift, algo_cin: int | None = None) -> torch.Tensor:
    """The exact-fp32 stem + max-pool in one launch (model.py:223-229; csrc/stem.hip: stem7x7_s2_pool_f32): x the molded NCHW
    image [B,3,H,W] fp32 (H, W multiples of 4), w OHWI [64,7,7,4] fp32 (channel 3 zero) → fp32 NHWC [B, H/4, W/4, 64]."""
    _need_gpu(x, w, scale, shift)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.size(1) == 3
    assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (64, 7, 7, 4)
    b, _, h, wd = x.shape
    assert h % 4 == 0 and wd % 4 == 0
    assert b * h * wd < (1 << 27), "stem_pool_f32: B*H*W < 2^27 pixels (32-bit byte offsets of the fp32 output)"
    y = torch.empty(b, h // 4, wd // 4, 64, dtype=torch.float32, device=x.device)
    prof = CONV_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.mrcnn_stem_conv7x7_s2_pool_f32(x.data_ptr(), b, h, wd, w.data_ptr(), _ptr(scale), _ptr(shift), y.data_ptr(), _stream()))
    if prof is not None:
        e1.record()
        m, k = b * (h // 2) * (wd // 2), 49 * (algo_cin or 3)
        # executed: 77 MFMAs of K = 2 per 32 x 32 block (k = 22 per filter row) on 15 x 33 conv pixels per 7 x 16 pooled ones,
        # padded to 512 GEMM rows
        tiles = b * -(-(h // 4) // 7) * -(-(wd // 4) // 16)
        prof.append((e0, e1, 2.0 * m * 64 * k, (m, 64, k), 4.0 * (x.numel() + 64 * 147 + y.numel()), "stem",
                     2.0 * tiles * 512 * 64 * 154))
    return y
