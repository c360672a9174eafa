"""ctypes binding of libmaskrcnn_hip.so (the C ABI declared in include/maskrcnn_hip.h).

There is no CPU fallback and no alternative backend: if the library is missing or does not export a
declared symbol, importing this module raises — run `python maskrcnn_amd/build.py` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes
import os
import re

PKG = os.path.dirname(os.path.abspath(__file__))
# MRCNN_LIB: load another build of the same library (an experiment / ablation build made by `build.py --variant NAME`,
# which never overwrites the product file). Same ABI check, same no-fallback rule.
LIB_PATH = os.environ.get("MRCNN_LIB") or os.path.join(PKG, "libmaskrcnn_hip.so")
HEADER = os.path.join(os.path.dirname(PKG), "include", "maskrcnn_hip.h")

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

_SIGS = {
    "mrcnn_abi_version": (ctypes.c_int, []),
    "mrcnn_last_error": (ctypes.c_char_p, []),
    "mrcnn_arch": (ctypes.c_char_p, []),
    "mrcnn_nms_max_boxes": (c_i64, []),
    "mrcnn_nms_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i64]),
    "mrcnn_nms_batched_f32": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp,
                                               c_f32, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mrcnn_nms_general_workspace_bytes": (ctypes.c_size_t, [c_i64, c_i32]),
    "mrcnn_nms_general": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mrcnn_crop_forward_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32,
                                                c_f32, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_crop_backward_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                                                 c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_roi_align_pyramid_nhwc_f32": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.POINTER(c_i32),
                                                          ctypes.POINTER(c_i32), c_i32, c_i32, c_vp,
                                                          c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp,
                                                          c_vp]),
    "mrcnn_roi_align_pyramid_counted_f32": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                                             c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_f32, c_vp,
                                                             c_i32, c_vp, c_vp]),
    "mrcnn_conv_bn_act_rows_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                                                    c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp]),
    "mrcnn_conv_rows_tile_m": (c_i32, [c_i32]),
    "mrcnn_roi_align_pyramid_f32": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                                     c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_i32,
                                                     c_vp, c_vp]),
    "mrcnn_maxpool_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp,
                                           c_i32, c_vp]),
    "mrcnn_conv_bn_act_nhwc_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32,
                                                    c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp,
                                                    c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_conv_bn_act_nhwc_f16mfma": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32,
                                                        c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp,
                                                        c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_conv_bn_act_nhwc_f16io": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32,
                                                      c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32,
                                                      c_vp, c_i32, c_vp]),
    "mrcnn_deconv2x2_bias_act_nhwc_f16io": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32,
                                                             c_vp, c_vp]),
    "mrcnn_conv_f16_pipelined_supported": (ctypes.c_int, [c_i32] * 12),
    "mrcnn_conv_f16_pipelined": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                                                  c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32,
                                                  c_i32, c_vp]),
    "mrcnn_conv_f16_pipelined_heads": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                                                        c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp]),
    "mrcnn_pack_afrags_f16": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_bottleneck_c2_f16_supported": (ctypes.c_int, [c_i32] * 6),
    "mrcnn_bottleneck_c2_f16": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mrcnn_mask_tail_f16_supported": (ctypes.c_int, [c_i32] * 6),
    "mrcnn_mask_tail_f16": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mrcnn_maxpool_nhwc_f16": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                                c_vp, c_vp]),
    "mrcnn_deconv2x2_bias_act_nhwc_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32,
                                                           c_vp, c_vp]),
    "mrcnn_deconv2x2_bias_act_nhwc_f16mfma": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32,
                                                               c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_maxpool_nhwc_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                                c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_rpn_scores_deltas_f32": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), c_i32, c_vp,
                                                     c_vp, c_vp]),
    "mrcnn_rpn_scores_deltas_v2_f32": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                                        ctypes.POINTER(c_i32), c_vp, c_i32, c_vp, c_vp, c_vp]),
    "mrcnn_winograd4_weights_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_conv3x3_winograd4_supported": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "mrcnn_conv3x3_winograd4_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32,
                                                     c_vp, c_vp, c_vp]),
    "mrcnn_conv3x3_winograd4_conv3_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp,
                                                           c_vp, c_vp, c_vp, c_vp]),
    "mrcnn_conv3x3_winograd4_heads_rows": (c_i64, [c_i32, c_i32, c_i32]),
    "mrcnn_conv3x3_winograd4_heads_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32,
                                                           c_vp, c_vp, c_vp]),
    "mrcnn_conv3x3_winograd_heads_rows": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "mrcnn_conv3x3_winograd_heads_tile_mode": (c_i32, [c_i32, c_i32]),
    "mrcnn_conv3x3_winograd_heads_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32,
                                                          c_vp, c_i32, c_vp, c_vp]),
    "mrcnn_proposal_decode_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                                   ctypes.POINTER(c_f32), c_f32, c_f32, c_vp, c_vp]),
    "mrcnn_detection_decode_f32": (ctypes.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                                    ctypes.POINTER(c_f32), c_f32, c_f32, c_f32, c_vp, c_vp, c_vp,
                                                    c_vp]),
    "mrcnn_winograd_set_spatial": (ctypes.c_int, [c_i32]),
    "mrcnn_winograd_weights_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_conv3x3_winograd_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32, c_i32, c_i32]),
    "mrcnn_conv3x3_winograd_nhwc_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32,
                                                         c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mrcnn_conv3x3_winograd_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32,
                                                    c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mrcnn_conv_bn_act_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32,
                                               c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32,
                                               c_vp]),
    "mrcnn_bottleneck_fused_supported": (ctypes.c_int, [c_i32, c_i32, c_i32, c_i32]),
    "mrcnn_bottleneck_fused_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                    c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mrcnn_bottleneck_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32]),
    "mrcnn_bottleneck_plan": (c_i32, [c_i32] * 10),
    "mrcnn_bottleneck_forward_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_vp), c_i32,
                                                      c_i32, c_vp, ctypes.c_size_t, c_vp, c_vp]),
    "mrcnn_nhwc_to_kblocked_f32": (ctypes.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "mrcnn_stem_conv7x7_s2_nhwc_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mrcnn_stem_conv7x7_s2_nchw_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mrcnn_stem_conv7x7_s2_pool_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mrcnn_stem_conv7x7_s2_pool_f16": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mrcnn_stem_conv7x7_s2_nchw_f16out": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mrcnn_topk_workspace_bytes": (ctypes.c_size_t, [c_i32]),
    "mrcnn_topk_desc_f32": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mrcnn_proposal_select_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp,
                                                   c_vp]),
    "mrcnn_detection_select_f32": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_f32,
                                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mrcnn_nchw_to_nhwc_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_nhwc_to_nchw_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mrcnn_resize_u8_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32]),
    "mrcnn_resize_bilinear_u8": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_vp, c_i32, c_i32,
                                                  c_vp, ctypes.c_size_t, c_vp]),
    "mrcnn_mold_image_u8": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                             ctypes.POINTER(ctypes.c_double), c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mrcnn_mold_images_u8": (ctypes.c_int, [c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                              ctypes.POINTER(ctypes.c_double), c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mrcnn_paste_masks_u8": (ctypes.c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp,
                                              c_i32, c_i32, c_i32, c_vp, c_vp]),
}


# Entry points of MRCNN_ABLATIONS builds only (include/maskrcnn_hip_ablations.h): bound when the loaded library has them.
_ABLATION_SIGS = {
    "mrcnn_rpn_level_workspace_bytes": (ctypes.c_size_t, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "mrcnn_rpn_level_fused_f32": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp,
                                                   c_i32, c_vp, ctypes.c_size_t, c_vp, c_vp]),
}


def declared_symbols(header: str = HEADER) -> list[str]:
    """Every function name include/maskrcnn_hip.h declares (used by the CPU export test)."""
    text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mrcnn_[a-z0-9_]+)\s*\(", text)) - {"mrcnn_stream_t"})


def header_abi_version(header: str = HEADER) -> int:
    m = re.search(r"^#define\s+MRCNN_ABI_VERSION\s+(\d+)", open(header).read(), flags=re.M)
    if not m:
        raise ImportError(f"{header}: MRCNN_ABI_VERSION not found")
    return int(m.group(1))


_CTYPE_OF = {"int": ctypes.c_int, "int32_t": c_i32, "int64_t": c_i64, "float": c_f32, "double": ctypes.c_double,
             "size_t": ctypes.c_size_t, "mrcnn_stream_t": c_vp}


def header_prototypes(header: str = HEADER) -> dict:
    """name -> (restype, [argtypes]) as ctypes, parsed from the header's prototypes: pointers to anything are
    c_void_p except the small host-side arrays the bindings pass by ctypes array (const float* const fm[4],
    const int32_t hw[5], const float std_dev[4], const double mean[3]: POINTER(elem))."""
    text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    out = {}
    for res, name, args in re.findall(r"^\s*([A-Za-z_][\w\s\*]*?)\s*\b(mrcnn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text,
                                      flags=re.M):
        res = " ".join(res.split())
        restype = ctypes.c_char_p if res == "const char*" else _CTYPE_OF[res]
        argtypes = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("void", ""):
                continue
            arr = re.match(r"^(?:const\s+)?(\w+)\s*(\*?)\s*(?:const\s+)?\w+\s*\[\d*\]$", a)
            if arr:  # host array parameter
                base = c_vp if arr.group(2) else _CTYPE_OF[arr.group(1)]
                argtypes.append(ctypes.POINTER(base))
            elif "*" in a:
                argtypes.append(c_vp)
            else:
                argtypes.append(_CTYPE_OF[a.replace("const ", "").split()[0]])
        out[name] = (restype, argtypes)
    return out


class MaskrcnnHipError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python maskrcnn_amd/build.py` (needs hipcc; cross-compiles for gfx950 without a GPU). "
            "maskrcnn_amd has no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    # a stale .so with the same symbol names but older argument lists would be called with mismatched ctypes
    # arguments (memory corruption / GPU fault): MRCNN_ABI_VERSION is bumped on every signature change
    for name, (res, args) in _ABLATION_SIGS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    built, want = int(lib.mrcnn_abi_version()), header_abi_version()
    if built != want:
        raise ImportError(f"{LIB_PATH} was built for ABI version {built}, include/maskrcnn_hip.h declares {want}: "
                          "rebuild it (python maskrcnn_amd/build.py --force)")
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != 0:
        raise MaskrcnnHipError(f"libmaskrcnn_hip error {rc}: {lib.mrcnn_last_error().decode()}")
