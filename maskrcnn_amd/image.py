"""Image pre-/post-processing around the hot path, on the GPU.

Mirrors the reference functions either side of MaskRCNN.predict:
    resize_image   utils.py:42-90      aspect-preserving resize + centre zero padding (scipy.misc.imresize → PIL)
    mold_image     model.py:1750-1754  float32(image) - MEAN_PIXEL; detect() then transposes to [1,3,H,W] (:1108-1110)
    full_masks     data.py:287-314     28x28 masks → PIL resize to the box → paste → > 127
    decode_boxes   data.py:331-343     boxes back to the original image's frame
    decode_masks   data.py:264-284     masks back to the original image's size (CenterCrop + PIL resize)
The host part below is the scalar bookkeeping (scale, sizes, pads — Python floats and round(), exactly the reference's
expressions); every pixel is produced by libmaskrcnn_hip.so (csrc/image.hip), bit-identical to Pillow's resample.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .config import InferenceConfig


def resize_plan(h: int, w: int, min_dim=None, max_dim=None, padding=False):
    """The arithmetic of utils.resize_image (utils.py:56-88) without touching pixels.
    → (new_h, new_w, window (y1,x1,y2,x2), scale, padding [(top,bottom),(left,right),(0,0)] or False)."""
    scale = 1
    if min_dim:
        scale = max(1, min_dim / min(h, w))            # :62-64 scale up but not down
    if max_dim:
        image_max = max(h, w)
        if round(image_max * scale) > max_dim:         # :67-69
            scale = max_dim / image_max
    new_h, new_w = (round(h * scale), round(w * scale)) if scale != 1 else (h, w)   # :72-74
    window = (0, 0, new_h, new_w)
    if padding:
        top_pad = (max_dim - new_h) // 2               # :79-84
        bottom_pad = max_dim - new_h - top_pad
        left_pad = (max_dim - new_w) // 2
        right_pad = max_dim - new_w - left_pad
        padding = [(top_pad, bottom_pad), (left_pad, right_pad), (0, 0)]
        window = (top_pad, left_pad, new_h + top_pad, new_w + left_pad)
    return new_h, new_w, window, scale, padding


def _to_device_u8(image, device) -> torch.Tensor:
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(np.ascontiguousarray(image))
    if image.dtype != torch.uint8 or image.dim() != 3 or image.size(2) != 3:
        raise RuntimeError(f"expected an RGB uint8 [h,w,3] image, got {image.dtype} {tuple(image.shape)}")
    return image.to(device, non_blocking=True)


def mold_inputs(images, cfg: InferenceConfig, device="cuda:0"):
    """A list of RGB uint8 [h,w,3] images (numpy or torch, any sizes) → molded fp32 [B,3,H,W] on the device, int
    windows [B,4], and per-image (scale, padding, original (h,w)) — detect()'s pre-processing (model.py:1097-1110)
    for a batch. Needs a square canvas, like the reference (IMAGE_MAX_DIM x IMAGE_MAX_DIM). Images of one size are molded
    TOGETHER (one host-to-device copy, one set of launches per size: ops.mold_images_u8), in whatever order they come."""
    if cfg.image_height != cfg.image_width or cfg.image_height != cfg.image_max_dim:
        raise RuntimeError("mold_inputs: resize_image pads to IMAGE_MAX_DIM x IMAGE_MAX_DIM; configure a square canvas "
                           "of image_max_dim, or mold images yourself")
    device = torch.device(device)
    out = torch.empty(len(images), 3, cfg.image_height, cfg.image_width, dtype=torch.float32, device=device)
    windows, metas, groups = [], [], {}
    for i, image in enumerate(images):
        if tuple(image.shape[2:]) != (3,) or len(image.shape) != 3 or image.dtype not in (torch.uint8, np.uint8):
            raise RuntimeError(f"expected an RGB uint8 [h,w,3] image, got {image.dtype} {tuple(image.shape)}")
        h, w = int(image.shape[0]), int(image.shape[1])
        new_h, new_w, window, scale, padding = resize_plan(h, w, cfg.image_min_dim, cfg.image_max_dim, True)
        windows.append(window)
        metas.append((scale, padding, (h, w)))
        groups.setdefault((h, w), []).append(i)
    for (h, w), idx in groups.items():
        new_h, new_w, window, _, _ = resize_plan(h, w, cfg.image_min_dim, cfg.image_max_dim, True)
        members = [images[i] for i in idx]
        if all(isinstance(m, np.ndarray) for m in members):
            stack = torch.from_numpy(np.stack(members)).to(device, non_blocking=True)          # one copy for the group
        else:
            stack = torch.stack([_to_device_u8(m, device) for m in members])
        whole = idx == list(range(idx[0], idx[0] + len(idx)))                                   # a contiguous run of the batch
        dst = out[idx[0]:idx[0] + len(idx)] if whole else torch.empty(len(idx), 3, cfg.image_height, cfg.image_width,
                                                                      dtype=torch.float32, device=device)
        ops.mold_images_u8(stack.contiguous(), new_h, new_w, window[0], window[1], dst, cfg.mean_pixel)
        if not whole:
            out[torch.tensor(idx, device=device)] = dst
    return out, torch.tensor(windows, dtype=torch.int64), metas


def resize_image(image, min_dim=None, max_dim=None, padding=False, device="cuda:0"):
    """utils.resize_image with the reference's signature and return values; the image comes back as a uint8 device
    tensor [H,W,3]."""
    img = _to_device_u8(image, torch.device(device))
    h, w = img.shape[:2]
    new_h, new_w, window, scale, pad = resize_plan(h, w, min_dim, max_dim, padding)
    if scale != 1:
        img = ops.resize_bilinear_u8(img, new_h, new_w)
    if padding:
        canvas = torch.zeros(max_dim, max_dim, 3, dtype=torch.uint8, device=img.device)
        canvas[window[0]:window[2], window[1]:window[3]] = img
        img = canvas
    return img, window, scale, pad


def full_masks(class_id: torch.Tensor, boxes: torch.Tensor, masks: torch.Tensor, height: int, width: int,
               channels_last: bool = False) -> torch.Tensor:
    """datalib.full_masks(class_id [N], boxes [N,4], masks [N,C,28,28], height, width) → bool [N,height,width].
    channels_last=True takes this library's [N,28,28,C] mask-head output directly."""
    return ops.paste_masks(masks, class_id, boxes, height, width, channels_last)


def decode_boxes(boxes: torch.Tensor, scale, window) -> torch.Tensor:
    """data.py:331-343: shift by the window origin, then multiply by 1/(scale + 1e-5) (fp32 tensor ops)."""
    if scale == 1:
        return boxes
    off = torch.tensor([window[0], window[1], window[0], window[1]], dtype=boxes.dtype, device=boxes.device)
    return (boxes - off) * torch.tensor([1.0 / (scale + 1e-5)] * 4, dtype=boxes.dtype, device=boxes.device)


def decode_masks(masks_l8: torch.Tensor, scale, window) -> torch.Tensor:
    """data.py:264-284 for masks given as 0/255 'L' images [N,H,W] (ops.paste_masks(as_l8=True)): CenterCrop to the
    window's size, PIL-resize by 1/scale → uint8 [N, round(h/scale), round(w/scale)] (grey levels, as the reference
    returns them). The caller handles scale == 1 (the reference returns the boolean masks untouched)."""
    n, hh, ww = masks_l8.shape
    ch, cw = window[2] - window[0], window[3] - window[1]
    top = int(round((hh - ch) / 2.0))      # torchvision center_crop; not the window origin when the pad is odd
    left = int(round((ww - cw) / 2.0))
    nh, nw = round(ch * 1.0 / scale), round(cw * 1.0 / scale)
    return ops.resize_bilinear_u8(masks_l8[:, top:top + ch, left:left + cw], nh, nw)
