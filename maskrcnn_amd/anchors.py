"""Pyramid anchors (reference utils.py:116-291, consumed at model.py:991-995).

Per level: h = scale/sqrt(ratio), w = scale*sqrt(ratio); centres on the cell ORIGIN (x*stride, no half-cell
offset); anchor index = ((y*W + x)*R + ratio_index); corners = centre -/+ size/2, in float64, then
narrowed to float32."""
from __future__ import annotations

import numpy as np
import torch


def level_anchors(scale, ratios, shape, feature_stride, anchor_stride=1) -> np.ndarray:
    r = np.sqrt(np.asarray(ratios, dtype=np.float64))
    hs, ws = scale / r, scale * r                                  # [R]
    ys = np.arange(0, shape[0], anchor_stride, dtype=np.int64) * feature_stride
    xs = np.arange(0, shape[1], anchor_stride, dtype=np.int64) * feature_stride
    out = np.empty((len(ys), len(xs), len(r), 4), dtype=np.float64)
    out[..., 0] = ys[:, None, None] - 0.5 * hs[None, None, :]
    out[..., 1] = xs[None, :, None] - 0.5 * ws[None, None, :]
    out[..., 2] = ys[:, None, None] + 0.5 * hs[None, None, :]
    out[..., 3] = xs[None, :, None] + 0.5 * ws[None, None, :]
    return out.reshape(-1, 4)


def pyramid_anchors(cfg) -> torch.Tensor:
    """float32 [sum_l H_l*W_l*R, 4] (261888 rows at 1024x1024)."""
    a = np.concatenate([level_anchors(cfg.rpn_anchor_scales[i], cfg.rpn_anchor_ratios,
                                      cfg.backbone_shapes[i], cfg.backbone_strides[i],
                                      cfg.rpn_anchor_stride)
                        for i in range(len(cfg.rpn_anchor_scales))], axis=0)
    return torch.from_numpy(a).float()
