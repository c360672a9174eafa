"""The reference's two refine stages with their own signatures (batch 1, dynamically sized results), on the HIP kernels of
the batched pipeline:

    rpn_refine(rpn_class, rpn_bbox, anchors, cfg)  ==  MaskRCNN.rpn_refine(rpn_class, rpn_bbox)        model.py:1307-1382
    mrn_refine(rois, probs, deltas, window, cfg)   ==  MaskRCNN.mrn_refine(rois, probs, deltas, window) model.py:1389-1487

`MaskRCNNInference.predict` keeps fixed shapes and never synchronises with the host; these two wrappers exist for callers that
hold the reference's tensors (and for parity tests against tests/golden/refine.npz, the reference's own outputs): like the
reference they return tensors cut to the number of survivors, which costs ONE device-to-host read per call.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .config import InferenceConfig
from .pipeline import MaskRCNNInference


def rpn_refine(rpn_class: torch.Tensor, rpn_bbox: torch.Tensor, anchors: torch.Tensor, cfg: InferenceConfig) -> torch.Tensor:
    """rpn_class [1, A, 2] (bg, fg) probabilities, rpn_bbox [1, A, 4] deltas, anchors [A, 4] pixel boxes (fp32, on the GPU) →
    rois [1, R, 4] normalised, R <= cfg.proposal_count: scores = rpn_class[:, :, 1]; top cfg.pre_nms_limit by score (:1345-1350);
    deltas * RPN_BBOX_STD_DEV, boxes_refine, clip to the image (:1341-1358); nms at cfg.rpn_nms_threshold (:1364);
    keep[:proposal_count]; normalise (:1366-1374) — top-k, decode, NMS and selection are one HIP launch each."""
    assert rpn_class.dim() == 3 and rpn_class.size(0) == 1 and rpn_class.size(2) == 2, "the reference is batch-1 here (model.py:1321)"
    assert tuple(rpn_bbox.shape) == (1, rpn_class.size(1), 4) and tuple(anchors.shape) == (rpn_class.size(1), 4)
    ns = SimpleNamespace(cfg=cfg, anchors=anchors.contiguous())
    rois, counts, _ = MaskRCNNInference.proposals(ns, rpn_class[:, :, 1].contiguous(), rpn_bbox.contiguous())
    return rois[:, :int(counts[0])]


def mrn_refine(rois: torch.Tensor, probs: torch.Tensor, deltas: torch.Tensor, window, cfg: InferenceConfig):
    """rois [1, N, 4] normalised, probs [N, num_classes] class probabilities, deltas [N, num_classes, 4], window (y1, x1, y2, x2)
    in pixels → (class_ids [1, D] int64, scores [1, D], boxes [1, D, 4] pixel, integral-valued), D <= cfg.detection_max_instances,
    by descending score; (None, None, None) when nothing is kept (:1445-1447). argmax class and its score (:1407-1415),
    class-specific refine * RPN_BBOX_STD_DEV, scale to pixels, clip to the window, round (:1418-1432), background / confidence
    filter (:1437-1443), per-class NMS (:1454-1475, one class-aware launch), top-D (:1478-1487). The decode kernel takes
    logits: log(probs) is handed over, whose softmax is probs again (to rounding: scores agree to ~1e-7)."""
    assert rois.dim() == 3 and rois.size(0) == 1, "the reference is batch-1 here"
    n = rois.size(1)
    assert tuple(probs.shape) == (n, cfg.num_classes) and tuple(deltas.shape) == (n, cfg.num_classes, 4)
    ns = SimpleNamespace(cfg=cfg)
    counts = torch.tensor([n], dtype=torch.int32, device=rois.device)
    windows = torch.tensor([[float(v) for v in window]], dtype=torch.float32, device=rois.device)
    ids, scores, boxes, _, kept = MaskRCNNInference.detections(ns, rois.contiguous(), counts, torch.log(probs).contiguous(),
                                                               deltas.contiguous(), windows)
    d = int(kept[0])
    if d == 0:
        return None, None, None
    return ids[:, :d], scores[:, :d], boxes[:, :d]


__all__ = ["rpn_refine", "mrn_refine"]
