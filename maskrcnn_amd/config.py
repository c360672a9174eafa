"""The constants the inference path reads (reference config.py:54-126,199-204), as a plain dataclass.
Two fields are parameters here that the reference hard-codes: `pre_nms_limit` (500 at model.py:1345) and
`proposal_count` (RPN_NMS_MAX_ROIS_NUM = 500, config.py:76) — BASELINE.json's metric is quoted at 1000
proposals per image."""
from __future__ import annotations

import math
from dataclasses import dataclass, field


@dataclass
class InferenceConfig:
    image_height: int = 1024            # IMAGE_MAX_DIM padded square (config.py:157-158)
    image_width: int = 1024
    backbone: str = "resnet101"         # hard-coded at model.py:985
    backbone_strides: tuple = (4, 8, 16, 32, 64)
    rpn_anchor_scales: tuple = (32, 64, 128, 256, 512)
    rpn_anchor_ratios: tuple = (0.5, 1, 2)
    rpn_anchor_stride: int = 1
    rpn_nms_threshold: float = 0.7
    pre_nms_limit: int = 500
    proposal_count: int = 500
    rpn_bbox_std_dev: tuple = (0.1, 0.1, 0.2, 0.2)
    pool_size: int = 7
    mask_pool_size: int = 14
    detection_max_instances: int = 50
    detection_min_confidence: float = 0.0   # CocoInferenceConfig (config.py:204): falsy → no filter
    detection_nms_threshold: float = 0.3
    num_classes: int = 81
    mean_pixel: tuple = (123.7, 116.8, 103.9)
    image_min_dim: int = 800            # config.py:145-149: resize so the short side is >= 800 (never shrink for it),
    image_max_dim: int = 1024           #   the long side <= 1024, then zero-pad to the canvas (IMAGE_PADDING = True)
    backbone_shapes: list = field(init=False)

    def __post_init__(self):
        if self.image_height % 64 or self.image_width % 64:  # model.py:978-983
            raise ValueError("Image size must be dividable by 2 at least 6 times")
        # kernel limits, checked here rather than deep inside the first predict(): the LDS-resident sorts of
        # csrc/nms.hip and csrc/select.hip hold at most 4096 boxes per image (mrcnn_nms_max_boxes())
        cap = 4096
        if not 1 <= self.pre_nms_limit <= cap:
            raise ValueError(f"pre_nms_limit={self.pre_nms_limit}: the NMS / top-k kernels take 1..{cap} boxes per image")
        if not 1 <= self.proposal_count <= cap:
            raise ValueError(f"proposal_count={self.proposal_count}: must be in 1..{cap}")
        if not 1 <= self.detection_max_instances <= self.proposal_count:
            raise ValueError(f"detection_max_instances={self.detection_max_instances}: must be in 1..proposal_count")
        if self.backbone not in ("resnet50", "resnet101"):
            raise ValueError(f"backbone={self.backbone!r}: resnet50 or resnet101 (model.py:217-219)")
        self.backbone_shapes = [(int(math.ceil(self.image_height / s)), int(math.ceil(self.image_width / s)))
                                for s in self.backbone_strides]
