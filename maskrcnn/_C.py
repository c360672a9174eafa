"""`maskrcnn._C`: the three functions the reference's pybind module exports
(c++ext/maskrcnn/csrc/vision.cpp:11-15), routed to torch.ops.maskrcnn (HIP, gfx950)."""
import torch

import maskrcnn_amd  # noqa: F401


def nms(dets, threshold):
    """non-maximum suppression (nms.h:15)"""
    return torch.ops.maskrcnn.nms(dets, float(threshold))


def crop_forward(image, boxes, box_index, extrapolation_value, crop_height, crop_width, crops):
    """crop forward (crop.h:14-22); `crops` is resized and overwritten in place."""
    torch.ops.maskrcnn.crop_forward(image, boxes, box_index, float(extrapolation_value),
                                    int(crop_height), int(crop_width), crops)


def crop_backward(grads, boxes, box_index, grads_image):
    """crop backward (crop.h:36-41); `grads_image` is zeroed and accumulated in place."""
    torch.ops.maskrcnn.crop_backward(grads, boxes, box_index, grads_image)
