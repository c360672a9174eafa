"""Drop-in replacement for the reference's `maskrcnn` extension package
(c++ext/maskrcnn/__init__.py), backed by libmaskrcnn_hip.so on MI355X.

    import maskrcnn
    keep  = maskrcnn.nms(dets, threshold)                              # __init__.py:21-22
    crops = maskrcnn.CropFunction(h, w, extrapolation)(image, boxes, box_ind)   # :25-45, model.py:373

`maskrcnn._C` exports nms / crop_forward / crop_backward with the pybind signatures of
csrc/vision.cpp:11-15, so model.py-shaped code imports this package unchanged.
"""
import torch

import maskrcnn_amd  # noqa: F401  loads the HIP library, registers torch.ops.maskrcnn
from . import _C


def nms(dets, threshold):
    return _C.nms(dets, threshold)


class _Crop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, boxes, box_ind, crop_height, crop_width, extrapolation_value):
        crops = torch.ops.maskrcnn.crop(image, boxes, box_ind, float(extrapolation_value),
                                        int(crop_height), int(crop_width))
        ctx.im_size = image.size()
        ctx.save_for_backward(boxes, box_ind)
        return crops

    @staticmethod
    def backward(ctx, grad_outputs):
        boxes, box_ind = ctx.saved_tensors
        grad_image = torch.empty(ctx.im_size, dtype=grad_outputs.dtype, device=grad_outputs.device)
        _C.crop_backward(grad_outputs.contiguous(), boxes, box_ind, grad_image)
        return grad_image, None, None, None, None, None


class CropFunction:
    """Same construction and call shape as the reference's legacy autograd Function (which modern
    torch refuses to run): CropFunction(crop_height, crop_width, extrapolation_value=0)(image, boxes,
    box_ind) -> crops [N, C, crop_height, crop_width]. Differentiable w.r.t. `image`."""

    def __init__(self, crop_height, crop_width, extrapolation_value=0):
        self.crop_height = crop_height
        self.crop_width = crop_width
        self.extrapolation_value = extrapolation_value

    def __call__(self, image, boxes, box_ind):
        return _Crop.apply(image, boxes, box_ind, self.crop_height, self.crop_width,
                           self.extrapolation_value)

    forward = __call__


__all__ = ["nms", "CropFunction", "_C"]
