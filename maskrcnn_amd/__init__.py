"""maskrcnn_amd — MI355X-native (gfx950) Mask R-CNN inference hot path.

Importing the package loads libmaskrcnn_hip.so and registers torch.ops.maskrcnn.*; there is no CPU or
PyTorch fallback for the ops (a missing library is an ImportError).
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP library is missing)
from . import ops  # noqa: F401  (registers torch.ops.maskrcnn.*)



def roi_align(inputs, pool_size, image_shape):
    """Drop-in for the reference's `roi_align(inputs, pool_size, image_shape)` (model.py:276-393), same arguments and result:
    inputs = [boxes [1, N, 4] or [N, 4] normalised (y1, x1, y2, x2)] + [P2, P3, P4, P5], each [1, C, H_l, W_l] NCHW fp32 on the
    GPU; image_shape = (height, width[, channels]) of the padded image → pooled [N, C, pool_size, pool_size] in RoI order.
    The reference loops over the four levels with a `nonzero()` / `any()` host synchronisation per level, one `CropFunction`
    call each, a `cat` and an inverse permutation (:340-387); here the level of every box (:331-338, same fp32 operations) is
    decided inside ONE launch of the pyramid kernel (mrcnn_roi_align_pyramid_nhwc_f32) and nothing touches the host. Like the
    reference (:312-313) the batch dimension of every entry of `inputs` is squeezed IN PLACE in the caller's list. The maps are
    converted to channels-last on the way in and the crops back on the way out (the pipeline, which keeps NHWC maps, calls
    ops.roi_align_pyramid directly)."""
    for i in range(len(inputs)):
        inputs[i] = inputs[i].squeeze(0)
    boxes, maps = inputs[0], inputs[1:]
    if len(maps) != 4:
        raise RuntimeError(f"roi_align: four pyramid levels expected, got {len(maps)}")
    fms = [ops.nchw_to_nhwc(m.unsqueeze(0).contiguous()) for m in maps]
    area = float(image_shape[0] * image_shape[1])
    pooled = ops.roi_align_pyramid(fms, boxes.contiguous(), int(pool_size), area, rois_per_image=boxes.size(0))
    return ops.nhwc_to_nchw(pooled)


__all__ = ["ops", "roi_align"]   # reference-signature refine stages: maskrcnn_amd.refine (imports the pipeline)
