"""maskrcnn_amd — MI355X-native (gfx950) Mask R-CNN inference hot path.

Importing the package loads libmaskrcnn_hip.so and registers torch.ops.maskrcnn.*; there is no CPU or
PyTorch fallback for the ops (a missing library is an ImportError).
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP library is missing)
from . import ops  # noqa: F401  (registers torch.ops.maskrcnn.*)

__all__ = ["ops"]
