"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/*.h declares
(no compute without a GPU), the drop-in `maskrcnn` package has the reference's surface, and the
product never touches oracle/."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from maskrcnn_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 10
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/maskrcnn_hip.h but not exported"
    assert set(_lib._SIGS) == set(names), "ctypes table and header disagree"
    assert _lib.lib.mrcnn_abi_version() == 1
    assert _lib.lib.mrcnn_arch() == b"gfx950"


def test_dropin_surface_matches_reference():
    import maskrcnn
    assert callable(maskrcnn.nms) and callable(maskrcnn.CropFunction(7, 7, 0))
    for name in ("nms", "crop_forward", "crop_backward"):  # csrc/vision.cpp:11-15
        assert callable(getattr(maskrcnn._C, name))
    for name in ("nms", "crop_forward", "crop_backward", "crop"):
        assert hasattr(torch.ops.maskrcnn, name)
    f = maskrcnn.CropFunction(14, 14)
    assert (f.crop_height, f.crop_width, f.extrapolation_value) == (14, 14, 0)


def test_cpu_tensors_are_rejected_loudly():
    import maskrcnn
    with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
        maskrcnn.nms(torch.zeros(4, 5), 0.5)
    with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
        maskrcnn.CropFunction(2, 2)(torch.zeros(1, 1, 4, 4), torch.zeros(1, 4),
                                    torch.zeros(1, dtype=torch.int32))


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]|liboracle", re.M)
    for pkg in ("maskrcnn_amd", "maskrcnn"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert not pat.search(text), f"{pkg}/{f} references oracle/"
