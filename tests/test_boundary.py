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
    assert _lib.lib.mrcnn_abi_version() == _lib.header_abi_version() >= 2
    assert _lib.lib.mrcnn_arch() == b"gfx950"


def test_ctypes_table_matches_header_prototypes():
    """_SIGS is a hand-kept copy of the header: argument counts and types are cross-checked against the prototypes
    (a mismatch would pass wrong-width arguments to a kernel launch)."""
    from maskrcnn_amd import _lib
    protos = _lib.header_prototypes()
    assert set(protos) == set(_lib._SIGS)
    for name, (res, args) in _lib._SIGS.items():
        hres, hargs = protos[name]
        assert res == hres, f"{name}: return type {res} vs header {hres}"
        assert len(args) == len(hargs), f"{name}: {len(args)} arguments vs {len(hargs)} in the header"
        for i, (a, h) in enumerate(zip(args, hargs)):
            assert a == h, f"{name}: argument {i} is {a} in _SIGS, {h} in the header"


def test_stale_library_is_rejected(tmp_path, monkeypatch):
    """An .so built from another revision of the header must not load (same names, other argument lists)."""
    from maskrcnn_amd import _lib
    fake = tmp_path / "maskrcnn_hip.h"
    fake.write_text(open(_lib.HEADER).read().replace(f"#define MRCNN_ABI_VERSION {_lib.header_abi_version()}",
                                                     "#define MRCNN_ABI_VERSION 9999"))
    monkeypatch.setattr(_lib, "HEADER", str(fake))
    monkeypatch.setattr(_lib, "header_abi_version", lambda header=str(fake): 9999)
    with pytest.raises(ImportError, match="ABI version"):
        _lib._load()


def test_inference_config_validates_kernel_limits():
    from maskrcnn_amd.config import InferenceConfig
    InferenceConfig(pre_nms_limit=4096, proposal_count=1000)
    for kw in (dict(pre_nms_limit=6000), dict(proposal_count=5000, pre_nms_limit=4096), dict(detection_max_instances=0),
               dict(backbone="resnet18"), dict(image_height=1000)):
        with pytest.raises(ValueError):
            InferenceConfig(**kw)


def test_dropin_surface_matches_reference():
    import maskrcnn
    assert callable(maskrcnn.nms) and callable(maskrcnn.CropFunction(7, 7, 0))
    for name in ("nms", "crop_forward", "crop_backward"):  # csrc/vision.cpp:11-15
        assert callable(getattr(maskrcnn._C, name))
    for name in ("nms", "crop_forward", "crop_backward", "crop"):
        assert hasattr(torch.ops.maskrcnn, name)
    f = maskrcnn.CropFunction(14, 14)
    assert (f.crop_height, f.crop_width, f.extrapolation_value) == (14, 14, 0)


def test_cpu_tensors_without_a_gpu_fail_loudly():
    """CPU tensors are staged through the GPU (tests/test_gpu_ops.py::test_cpu_tensors_through_the_dropin); there is no CPU
    arithmetic in the product, so without a visible GPU the call must raise — never fall back to anything."""
    import maskrcnn
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the staged path runs (covered under -m gpu)")
    with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
        maskrcnn.nms(torch.rand(4, 5), 0.5)
    with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
        maskrcnn.CropFunction(2, 2)(torch.zeros(1, 1, 4, 4), torch.zeros(1, 4),
                                    torch.zeros(1, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
        torch.ops.maskrcnn.conv_bn_act(torch.zeros(1, 4, 4, 32), torch.zeros(32, 1, 1, 32), None, None, 1, [0, 0, 0, 0], False,
                                       None, 1)


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]|liboracle", re.M)
    for pkg in ("maskrcnn_amd", "maskrcnn"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert not pat.search(text), f"{pkg}/{f} references oracle/"


def test_entry_points_refuse_tensors_beyond_32bit_offsets():
    """Argument validation runs before any HIP call (no GPU needed, the pointers are never touched): the fp32 stem + pool entry
    point writes 16 B per input pixel through 32-bit buffer byte offsets — 256 images of 1024 x 1024 (2^28 pixels, a 4 GiB
    output) must be refused, not wrapped (round-5 advice)."""
    from maskrcnn_amd import _lib
    lib = _lib.lib
    dummy = ctypes.c_void_p(16)
    rc = lib.mrcnn_stem_conv7x7_s2_pool_f32(dummy, 256, 1024, 1024, dummy, None, None, dummy, None)
    assert rc != 0 and b"too large" in lib.mrcnn_last_error()
    rc = lib.mrcnn_conv3x3_winograd4_f32(dummy, 1, 64, 64, 12, dummy, 64, None, None, 1, dummy, None, None)
    assert rc != 0 and b"Cin" in lib.mrcnn_last_error()


def test_native_cxx_module_loads_and_registers():
    """maskrcnn/_C_native.so (maskrcnn/csrc/vision_hip.cpp): the reference's pybind module — the three names and doc strings of
    c++ext/maskrcnn/csrc/vision.cpp:11-15 — built on the C ABI, plus the TORCH_LIBRARY registration maskrcnn_native::* a C++ /
    TorchScript caller uses. No GPU here: it must import, expose the functions and the dispatcher schemas, and refuse CPU tensors
    the way the reference built without CPU support would (nms.h:24)."""
    from maskrcnn import build_native
    build_native.build()
    m = build_native.load()
    assert m.nms.__doc__.strip().endswith("non-maximum suppression")
    assert m.crop_forward.__doc__.strip().endswith("crop forward") and m.crop_backward.__doc__.strip().endswith("crop backward")
    s = torch.ops.maskrcnn_native.nms.default._schema
    assert str(s) == "maskrcnn_native::nms(Tensor dets, float threshold) -> Tensor"
    assert "Tensor(a!) crops" in str(torch.ops.maskrcnn_native.crop_forward.default._schema)
    assert "Tensor(a!) grads_image" in str(torch.ops.maskrcnn_native.crop_backward.default._schema)
    with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
        m.nms(torch.zeros(3, 5), 0.5)
