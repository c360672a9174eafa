"""maskrcnn._C_native (maskrcnn/csrc/vision_hip.cpp): the reference's pybind module and the C++-side TORCH_LIBRARY registration on
the HIP library. Same kernels as torch.ops.maskrcnn.* (registered from Python over ctypes): the two must agree bit for bit, on
the reference-generated golden vectors too, and the dispatcher ops must run from TorchScript."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from maskrcnn import build_native
    import maskrcnn_amd  # noqa: F401  (torch.ops.maskrcnn.*)
    return build_native.load()


def _dets(n, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 2, generator=g, dtype=torch.float64) * 300
    wh = torch.rand(n, 2, generator=g, dtype=torch.float64) * 80 + 2
    s = torch.randperm(n, generator=g).double() / n
    return torch.cat([c, c + wh, s[:, None]], 1).to(dtype)


@pytest.mark.parametrize("n,dtype", [(1, torch.float32), (500, torch.float32), (1000, torch.float32), (5000, torch.float32),
                                     (20000, torch.float32), (1000, torch.float64)])
def test_native_nms_equals_python_registered_op(native, n, dtype):
    dev = torch.device("cuda:0")
    d = _dets(n, n, dtype).to(dev)
    want = torch.ops.maskrcnn.nms(d, 0.7)
    got = native.nms(d, 0.7)
    assert got.dtype == torch.int64 and got.device == d.device and torch.equal(got, want)
    assert torch.equal(torch.ops.maskrcnn_native.nms(d, 0.7), want)
    # a strided view (every second row of a wider tensor), as the reference's callers may pass
    wide = torch.zeros(2 * n, 7, dtype=dtype, device=dev)
    wide[::2, 1:6] = d
    assert torch.equal(native.nms(wide[::2, 1:6], 0.7), want)


def test_native_nms_on_reference_golden_vectors(native):
    from conftest import load_golden
    z = load_golden("nms")
    dev = torch.device("cuda:0")
    keys = sorted(k[:-5] for k in z.files if k.endswith("_dets"))
    assert keys
    for k in keys:
        dets = torch.from_numpy(z[k + "_dets"])
        thr = float(z[k + "_thr"]) if (k + "_thr") in z.files else 0.7
        keep = torch.from_numpy(z[k + "_keep"]).long()
        assert torch.equal(native.nms(dets.to(dev), thr).cpu(), keep), k


def test_native_crop_forward_backward_equal_python_registered_ops(native):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    image = torch.randn(3, 32, 40, 56, generator=g).to(dev)
    bx = torch.rand(64, 4, generator=g)
    bx = torch.cat([torch.minimum(bx[:, :2], bx[:, 2:]) - 0.1, torch.maximum(bx[:, :2], bx[:, 2:]) + 0.1], 1).to(dev)   # some outside
    bi = torch.randint(0, 3, (64,), generator=g, dtype=torch.int32).to(dev)
    want = torch.empty(1, device=dev)
    torch.ops.maskrcnn.crop_forward(image, bx, bi, 0.5, 7, 9, want)
    got = torch.empty(5, 5, device=dev)            # any float tensor: resized in place (crop_cpu.cpp:141-143)
    native.crop_forward(image, bx, bi, 0.5, 7, 9, got)
    assert tuple(got.shape) == (64, 32, 7, 9) and torch.equal(got, want)
    got2 = torch.empty(0, device=dev)
    torch.ops.maskrcnn_native.crop_forward(image, bx, bi, 0.5, 7, 9, got2)
    assert torch.equal(got2, want)
    grads = torch.randn(64, 32, 7, 9, generator=g).to(dev)
    gi_want, gi_got = torch.full_like(image, 7.0), torch.full_like(image, -3.0)   # zeroed inside (crop_cpu.cpp:197)
    torch.ops.maskrcnn.crop_backward(grads, bx, bi, gi_want)
    native.crop_backward(grads, bx, bi, gi_got)
    # fp32 atomics: the summation order is not fixed between two launches; the two launches agree to rounding
    assert torch.allclose(gi_got, gi_want, rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError, match="Int"):
        native.crop_forward(image, bx, bi.long(), 0.5, 7, 9, got)


def test_native_ops_run_from_torchscript(native):
    @torch.jit.script
    def keep_boxes(dets: torch.Tensor, thr: float) -> torch.Tensor:
        return dets[torch.ops.maskrcnn_native.nms(dets, thr)]

    dev = torch.device("cuda:0")
    d = _dets(800, 11).to(dev)
    want = d[torch.ops.maskrcnn.nms(d, 0.5)]
    assert torch.equal(keep_boxes(d, 0.5), want)
    assert "maskrcnn_native::nms" in str(keep_boxes.graph)
