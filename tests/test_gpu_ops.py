"""GPU parity tests (-m gpu): the HIP path (through the C ABI in libmaskrcnn_hip.so) against the CPU
oracle and the committed golden vectors. NMS keep indices and crop/RoIAlign activations are compared
BIT-EXACT (the kernels restate the reference's fp32 op order with FP contraction off); crop_backward
(atomic accumulation order) within 1e-5 abs."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import maskrcnn_amd  # noqa: F401  must load libmaskrcnn_hip.so or fail loudly
    return torch.device("cuda:0")


def _cases(z):
    i = 0
    while f"c{i}_tag" in z.files:
        yield i, str(z[f"c{i}_tag"])
        i += 1


def _rand_dets(g, n, extent=600.0, spread=10.0):
    k = max(1, n // 10)
    centres = torch.rand(k, 2, generator=g) * extent
    c = centres[torch.randint(0, k, (n,), generator=g)] + torch.randn(n, 2, generator=g) * spread
    hw = torch.exp(torch.rand(n, 2, generator=g) * 2.5 + 2.0)
    b = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, extent)
    s = (torch.randperm(n, generator=g).float() + torch.rand(n, generator=g) * 0.5) / n  # distinct
    return torch.cat([b, s[:, None]], 1)


# ---------------------------------------------------------------------------------------------- NMS
def test_nms_golden(dev):
    import maskrcnn
    z = load_golden("nms")
    for i, tag in _cases(z):
        dets = torch.from_numpy(z[f"c{i}_dets"])   # one case is float64: the reference dispatches over the floating types
        keep = maskrcnn.nms(dets.to(dev), float(z[f"c{i}_thr"]))
        assert keep.dtype == torch.int64 and keep.device.type == "cuda"
        assert np.array_equal(keep.cpu().numpy(), z[f"c{i}_keep"]), tag
    z = load_golden("nms_strided")
    wide = torch.from_numpy(z["wide"]).to(dev)
    assert np.array_equal(maskrcnn.nms(wide[:, 1:6], float(z["thr"])).cpu().numpy(), z["keep"])
    assert maskrcnn.nms(torch.zeros(0, 5, device=dev), 0.5).numel() == 0


def test_nms_random_vs_oracle(dev, oracle):
    import maskrcnn
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(7)
    for n in (1, 5, 64, 65, 255, 256, 257, 777, 1000, 1024, 1025, 2048, 3000, 4096, 6000, 16384):
        for thr in (0.3, 0.7):
            d = _rand_dets(g, n)
            want = oracle.nms(d, thr)
            got = maskrcnn.nms(d.to(dev), thr).cpu()
            assert torch.equal(got, want), (n, thr)
            # both device paths: chip-wide pair mask (workspace, up to 16384 boxes) and the single-launch LDS kernel (4096)
            for ws in ((True, False) if n <= 4096 else (True,)):
                keep, cnt = ops.nms_batched(d.to(dev).unsqueeze(0), thr, use_workspace=ws)
                assert torch.equal(keep[0, :int(cnt[0])].cpu(), want), (n, thr, ws)
    with pytest.raises(RuntimeError):
        ops.nms_batched(_rand_dets(g, 5000).to(dev).unsqueeze(0), 0.5, use_workspace=False)


def test_nms_any_size_and_float64(dev, oracle):
    """The rest of the reference's nms surface (nms.h:15-30, cpu/nms_cpu.cpp:73-79): more boxes than the fp32 pair-mask
    path takes (16384), and float64 boxes at any size — csrc/nms_general.hip, bit-exact keep indices vs the CPU oracle (whose
    f64 instantiation is pinned by the reference-generated float64 golden case)."""
    import maskrcnn
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(70)
    for n, thr in ((16385, 0.5), (20000, 0.7), (40000, 0.3)):
        d = _rand_dets(g, n, extent=4096.0)
        assert torch.equal(maskrcnn.nms(d.to(dev), thr).cpu(), oracle.nms(d, thr)), (n, thr)
    for n, thr in ((1, 0.5), (63, 0.5), (64, 0.3), (65, 0.7), (150, 0.7), (1000, 0.5), (5000, 0.7), (17000, 0.5)):
        d = _rand_dets(g, n).double()
        d[:, :4] += torch.rand(n, 4, generator=g, dtype=torch.float64) * 1e-9   # bits a float cannot hold
        want = oracle.nms(d, thr)
        assert torch.equal(maskrcnn.nms(d.to(dev), thr).cpu(), want), (n, thr)
        if n >= 1000:   # differs from what the same boxes rounded to float32 give? then the f64 arithmetic mattered
            keep, cnt = ops.nms_general(d.to(dev), thr)
            assert torch.equal(keep[:int(cnt)].cpu(), want) and bool((keep[int(cnt):] == -1).all())
    # the general path on float32 boxes equals the pair-mask path (same order, same arithmetic)
    d = _rand_dets(g, 3000)
    keep, cnt = ops.nms_general(d.to(dev), 0.6)
    assert torch.equal(keep[:int(cnt)].cpu(), oracle.nms(d, 0.6))
    # ties, NaN / inf / signed-zero scores, a NaN coordinate, strided rows: the same order rules as the fp32 kernels
    d = _rand_dets(g, 300).double()
    d[:, 4] = torch.randint(0, 5, (300,), generator=g).double() / 4
    d[7, 4], d[9, 4], d[11, 4], d[12, 4] = float("nan"), float("inf"), -0.0, 0.0
    d[21, 0] = float("nan")
    assert torch.equal(maskrcnn.nms(d.to(dev), 0.5).cpu(), oracle.nms(d, 0.5))
    wide = torch.randn(200, 9, generator=g, dtype=torch.float64)
    wide[:, 1:6] = _rand_dets(g, 200).double()
    assert torch.equal(maskrcnn.nms(wide.to(dev)[:, 1:6], 0.6).cpu(), oracle.nms(wide[:, 1:6], 0.6))
    with pytest.raises(RuntimeError):
        maskrcnn.nms(_rand_dets(g, 10).half().to(dev), 0.5)


def test_nms_ties_and_specials(dev, oracle):
    import maskrcnn
    g = torch.Generator().manual_seed(8)
    d = _rand_dets(g, 300)
    d[:, 4] = torch.randint(0, 5, (300,), generator=g).float() / 4  # heavy ties → index order
    d[7, 4] = float("nan")
    d[9, 4] = float("inf")
    d[11, 4] = -0.0
    d[12, 4] = 0.0
    d[20, :4] = torch.tensor([5., 5., 3., 3.])  # degenerate
    d[21, 0] = float("nan")
    assert torch.equal(maskrcnn.nms(d.to(dev), 0.5).cpu(), oracle.nms(d, 0.5))


def test_nms_batched_class_aware(dev, oracle):
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(9)
    S, N = 8, 1000
    dets = torch.stack([_rand_dets(g, N) for _ in range(S)])
    counts = torch.tensor([1000, 999, 512, 1, 0, 64, 65, 300], dtype=torch.int32)
    cls = torch.randint(1, 9, (S, N), generator=g, dtype=torch.int32)
    keep1, cnt1 = ops.nms_batched(dets.to(dev), 0.3, counts.to(dev), cls.to(dev), use_workspace=False)
    keep, cnt = ops.nms_batched(dets.to(dev), 0.3, counts.to(dev), cls.to(dev))
    assert torch.equal(keep, keep1) and torch.equal(cnt, cnt1)
    keep, cnt = keep.cpu(), cnt.cpu()
    for s in range(S):
        n = int(counts[s])
        want = oracle.nms(dets[s, :n], 0.3, class_ids=cls[s, :n]) if n else torch.empty(0, dtype=torch.int64)
        assert int(cnt[s]) == want.numel(), s
        assert torch.equal(keep[s, :want.numel()], want), s
        assert bool((keep[s, want.numel():] == -1).all())


def test_nms_properties_full_size(dev):
    """BASELINE size (1000 proposals x 8 images): size-independent properties."""
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(10)
    S, N = 8, 1000
    dets = torch.stack([_rand_dets(g, N, extent=1024.0) for _ in range(S)]).to(dev)
    keep, cnt = ops.nms_batched(dets, 0.7)
    for s in range(S):
        k = keep[s, :int(cnt[s])]
        assert bool((k[1:] > k[:-1]).all())  # ascending input indices
        kept = dets[s, k]
        keep2, cnt2 = ops.nms_batched(kept.unsqueeze(0).contiguous(), 0.7)
        assert int(cnt2[0]) == k.numel()  # idempotence: survivors do not suppress each other
        # the top-scoring box always survives
        assert int(dets[s, :, 4].argmax()) in set(k.tolist())


# --------------------------------------------------------------------------------------------- crop
def test_crop_forward_golden(dev):
    import maskrcnn
    z = load_golden("crop_forward")
    for i, tag in _cases(z):
        extrap, ch, cw = z[f"c{i}_args"]
        img = torch.from_numpy(z[f"c{i}_image"]).to(dev)
        boxes = torch.from_numpy(z[f"c{i}_boxes"]).to(dev)
        ind = torch.from_numpy(z[f"c{i}_ind"]).to(dev)
        got = maskrcnn.CropFunction(int(ch), int(cw), float(extrap))(img, boxes, ind)
        assert np.array_equal(got.cpu().numpy(), z[f"c{i}_crops"]), tag
        # out-param form of crop.h:14-22: any float tensor, resized by the callee
        crops = torch.zeros_like(img)
        maskrcnn._C.crop_forward(img, boxes, ind, float(extrap), int(ch), int(cw), crops)
        assert np.array_equal(crops.cpu().numpy(), z[f"c{i}_crops"]), tag


def test_cpu_tensors_through_the_dropin(dev):
    """The reference's dispatch takes CPU tensors and returns CPU tensors (nms.h:15-30, crop.h:14-53; SURVEY 8b "CPU in -> CPU
    out"). Here they are staged through the GPU — same HIP kernels, never the oracle — so the reference-generated golden
    vectors must come back bit for bit as CPU tensors: nms int64 ascending; `crops` resized in place; crop_backward in place."""
    import maskrcnn
    z = load_golden("nms")
    for i, tag in _cases(z):
        dets = torch.from_numpy(z[f"c{i}_dets"])
        keep = maskrcnn.nms(dets, float(z[f"c{i}_thr"]))
        assert keep.dtype == torch.int64 and keep.device.type == "cpu"
        assert np.array_equal(keep.numpy(), z[f"c{i}_keep"]), tag
    empty = maskrcnn.nms(torch.zeros(0, 5), 0.5)
    assert empty.numel() == 0 and empty.dtype == torch.int64 and empty.device.type == "cpu"
    z = load_golden("crop_forward")
    for i, tag in _cases(z):
        extrap, ch, cw = z[f"c{i}_args"]
        img, boxes, ind = (torch.from_numpy(z[f"c{i}_{k}"]) for k in ("image", "boxes", "ind"))
        got = maskrcnn.CropFunction(int(ch), int(cw), float(extrap))(img, boxes, ind)
        assert got.device.type == "cpu" and np.array_equal(got.numpy(), z[f"c{i}_crops"]), tag
        crops = torch.zeros_like(img)                       # model.py-shaped caller: __init__.py:36
        maskrcnn._C.crop_forward(img, boxes, ind, float(extrap), int(ch), int(cw), crops)
        assert crops.device.type == "cpu" and np.array_equal(crops.numpy(), z[f"c{i}_crops"]), tag
    # backward on CPU tensors == backward on device tensors (same kernel; the atomics' order may differ: 1e-5 as elsewhere)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 5, 12, 10, generator=g, requires_grad=True)
    boxes = torch.rand(6, 4, generator=g)
    boxes = torch.cat([torch.minimum(boxes[:, :2], boxes[:, 2:]), torch.maximum(boxes[:, :2], boxes[:, 2:])], 1)
    ind = torch.randint(0, 2, (6,), generator=g, dtype=torch.int32)
    out = maskrcnn.CropFunction(3, 4, 0)(img, boxes, ind)
    out.backward(torch.ones_like(out))
    img_d = img.detach().to(dev).requires_grad_(True)
    out_d = maskrcnn.CropFunction(3, 4, 0)(img_d, boxes.to(dev), ind.to(dev))
    out_d.backward(torch.ones_like(out_d))
    assert img.grad.device.type == "cpu" and torch.equal(out.detach(), out_d.detach().cpu())
    assert (img.grad - img_d.grad.cpu()).abs().max().item() <= 1e-5
    # dtype errors are the reference's on either device
    with pytest.raises(RuntimeError):
        maskrcnn._C.crop_forward(img.detach(), boxes, ind.long(), 0.0, 3, 4, torch.zeros(1))


def _rand_boxes(g, n, lo, hi, spill=False):
    c = torch.rand(n, 2, generator=g)
    hw = torch.exp(torch.rand(n, 2, generator=g) * (np.log(hi) - np.log(lo)) + np.log(lo))
    b = torch.cat([c - hw / 2, c + hw / 2], 1)
    return b if spill else b.clamp(0, 1)


def test_crop_forward_random_vs_oracle(dev, oracle):
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(21)
    for (b, c, h, w, n, ch, cw) in [(1, 256, 64, 64, 50, 7, 7), (2, 33, 37, 19, 40, 14, 14),
                                    (3, 8, 128, 96, 64, 28, 28), (1, 4, 9, 9, 30, 1, 1),
                                    (1, 300, 16, 16, 10, 3, 5), (2, 16, 32, 32, 33, 64, 2)]:
        img = torch.randn(b, c, h, w, generator=g)
        boxes = _rand_boxes(g, n, 0.02, 0.9, spill=True)
        ind = torch.randint(0, b, (n,), generator=g, dtype=torch.int32)
        want = oracle.crop_forward(img, boxes, ind, 0.25, ch, cw)
        got = ops.crop(img.to(dev), boxes.to(dev), ind.to(dev), 0.25, ch, cw)
        assert torch.equal(got.cpu(), want), (b, c, h, w, n, ch, cw)


def _bits_equal(a, b):
    """Bit for bit, the sign of a zero included (torch.equal would pass -0.0 == +0.0)."""
    return torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


def test_crop_forward_staged_path_vs_oracle(dev, oracle, monkeypatch):
    """The LDS-staged NCHW kernel (crop sizes with ch*cw % 4 == 0 and <= 256, W % 4 == 0): small and large footprints (the
    latter fall back to the gather path inside the same launch), boxes partly / wholly outside, reversed boxes, samples that
    land exactly on pixels (lo == hi taps, lerp 0), channel counts that leave waves / groups ragged, several images, a bad
    box_index, negative zeros in the image. Bit for bit against the CPU oracle AND against the gather kernel."""
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(77)
    cases = [(2, 70, 64, 64, 40, 14, 14, 0.02, 0.9), (1, 256, 256, 256, 48, 14, 14, 0.02, 0.12),
             (3, 37, 40, 48, 50, 16, 16, 0.02, 0.5), (1, 5, 8, 8, 20, 2, 2, 0.1, 0.9), (1, 9, 12, 20, 20, 4, 1, 0.05, 0.9),
             (1, 130, 32, 32, 30, 8, 8, 0.02, 0.3), (2, 64, 128, 128, 30, 1, 4, 0.02, 0.4), (1, 19, 16, 36, 25, 14, 14, 0.3, 1.5),
             (1, 1, 4, 4, 7, 2, 6, 0.1, 0.9), (1, 256, 32, 32, 64, 14, 14, 0.02, 0.12),
             # plane = 256 crops whose whole-image box takes the in-launch gather fallback: its Tap table (32 B per position)
             # + sample table is LARGER than the default DMA ring — the launch must size its LDS for it (8 x 32: 8832 B)
             (1, 24, 96, 96, 12, 8, 32, 0.3, 0.95), (2, 17, 80, 64, 9, 32, 8, 0.4, 0.95), (1, 33, 72, 72, 8, 16, 16, 0.5, 0.99)]
    for (b, c, h, w, n, ch, cw, lo, hi) in cases:
        img = torch.randn(b, c, h, w, generator=g)
        img[img.abs() < 0.05] = -0.0
        boxes = _rand_boxes(g, n, lo, hi, spill=True)
        boxes[0] = torch.tensor([0.0, 0.0, 1.0, 1.0])                       # the whole image
        boxes[1] = torch.tensor([0.75, 0.9, 0.25, 0.1])                     # reversed
        boxes[2] = torch.tensor([2.0 / (h - 1), 1.0 / (w - 1), (2.0 + ch - 1) / (h - 1), (1.0 + cw - 1) / (w - 1)]) \
            if ch > 1 and cw > 1 else boxes[2]                              # samples exactly on pixels
        boxes[3] = torch.tensor([-0.5, -0.5, -0.1, -0.1])                   # wholly outside
        boxes[4] = torch.tensor([0.5, 0.5, 0.5, 0.5])                       # a single point
        ind = torch.randint(0, b, (n,), generator=g, dtype=torch.int32)
        want = oracle.crop_forward(img, boxes, ind, -1.25, ch, cw)
        ind_bad = ind.clone()
        ind_bad[5] = b + 3
        monkeypatch.setenv("MRCNN_CROP_STAGED", "1")
        got = ops.crop(img.to(dev), boxes.to(dev), ind.to(dev), -1.25, ch, cw)
        got_bad = ops.crop(img.to(dev), boxes.to(dev), ind_bad.to(dev), -1.25, ch, cw)
        monkeypatch.setenv("MRCNN_CROP_STAGED", "0")
        gather = ops.crop(img.to(dev), boxes.to(dev), ind.to(dev), -1.25, ch, cw)
        monkeypatch.delenv("MRCNN_CROP_STAGED")
        assert _bits_equal(got.cpu(), want), (b, c, h, w, n, ch, cw)
        assert _bits_equal(got.cpu(), gather.cpu()), (b, c, h, w, n, ch, cw)
        assert bool((got_bad[5] == -1.25).all()) and _bits_equal(got_bad[6:].cpu(), want[6:])


def test_crop_config2_microbench_shape(dev, oracle):
    """BASELINE config 2: 256 RoIs x 256 ch x 14x14 on P2 (256x256 map), SURVEY §8d inputs."""
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(1234)
    fm = torch.randn(1, 256, 256, 256, generator=g)
    c = torch.rand(256, 2, generator=g)
    hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
    boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1)
    ind = torch.zeros(256, dtype=torch.int32)
    want = oracle.crop_forward(fm, boxes, ind, 0.0, 14, 14)
    got = ops.crop(fm.to(dev), boxes.to(dev), ind.to(dev), 0.0, 14, 14)
    assert torch.equal(got.cpu(), want)


def test_crop_bad_index_and_dtypes(dev):
    from maskrcnn_amd import ops
    img = torch.arange(32, dtype=torch.float32, device=dev).view(1, 2, 4, 4)
    boxes = torch.tensor([[0., 0., 1., 1.]], device=dev)
    out = ops.crop(img, boxes, torch.tensor([3], dtype=torch.int32, device=dev), -7.0, 2, 2)
    assert bool((out == -7.0).all())  # documented GPU behaviour: extrapolation value, no exit()
    with pytest.raises(RuntimeError):
        ops.crop(img, boxes, torch.tensor([0], dtype=torch.int64, device=dev), 0.0, 2, 2)
    with pytest.raises(RuntimeError):
        ops.crop(img.half(), boxes, torch.tensor([0], dtype=torch.int32, device=dev), 0.0, 2, 2)
    with pytest.raises(RuntimeError):
        ops.crop(img.cpu(), boxes.cpu(), torch.tensor([0], dtype=torch.int32), 0.0, 2, 2)


def test_crop_backward_vs_oracle(dev, oracle):
    import maskrcnn
    z = load_golden("crop_backward")
    grads, boxes, ind = (torch.from_numpy(z[k]) for k in ("grads", "boxes", "ind"))
    gi = torch.full(z["grads_image"].shape, 3.0, device=dev)  # must be zeroed by the callee
    maskrcnn._C.crop_backward(grads.to(dev), boxes.to(dev), ind.to(dev), gi)
    np.testing.assert_allclose(gi.cpu().numpy(), z["grads_image"], rtol=0, atol=1e-5)
    # autograd through CropFunction == crop_backward
    g = torch.Generator().manual_seed(5)
    img = torch.randn(2, 6, 10, 12, generator=g).to(dev).requires_grad_(True)
    bx = _rand_boxes(g, 9, 0.1, 0.8, spill=True)
    ix = torch.randint(0, 2, (9,), generator=g, dtype=torch.int32)
    out = maskrcnn.CropFunction(7, 7, 0)(img, bx.to(dev), ix.to(dev))
    go = torch.randn(out.shape, generator=g)
    out.backward(go.to(dev))
    want = oracle.crop_backward(go, bx, ix, img.shape)
    np.testing.assert_allclose(img.grad.cpu().numpy(), want.numpy(), rtol=0, atol=1e-5)


# ------------------------------------------------------------------------------- pyramid RoIAlign
def test_roi_align_pyramid_golden(dev):
    from maskrcnn_amd import ops
    z = load_golden("roi_align")
    fms = [torch.from_numpy(z[f"fm{i}"]).to(dev).permute(0, 2, 3, 1).contiguous() for i in range(4)]
    boxes = torch.from_numpy(z["boxes"]).to(dev)
    area = float(z["image_shape"][0] * z["image_shape"][1])
    for pool in (7, 14):
        got = ops.roi_align_pyramid(fms, boxes, pool, area, rois_per_image=boxes.size(0))
        assert np.array_equal(got.permute(0, 3, 1, 2).cpu().numpy(), z[f"pooled{pool}"])


def test_roi_align_dropin_reference_signature_golden(dev):
    """maskrcnn_amd.roi_align(inputs, pool_size, image_shape) — the reference's own signature (model.py:276): list of
    [boxes [1,N,4]] + four [1,C,H,W] NCHW maps in, [N,C,p,p] out, batch dimensions squeezed in the caller's list — against the
    reference's output on the same inputs (tests/golden/roi_align.npz, generated by the reference's model.roi_align), bit for bit."""
    import maskrcnn_amd
    z = load_golden("roi_align")
    shape = tuple(int(v) for v in z["image_shape"])
    for pool in (7, 14):
        inputs = [torch.from_numpy(z["boxes"]).to(dev).unsqueeze(0)] + [torch.from_numpy(z[f"fm{i}"]).to(dev) for i in range(4)]
        got = maskrcnn_amd.roi_align(inputs, pool, shape)
        assert got.shape == z[f"pooled{pool}"].shape and got.is_contiguous()
        assert np.array_equal(got.cpu().numpy(), z[f"pooled{pool}"])
        assert inputs[0].dim() == 2 and inputs[1].dim() == 3      # squeezed in place, as the reference does (:312-313)
    with pytest.raises(RuntimeError, match="four pyramid levels"):
        maskrcnn_amd.roi_align([torch.zeros(1, 2, 4, device=dev), torch.zeros(1, 8, 4, 4, device=dev)], 7, (64, 64, 3))


def test_roi_align_pyramid_batched_vs_oracle(dev, oracle):
    """1000 proposals/img x 2 images x 256 ch on a 512x512 pyramid vs per-image oracle roi_align."""
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(77)
    B, R, C = 2, 1000, 256
    shape = (512, 512, 3)
    fms = [torch.randn(B, C, 512 // s, 512 // s, generator=g) for s in (4, 8, 16, 32)]
    rois = _rand_boxes(g, B * R, 0.02, 0.6)
    rois[:4] = torch.tensor([[0.1, 0.1, 0.1, 0.5], [0.5, 0.5, 0.4, 0.4], [0., 0., 1., 1.],
                             [0.2, 0.2, 0.2, 0.2]])  # zero / negative area, whole image
    nhwc = [f.to(dev).permute(0, 2, 3, 1).contiguous() for f in fms]
    got, levels = ops.roi_align_pyramid(nhwc, rois.to(dev), 7, float(shape[0] * shape[1]),
                                        rois_per_image=R, return_levels=True)
    got = got.permute(0, 3, 1, 2).cpu()
    for b in range(B):
        r = rois[b * R:(b + 1) * R]
        ok = (r[:, 2] > r[:, 0]) & (r[:, 3] > r[:, 1])  # reference level is UB for area <= 0
        want_lv = oracle.roi_levels(r, shape)
        assert torch.equal(levels[b * R:(b + 1) * R].cpu()[ok], want_lv[ok])
        want = oracle.roi_align(r[ok], [f[b:b + 1] for f in fms], 7, shape)
        assert torch.equal(got[b * R:(b + 1) * R][ok], want)


def test_refine_stages_reference_signature_golden(dev):
    """maskrcnn_amd.refine.rpn_refine / mrn_refine — the reference's MaskRCNN.rpn_refine / mrn_refine signatures on the HIP
    kernels — against the reference's OWN outputs on the same inputs (tests/golden/refine.npz, generated by running the
    reference's methods: model.py:1307-1382, :1389-1487): same number of proposals, rois equal up to the expf ulp of the box
    decode (1e-6 in normalised coordinates), the same detections in the same order with integral boxes equal exactly and
    scores to 1e-6 (the decode kernel re-normalises log(probs))."""
    from maskrcnn_amd import refine
    from maskrcnn_amd.anchors import pyramid_anchors
    from maskrcnn_amd.config import InferenceConfig
    z = load_golden("refine")
    cfg = InferenceConfig(image_height=256, image_width=256)      # the fixture's configuration: 16368 anchors, top 500, 500 rois
    anchors = pyramid_anchors(cfg).to(dev)
    assert tuple(anchors.shape) == tuple(int(v) for v in z["anchors_shape"])
    rois = refine.rpn_refine(torch.from_numpy(z["rpn_class"]).to(dev), torch.from_numpy(z["rpn_bbox"]).to(dev), anchors, cfg)
    want = torch.from_numpy(z["rois"])
    assert rois.shape == want.shape, f"{tuple(rois.shape)} proposals kept, the reference keeps {tuple(want.shape)}"
    # the same SET of rois (the reference orders equal scores as torch.sort happens to, model.py:1345; the kernel by index),
    # each within the expf ulp of the decode; and row by row wherever the scores are distinct
    d = (rois[0].cpu()[:, None, :] - want[0][None, :, :]).abs().amax(dim=2)           # [R, R] pairwise row distance
    assert d.min(dim=1).values.max().item() <= 1e-6 and d.min(dim=0).values.max().item() <= 1e-6
    same_row = (rois[0].cpu() - want[0]).abs().amax(dim=1) <= 1e-6
    assert same_row.float().mean().item() >= 0.95, f"only {same_row.float().mean().item():.3f} of the rows are in the reference's order"
    cls, sc, bx = refine.mrn_refine(want.to(dev), torch.from_numpy(z["probs"]).to(dev), torch.from_numpy(z["deltas"]).to(dev),
                                    tuple(int(v) for v in z["window"]), cfg)
    assert torch.equal(cls.cpu(), torch.from_numpy(z["det_class_ids"]))
    assert torch.equal(bx.cpu(), torch.from_numpy(z["det_boxes"]))
    assert (sc.cpu() - torch.from_numpy(z["det_scores"])).abs().max().item() <= 1e-6
    # nothing kept: background everywhere → the reference's (None, None, None) (model.py:1445-1447)
    bg = torch.zeros_like(torch.from_numpy(z["probs"]))
    bg[:, 0] = 1.0
    bg = (bg + 1e-12) / (1.0 + 81e-12)
    assert refine.mrn_refine(want.to(dev), bg.to(dev), torch.from_numpy(z["deltas"]).to(dev), (0, 0, 256, 256), cfg) == (None, None, None)


# ------------------------------------------------------------------------------------ RPN glue kernels
def test_rpn_scores_deltas_and_proposal_decode(dev, oracle):
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(31)
    B = 2
    sizes = [(16, 12), (8, 6), (4, 3), (2, 2), (1, 1)]
    heads = [torch.randn(B, h, w, 18, generator=g) * 2 for h, w in sizes]
    scores, deltas = ops.rpn_scores_deltas([t.to(dev) for t in heads])
    # reference formulation: permute/view/softmax/cat of model.py:627-641,1294-1304 (heads are already NHWC)
    logits = torch.cat([t[..., :6].reshape(B, -1, 2) for t in heads], 1)
    want_scores = torch.softmax(logits, dim=2)[..., 1]
    want_deltas = torch.cat([t[..., 6:].reshape(B, -1, 4) for t in heads], 1)
    assert torch.equal(deltas.cpu(), want_deltas)
    assert (scores.cpu() - want_scores).abs().max().item() <= 2e-7
    A = want_scores.size(1)
    anchors = torch.rand(A, 4, generator=g) * 200
    anchors[:, 2:] += anchors[:, :2] + 4
    top, order = want_scores.topk(50, dim=1)
    std = [0.1, 0.1, 0.2, 0.2]
    dets = ops.proposal_decode(anchors.to(dev), deltas, order.to(dev), top.to(dev), std, 256, 320).cpu()
    for b in range(B):
        d = oracle.boxes_scale(want_deltas[b][order[b]], std)
        want = oracle.boxes_clamp(oracle.boxes_refine(anchors[order[b]], d), [0, 0, 256, 320])
        assert torch.allclose(dets[b, :, :4], want, rtol=1e-6, atol=1e-4)   # expf ulp differences only
        assert torch.equal(dets[b, :, 4], top[b])


def test_detection_decode_vs_oracle_math(dev, oracle):
    """softmax/argmax + class-specific refine + window clip + round + validity, vs the oracle's box math."""
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(41)
    B, P, C = 2, 300, 81
    logits = torch.randn(B * P, C, generator=g) * 2
    bbox = torch.randn(B * P, C, 4, generator=g) * 0.3
    c = torch.rand(B, P, 2, generator=g)
    hw = torch.rand(B, P, 2, generator=g) * 0.3 + 0.02
    rois = torch.cat([c - hw / 2, c + hw / 2], -1).clamp(0, 1)
    counts = torch.tensor([300, 123], dtype=torch.int32)
    windows = torch.tensor([[0., 0., 512., 640.], [64., 0., 448., 640.]])
    std = [0.1, 0.1, 0.2, 0.2]
    for min_conf in (0.0, 0.3):
        dets, nms_cls, cls = ops.detection_decode(logits.to(dev), bbox.to(dev), rois.to(dev), counts.to(dev),
                                                  windows.to(dev), std, 512, 640, min_conf)
        dets, nms_cls, cls = dets.cpu(), nms_cls.cpu(), cls.cpu()
        probs = torch.softmax(logits, dim=1)
        score, ids = probs.max(dim=1)
        # slots beyond an image's RoI count hold no RoI (round 4: their logits / bbox rows may never have been written — the head
        # skips them — so nothing of them is read): a fixed excluded record instead of the decode of whatever lies there
        live = (torch.arange(P)[None, :] < counts[:, None]).reshape(-1)
        assert torch.equal(cls.view(-1)[live], ids[live]) and bool((cls.view(-1)[~live] == 0).all())
        assert torch.allclose(dets[..., 4].reshape(-1)[live], score[live], rtol=0, atol=1e-6)
        assert bool((dets.reshape(-1, 5)[~live] == 0).all())
        d = bbox[torch.arange(B * P), ids] * torch.tensor(std)
        refined = oracle.boxes_refine(rois.view(-1, 4), d) * torch.tensor([512., 640., 512., 640.])
        for b in range(B):
            n = int(counts[b])
            want = torch.round(oracle.boxes_clamp(refined[b * P:(b + 1) * P], windows[b].tolist()))
            diff = (dets[b, :n, :4] - want[:n]).abs()
            assert float(diff.max()) <= 1.0 and float((diff > 0).float().mean()) < 0.01  # expf ulp at .5 only
            slot = torch.arange(P)
            valid = (ids[b * P:(b + 1) * P] > 0) & (slot < counts[b])
            if min_conf:
                valid &= dets[b, :, 4] >= min_conf
            assert torch.equal(nms_cls[b] > 0, valid)
            assert torch.equal(nms_cls[b][valid].long(), ids[b * P:(b + 1) * P][valid])
            assert nms_cls[b][~valid].unique().numel() == int((~valid).sum())   # unique negatives
        # NaNs in the rows of empty slots (memory the head never wrote) reach nothing
        lg2, bb2 = logits.clone(), bbox.clone()
        lg2[~live] = float("nan")
        bb2[~live] = float("nan")
        dets2, nms2, cls2 = ops.detection_decode(lg2.to(dev), bb2.to(dev), rois.to(dev), counts.to(dev), windows.to(dev), std, 512,
                                                 640, min_conf)
        assert torch.equal(dets2.cpu(), dets) and torch.equal(nms2.cpu(), nms_cls) and torch.equal(cls2.cpu(), cls)


# --------------------------------------------------------------------------------------------------
# selection kernels (csrc/select.hip): exact integer/compare work, compared bit for bit
# --------------------------------------------------------------------------------------------------
def _ref_topk(scores, k):
    """descending score, ties by ascending index — a stable sort of the negated keys (CPU)."""
    order = torch.sort(scores.cpu(), dim=1, descending=True, stable=True).indices[:, :k]
    return scores.cpu().gather(1, order), order


def test_topk_desc(dev):
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(21)
    cases = [(8, 261888, 1000), (2, 261888, 500), (3, 1000, 1000), (1, 5, 3), (2, 70000, 4096), (1, 1, 1),
             (4, 12345, 77)]
    for b, n, k in cases:
        s = torch.rand(b, n, generator=g)
        top, order = ops.topk_desc(s.to(dev), k)
        rt, ro = _ref_topk(s, k)
        assert torch.equal(order.cpu(), ro) and torch.equal(top.cpu(), rt), (b, n, k)
    # heavy ties (saturated scores), negative values, -0.0 / +0.0, infinities
    s = torch.randint(0, 4, (3, 50000), generator=g).float() / 3.0
    s[0, :100] = 1.0
    s[1] = -s[1]
    s[2, ::7] = float("inf")
    s[2, 1::7] = float("-inf")
    s[2, 2::7] = -0.0
    for k in (1, 64, 1000, 4096):
        top, order = ops.topk_desc(s.to(dev), k)
        rt, ro = _ref_topk(s, k)
        # -0.0 and +0.0 compare equal on the CPU sort; the kernel orders -0.0 below +0.0 — compare values, and
        # indices wherever the value is not a zero
        assert torch.equal(top.cpu(), rt), k
        nz = rt != 0
        assert torch.equal(order.cpu()[nz], ro[nz]), k
    s = torch.full((2, 3000), 0.5)                                       # all equal: the first k indices
    top, order = ops.topk_desc(s.to(dev), 1000)
    assert torch.equal(order.cpu(), torch.arange(1000).expand(2, 1000))


def test_proposal_select(dev):
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(22)
    b, k, p = 4, 1000, 600
    dets = torch.rand(b, k, 5, generator=g) * 1000
    keep = torch.full((b, k), -1, dtype=torch.int64)
    counts = torch.tensor([700, 600, 10, 0], dtype=torch.int32)
    for i in range(b):
        n = int(counts[i])
        keep[i, :n] = torch.sort(torch.randperm(k, generator=g)[:n]).values
    rois, cnt = ops.proposal_select(dets.to(dev), keep.to(dev), counts.to(dev), p, 1024, 768)
    norm = torch.tensor([1024.0, 768.0, 1024.0, 768.0])
    assert cnt.cpu().tolist() == [600, 600, 10, 0]
    for i in range(b):
        n = min(int(counts[i]), p)
        ref = torch.zeros(p, 4)
        ref[:n] = dets[i, keep[i, :n], :4] / norm                       # model.py:1367-1374
        assert torch.equal(rois[i].cpu(), ref), i


def test_detection_select(dev):
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(23)
    for (b, p, d) in ((3, 1000, 50), (2, 200, 20), (1, 37, 37), (2, 4096, 100)):
        dets = torch.rand(b, p, 5, generator=g)
        dets[..., :4] = (dets[..., :4] * 900).round()
        dets[0, : p // 2, 4] = dets[0, 0, 4]                              # ties → lowest index first
        class_ids = torch.randint(0, 81, (b, p), generator=g)
        nms_cls = torch.where(class_ids > 0, class_ids, -torch.arange(1, p + 1).expand(b, p)).to(torch.int32)
        keep = torch.full((b, p), -1, dtype=torch.int64)
        counts = torch.zeros(b, dtype=torch.int32)
        for i in range(b):
            n = int(torch.randint(0, p + 1, (1,), generator=g)) if i else p * 3 // 4
            keep[i, :n] = torch.sort(torch.randperm(p, generator=g)[:n]).values
            counts[i] = n
        ids, scores, boxes, rois, cnt = ops.detection_select(dets.to(dev), nms_cls.to(dev), class_ids.to(dev),
                                                             keep.to(dev), counts.to(dev), d, 1024, 1024)
        for i in range(b):
            kept = torch.zeros(p, dtype=torch.bool)
            kept[keep[i, : int(counts[i])]] = True
            cand = (kept & (nms_cls[i] > 0)).nonzero().flatten()
            order = cand[torch.sort(dets[i, cand, 4], descending=True, stable=True).indices][:d]
            n = len(order)
            assert int(cnt[i]) == n
            assert torch.equal(ids[i, :n].cpu(), class_ids[i, order]) and not ids[i, n:].any()
            assert torch.equal(scores[i, :n].cpu(), dets[i, order, 4]) and not scores[i, n:].any()
            assert torch.equal(boxes[i, :n].cpu(), dets[i, order, :4]) and not boxes[i, n:].any()
            assert torch.equal(rois[i, :n].cpu(), dets[i, order, :4] / 1024.0) and not rois[i, n:].any()


def test_device_guard(dev):
    """Every binding runs with its tensors' device current and refuses tensors on different devices (a
    launch on the current device's stream against another device's pointers is a fault or silent xGMI peer traffic)."""
    from maskrcnn_amd import ops
    d = torch.rand(1, 64, 5, device=dev)
    keep, counts = ops.nms_batched(d, 0.5)
    assert keep.device == d.device and counts.device == d.device
    if torch.cuda.device_count() < 2:
        # one GPU on this box: the guard's device comparison is still exercised through its error path
        class Fake:
            pass
        with pytest.raises(RuntimeError, match="Not compiled with CPU support"):
            ops.nms_batched(d.cpu(), 0.5)
        return
    other = torch.device("cuda:1")
    assert torch.cuda.current_device() == 0
    d1 = d.to(other)
    keep1, counts1 = ops.nms_batched(d1, 0.5)          # current device is 0: the guard switches to 1 for the call
    assert keep1.device == other and torch.equal(keep1.cpu(), keep.cpu()) and torch.equal(counts1.cpu(), counts.cpu())
    assert torch.cuda.current_device() == 0
    x = torch.randn(1, 16, 16, 64, device=other)
    w = torch.randn(64, 3, 3, 64, device=other) * 0.05
    y1 = ops.conv3x3_winograd(x, ops.winograd_weights(w), None, None)     # > 64 KB of LDS: per-device attribute
    y0 = ops.conv3x3_winograd(x.to(dev), ops.winograd_weights(w.to(dev)), None, None)
    assert torch.equal(y1.cpu(), y0.cpu())
    with pytest.raises(RuntimeError, match="different devices"):
        ops.conv_bn_act(x, w.to(dev), None, None, 1, (1, 1, 1, 1))


def test_kblocked_outputs_of_roialign_and_maxpool(dev):
    """RoIAlign and max-pool can write the layout the Winograd kernel reads ([C/8][pixels][8]): bit-identical to the
    NHWC result passed through nhwc_to_kblocked (no kblock_kernel launch in the step)."""
    from maskrcnn_amd import ops
    g = torch.Generator().manual_seed(91)
    fms = [torch.randn(2, 128 // s, 160 // s, 64, generator=g).to(dev) for s in (4, 8, 16, 32)]
    rois = _rand_boxes(g, 2 * 37, 0.05, 0.7).to(dev)
    a = ops.roi_align_pyramid(fms, rois, 14, 128.0 * 160.0, rois_per_image=37)
    k = ops.roi_align_pyramid(fms, rois, 14, 128.0 * 160.0, rois_per_image=37, out_kblocked=True)
    assert tuple(k.shape) == (8, 74, 14, 14, 8)
    assert torch.equal(k, ops.nhwc_to_kblocked(a))
    # fp16 NHWC output (the "f16" mode's heads): the fp32 result rounded once, nothing else
    h = ops.roi_align_pyramid(fms, rois, 14, 128.0 * 160.0, rois_per_image=37, out_f16=True)
    assert h.dtype == torch.float16 and torch.equal(h, a.half())
    x = torch.randn(3, 10, 12, 32, generator=g).to(dev)
    for kern, stride, pad in ((1, 2, (0, 0, 0, 0)), (3, 2, (0, 0, 1, 1))):
        assert torch.equal(ops.maxpool(x, kern, stride, pad, out_kblocked=True),
                           ops.nhwc_to_kblocked(ops.maxpool(x, kern, stride, pad)))
