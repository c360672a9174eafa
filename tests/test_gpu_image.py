"""GPU parity (-m gpu) of the image pre-/post-processing kernels (csrc/image.hip) through the C ABI: bit-exact against
the golden vectors (made with Pillow + the reference's own functions) and against the CPU oracle on seeded random
cases, at small and at full (1024 x 1024 canvas) size. Byte/integer work: every comparison is exact; the molded
image is compared exactly too (float(double(u8) - mean) has one correct value)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold():
    return load_golden("image")


@pytest.fixture(scope="module")
def ops():
    from maskrcnn_amd import ops as o
    return o


def test_resize_golden(ops, gold):
    for i, (h, w, c, oh, ow) in enumerate(gold["resize_cases"].tolist()):
        got = ops.resize_bilinear_u8(torch.from_numpy(gold[f"resize_{i}_in"]).to(DEV), oh, ow)
        assert np.array_equal(got.cpu().numpy(), gold[f"resize_{i}_out"]), (i, h, w, c, oh, ow)


def test_resize_random_vs_oracle(ops, oracle):
    rng = np.random.default_rng(11)
    for _ in range(60):
        h, w = (int(v) for v in rng.integers(1, 200, 2))
        oh, ow = (int(v) for v in rng.integers(1, 300, 2))
        c = int(rng.choice([1, 3, 4]))
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        src = a[:, :, 0] if c == 1 else a
        got = ops.resize_bilinear_u8(torch.from_numpy(src).to(DEV), oh, ow).cpu().numpy()
        assert np.array_equal(got, oracle.pil_resize_u8(src, oh, ow)), (h, w, c, oh, ow)


def test_resize_full_size(ops, oracle):
    rng = np.random.default_rng(12)
    for (h, w, oh, ow) in ((480, 640, 768, 1024), (1500, 2000, 768, 1024), (1024, 1024, 1024, 1024)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = ops.resize_bilinear_u8(torch.from_numpy(a).to(DEV), oh, ow).cpu().numpy()
        assert np.array_equal(got, oracle.pil_resize_u8(a, oh, ow)), (h, w, oh, ow)
    a = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)              # resizing to the same size is the identity
    assert np.array_equal(ops.resize_bilinear_u8(torch.from_numpy(a).to(DEV), 64, 64).cpu().numpy(), a)


def test_resize_batched_cropped_view(ops, oracle):
    """[N,H,W] batches and non-contiguous (cropped) views: decode_masks's CenterCrop + Resize without a copy."""
    rng = np.random.default_rng(13)
    a = (rng.random((5, 96, 128)) > 0.5).astype(np.uint8) * 255
    t = torch.from_numpy(a).to(DEV)
    view = t[:, 10:90, 4:100]
    got = ops.resize_bilinear_u8(view, 50, 60).cpu().numpy()
    for i in range(5):
        assert np.array_equal(got[i], oracle.pil_resize_u8(a[i, 10:90, 4:100], 50, 60))


def test_resize_fast_paths_vs_oracle(ops, oracle):
    """Round 5: the bandwidth-shaped resample kernels behind decode_masks — four output columns per thread (horizontal), sixteen
    bytes per thread (vertical), and both passes in ONE launch per 16 x 256 output tile (input footprint and 8-bit intermediate
    in LDS; all-zero footprints stored as zeros) for single-channel enlargements with 16-byte rows — against Pillow's arithmetic
    (the oracle), bit for bit: batches, cropped views, ragged row tiles, sizes that take each of the paths (fused: out_w % 16 == 0 and enlarging; horizontal-fast only: out_w % 4 == 0; generic: the rest)."""
    rng = np.random.default_rng(15)
    for (n, h, w, oh, ow) in ((3, 40, 64, 75, 128), (2, 17, 33, 40, 48), (5, 7, 9, 28, 32), (2, 31, 20, 31, 64), (1, 64, 64, 64, 64),
                              (2, 40, 64, 75, 120), (2, 40, 64, 75, 122), (2, 40, 64, 30, 128), (1, 300, 400, 1000, 1024)):
        a = rng.integers(0, 256, (n, h + 6, w + 10), dtype=np.uint8)
        t = torch.from_numpy(a).to(DEV)
        view = t[:, 3:3 + h, 5:5 + w]                                   # a cropped view: strides, no copy
        got = ops.resize_bilinear_u8(view, oh, ow).cpu().numpy()
        for i in range(n):
            assert np.array_equal(got[i], oracle.pil_resize_u8(a[i, 3:3 + h, 5:5 + w], oh, ow)), (n, h, w, oh, ow, i)
    # binary 0 / 255 masks (what decode_masks resizes) at the bench's size, a few of them
    a = (rng.random((3, 1024, 1024)) > 0.5).astype(np.uint8) * 255
    got = ops.resize_bilinear_u8(torch.from_numpy(a).to(DEV)[:, 192:832, :], 1200, 1920).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], oracle.pil_resize_u8(a[i, 192:832, :], 1200, 1920)), i
    # pasted masks as detect() sees them: zero but for a box (tiles whose input footprint is all zero are stored as zeros without
    # being computed), with box edges and single pixels ON the tile borders of the fused kernel (16 output rows x 256 columns)
    a = np.zeros((4, 1024, 1024), dtype=np.uint8)
    a[0, 300:420, 100:333] = 255
    a[1, 192 + 8:192 + 9, 136:137] = 255                                # one pixel; 136 * 1.875 = 255: the border of column tile 0 / 1
    a[1, 831, 1023] = 200                                              # the last pixel of the window
    a[2, 192:832, :][rng.random((640, 1024)) > 0.999] = 255            # sparse dots: most tiles empty, neighbours of a dot are not
    a[3, 500:501, :] = 1                                               # one faint row
    got = ops.resize_bilinear_u8(torch.from_numpy(a).to(DEV)[:, 192:832, :], 1200, 1920).cpu().numpy()
    for i in range(4):
        assert np.array_equal(got[i], oracle.pil_resize_u8(a[i, 192:832, :], 1200, 1920)), ("sparse", i)


def test_mold_golden(ops, gold):
    from maskrcnn_amd import image as imagelib
    from maskrcnn_amd.config import InferenceConfig
    mean = gold["mean_pixel"].tolist()
    for i, (h, w, min_dim, max_dim) in enumerate(gold["mold_cases"].tolist()):
        cfg = InferenceConfig(image_height=max_dim, image_width=max_dim, image_min_dim=min_dim, image_max_dim=max_dim,
                              mean_pixel=tuple(mean))
        molded, windows, metas = imagelib.mold_inputs([gold[f"mold_{i}_in"]], cfg, DEV)
        assert np.array_equal(molded[0].cpu().numpy(), gold[f"mold_{i}_out"]), i
        assert windows[0].tolist() == gold[f"mold_{i}_window"].tolist()
        assert float(metas[0][0]) == float(gold[f"mold_{i}_scale"])


def test_mold_batch_full_size_vs_oracle(ops, oracle):
    from maskrcnn_amd import image as imagelib
    from maskrcnn_amd.config import InferenceConfig
    cfg = InferenceConfig()                                            # 1024 canvas, min 800 / max 1024
    rng = np.random.default_rng(14)
    images = [rng.integers(0, 256, s, dtype=np.uint8) for s in ((480, 640, 3), (1300, 900, 3), (1024, 1024, 3))]
    molded, windows, metas = imagelib.mold_inputs(images, cfg, DEV)
    for i, img in enumerate(images):
        ref_img, window, scale, padding = oracle.resize_image(img, cfg.image_min_dim, cfg.image_max_dim, True)
        ref = oracle.mold_image(ref_img, cfg.mean_pixel)[0]
        assert tuple(windows[i].tolist()) == tuple(window) and metas[i][0] == scale
        assert torch.equal(molded[i].cpu(), ref), i


def test_resize_image_signature(ops, oracle):
    from maskrcnn_amd import image as imagelib
    rng = np.random.default_rng(15)
    a = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    img, window, scale, padding = imagelib.resize_image(a, min_dim=100, max_dim=128, padding=True, device=DEV)
    ref, rwin, rscale, rpad = oracle.resize_image(a, 100, 128, True)
    assert np.array_equal(img.cpu().numpy(), ref) and tuple(window) == tuple(rwin) and scale == rscale and padding == rpad


def test_full_masks_golden(ops, gold):
    from maskrcnn_amd import image as imagelib
    h, w = gold["fm_canvas"].tolist()
    masks = torch.from_numpy(gold["fm_masks"]).to(DEV)                 # reference layout [N,C,28,28]
    ids = torch.from_numpy(gold["fm_class_id"]).to(DEV)
    boxes = torch.from_numpy(gold["fm_boxes"]).to(DEV)
    got = imagelib.full_masks(ids, boxes, masks, h, w)
    assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), gold["fm_out"])
    nhwc = masks.permute(0, 2, 3, 1).contiguous()                      # this library's mask-head layout
    got2 = imagelib.full_masks(ids, boxes, nhwc, h, w, channels_last=True)
    assert np.array_equal(got2.cpu().numpy(), gold["fm_out"])


def test_decode_golden(ops, gold):
    from maskrcnn_amd import image as imagelib
    h, w = gold["fm_canvas"].tolist()
    window = tuple(gold["dec_window"].tolist())
    l8 = ops.paste_masks(torch.from_numpy(gold["fm_masks"]).to(DEV), torch.from_numpy(gold["fm_class_id"]).to(DEV),
                         torch.from_numpy(gold["fm_boxes"]).to(DEV), h, w, channels_last=False, as_l8=True)
    assert np.array_equal(l8.cpu().numpy(), gold["fm_out"].astype(np.uint8) * 255)
    for tag in ("up", "down"):
        scale = float(gold[f"dec_{tag}_scale"])
        got = imagelib.decode_masks(l8, scale, window)
        assert np.array_equal(got.cpu().numpy(), gold[f"dec_{tag}_masks"]), tag
        b = imagelib.decode_boxes(torch.from_numpy(gold["fm_boxes"]).to(DEV), scale, window)
        assert np.array_equal(b.cpu().numpy(), gold[f"dec_{tag}_boxes"])


def _random_detections(rng, n, c, hh, ww):
    y1 = rng.integers(0, hh - 1, n)
    x1 = rng.integers(0, ww - 1, n)
    bh = np.minimum(rng.integers(1, hh, n) // rng.choice([1, 2, 8, 40], n), hh - y1)
    bw = np.minimum(rng.integers(1, ww, n) // rng.choice([1, 2, 8, 40], n), ww - x1)
    boxes = np.stack([y1, x1, y1 + np.maximum(bh, 1), x1 + np.maximum(bw, 1)], 1).astype(np.float32)
    masks = (1.0 / (1.0 + np.exp(-rng.normal(0, 3, (n, 28, 28, c))))).astype(np.float32)
    return boxes, masks, rng.integers(1, c, n).astype(np.int64)


def test_full_masks_full_size_vs_oracle(ops, oracle):
    """BASELINE sizes: 50 detection slots on a 1024 x 1024 canvas, 81 classes, boxes from 1 pixel to the whole image."""
    rng = np.random.default_rng(16)
    n, c, hh, ww = 50, 81, 1024, 1024
    boxes, masks, ids = _random_detections(rng, n, c, hh, ww)
    boxes[0] = [0, 0, hh, ww]
    boxes[1] = [1023, 1023, 1024, 1024]
    boxes[2] = [100, 200, 100, 260]                                    # empty box: the reference raises; here all-zero
    ids[3] = 0
    boxes[3] = [0, 0, 0, 0]                                            # a padded detection slot
    got = ops.paste_masks(torch.from_numpy(masks).to(DEV), torch.from_numpy(ids).to(DEV),
                          torch.from_numpy(boxes).to(DEV), hh, ww, channels_last=True).cpu()
    ok = [i for i in range(n) if i not in (2, 3)]
    ref = oracle.full_masks(torch.from_numpy(ids[ok]), torch.from_numpy(boxes[ok]),
                            torch.from_numpy(masks[ok]).permute(0, 3, 1, 2), hh, ww)
    assert torch.equal(got[ok], ref)
    assert not got[2].any() and not got[3].any()
    # nothing is ever written outside the box
    for i in ok[:10]:
        y1, x1, y2, x2 = (int(v) for v in boxes[i])
        m = got[i].clone()
        m[y1:y2, x1:x2] = False
        assert not m.any()


@pytest.mark.parametrize("hh,ww", [(832, 1344), (100, 132)])          # BASELINE config 5 canvas; a width % 16 != 0
def test_full_masks_rectangular_canvas(ops, oracle, hh, ww):
    rng = np.random.default_rng(17)
    n, c = 12, 7
    boxes, masks, ids = _random_detections(rng, n, c, hh, ww)
    got = ops.paste_masks(torch.from_numpy(masks).to(DEV), torch.from_numpy(ids).to(DEV),
                          torch.from_numpy(boxes).to(DEV), hh, ww, channels_last=True).cpu()
    ref = oracle.full_masks(torch.from_numpy(ids), torch.from_numpy(boxes),
                            torch.from_numpy(masks).permute(0, 3, 1, 2), hh, ww)
    assert torch.equal(got, ref)


def test_bad_arguments(ops):
    from maskrcnn_amd._lib import MaskrcnnHipError
    with pytest.raises(RuntimeError):
        ops.resize_bilinear_u8(torch.zeros(4, 4, dtype=torch.float32, device=DEV), 2, 2)
    with pytest.raises(RuntimeError):
        ops.resize_bilinear_u8(torch.zeros(4, 4, dtype=torch.uint8), 2, 2)            # CPU tensor
    with pytest.raises(ValueError):
        ops.resize_bilinear_u8(torch.zeros(4, 4, dtype=torch.uint8, device=DEV), 0, 2)
    with pytest.raises(MaskrcnnHipError):
        ops.paste_masks(torch.zeros(1, 28, 28, 3, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV),
                        torch.zeros(1, 4, device=DEV), 64, 66, channels_last=True)    # width % 4 != 0


def test_detect_end_to_end(ops, oracle):
    """detect(): images of different sizes in, per-image results out. The molded batch and the pasted / decoded masks
    are checked against the oracle composed on the pipeline's own detections (the network in between is covered by
    test_gpu_pipeline.py)."""
    from maskrcnn_amd import image as imagelib
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    cfg = InferenceConfig(image_height=256, image_width=256, image_min_dim=200, image_max_dim=256, backbone="resnet50",
                          pre_nms_limit=300, proposal_count=100, detection_max_instances=10)
    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(5)
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_class.bias"] = torch.randn(81, generator=g) * 0.5
    net = MaskRCNNInference(sd, cfg, DEV)
    rng = np.random.default_rng(18)
    images = [rng.integers(0, 256, s, dtype=np.uint8) for s in ((120, 160, 3), (256, 256, 3), (400, 300, 3))]
    results = net.detect(images)
    assert len(results) == 3
    molded, windows, metas = imagelib.mold_inputs(images, cfg, DEV)
    det = net.predict(molded, windows)
    for b, (ids, scores, boxes, masks) in enumerate(results):
        n = int(det.counts[b])
        if n == 0:
            assert ids is None
            continue
        scale, window = metas[b][0], tuple(windows[b].tolist())
        assert torch.equal(ids, det.class_ids[b, :n])
        m28 = det.masks[b, :n].permute(0, 3, 1, 2).cpu()
        bx = det.boxes[b, :n].cpu()
        # random weights can give empty boxes: the reference raises on those (PIL), this library pastes nothing
        ok = ((bx[:, 2] > bx[:, 0]) & (bx[:, 3] > bx[:, 1])).nonzero().flatten()
        full = torch.zeros(n, 256, 256, dtype=torch.bool)
        if len(ok):
            full[ok] = oracle.full_masks(det.class_ids[b, :n].cpu()[ok], bx[ok], m28[ok], 256, 256)
        ref_masks = oracle.decode_masks(full, scale, window)
        assert torch.equal(masks.cpu(), ref_masks), b
        assert torch.equal(boxes.cpu(), oracle.decode_boxes(det.boxes[b, :n].cpu(), scale, window))
        if scale != 1:
            h0, w0 = images[b].shape[:2]
            assert abs(masks.shape[1] - h0) <= 1 and abs(masks.shape[2] - w0) <= 1


def test_detect_batches_images_of_one_size(ops):
    """Round 5: detect() molds, pastes and decodes the images that share a size TOGETHER (one launch per stage and group, over
    the group's valid detections) with one host synchronisation per batch. A batch with two sizes in mixed order (A B A A B)
    must give, image by image, exactly what each image alone gives — and the per-image results are views into the group's
    tensors, never copies per image."""
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    cfg = InferenceConfig(image_height=256, image_width=256, image_min_dim=200, image_max_dim=256, backbone="resnet50",
                          pre_nms_limit=300, proposal_count=100, detection_max_instances=10)
    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(5)
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_class.bias"] = torch.randn(81, generator=g) * 0.5
    net = MaskRCNNInference(sd, cfg, DEV)
    rng = np.random.default_rng(19)
    shapes = ((300, 480, 3), (256, 256, 3), (300, 480, 3), (300, 480, 3), (256, 256, 3))
    images = [rng.integers(0, 256, s, dtype=np.uint8) for s in shapes]
    timings = {}
    results = net.detect(images, timings=timings)
    assert set(timings) == {"mold_ms", "predict_ms", "paste_decode_ms"} and all(v >= 0 for v in timings.values())
    some = 0
    for i, img in enumerate(images):
        alone = net.detect([img])[0]
        if alone[0] is None:
            assert results[i][0] is None
            continue
        some += 1
        for a, c in zip(results[i], alone):
            assert a.dtype == c.dtype and torch.equal(a, c), i
        h0, w0 = img.shape[:2]
        assert abs(results[i][3].shape[1] - h0) <= 1 and abs(results[i][3].shape[2] - w0) <= 1
    assert some >= 3
    # torch tensors (already on the device) are accepted beside numpy arrays
    mixed = [torch.from_numpy(images[0]).to(DEV), images[2]]
    r2 = net.detect(mixed)
    for a, c in zip(r2[1], results[2]):
        assert (a is None and c is None) or torch.equal(a, c)


def test_config1_car_image_mold_and_detect(ops, oracle):
    """BASELINE configs[0] — `predict.py images/car58a54312d.jpg`: the reference's sample image (decoded pixels in the
    fixture) through detect(): 1200 x 1920 → 640 x 1024 → 1024^2 canvas, window (192, 0, 832, 1024) (utils.py:42-90,
    predict.py:55-60). The molded tensor is bit-identical to the golden one (sha256 of the reference's own mold_image
    output); detect() on it (synthetic weights: no checkpoint offline) returns boxes and masks in the ORIGINAL image's
    frame, equal to the oracle's full_masks / decode_masks / decode_boxes composed on the pipeline's detections."""
    import hashlib

    from maskrcnn_amd import image as imagelib
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    z = load_golden("config1")
    a = z["image"]
    cfg = InferenceConfig(image_height=1024, image_width=1024, backbone="resnet50", pre_nms_limit=500, proposal_count=500,
                          image_min_dim=int(z["min_dim"]), image_max_dim=int(z["max_dim"]),
                          mean_pixel=tuple(z["mean_pixel"].tolist()))
    molded, windows, metas = imagelib.mold_inputs([a], cfg, DEV)
    assert windows[0].tolist() == z["window"].tolist() == [192, 0, 832, 1024]
    assert float(metas[0][0]) == float(z["scale"]) and [list(p) for p in metas[0][1]] == z["padding"].tolist()
    m = molded[0].cpu().contiguous()
    assert hashlib.sha256(m.numpy().tobytes()).hexdigest() == str(z["molded_sha256"])
    assert np.array_equal(m.numpy()[:, ::16, ::16], z["molded_sample"])
    resized = imagelib.resize_image(a, cfg.image_min_dim, cfg.image_max_dim, padding=False, device=DEV)[0]
    assert np.array_equal(resized.cpu().numpy(), z["resized"])

    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(5)
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_class.bias"] = torch.randn(81, generator=g) * 0.5
    net = MaskRCNNInference(sd, cfg, DEV)
    (ids, scores, boxes, masks), = net.detect([a])
    det = net.predict(molded, windows)
    n = int(det.counts[0])
    if n == 0:
        assert ids is None
        return
    scale, window = metas[0][0], tuple(windows[0].tolist())
    assert torch.equal(ids, det.class_ids[0, :n])
    bx = det.boxes[0, :n].cpu()
    assert bool((bx[:, 0] >= 192).all() and (bx[:, 2] <= 832).all())          # clipped to the window (model.py:1429)
    m28 = det.masks[0, :n].permute(0, 3, 1, 2).cpu()
    ok = ((bx[:, 2] > bx[:, 0]) & (bx[:, 3] > bx[:, 1])).nonzero().flatten()
    full = torch.zeros(n, 1024, 1024, dtype=torch.bool)
    if len(ok):
        full[ok] = oracle.full_masks(det.class_ids[0, :n].cpu()[ok], bx[ok], m28[ok], 1024, 1024)
    assert torch.equal(masks.cpu(), oracle.decode_masks(full, scale, window))
    assert torch.equal(boxes.cpu(), oracle.decode_boxes(bx, scale, window))
    assert tuple(masks.shape[1:]) == (1200, 1920)                              # back at the original image's size


def test_predict_cli_on_an_image_file(tmp_path, capsys):
    """predict.py — the reference's entry point (`python predict.py [-model m] image`, predict.py:30-72 there): image file in,
    one printed line per detection, full-size masks out. Random weights of the R50 architecture (no checkpoint offline): what
    is checked is the plumbing — the file is read, molded, pushed through detect(), and the printed / saved results agree with
    each other and with the image's size."""
    import importlib.util
    import os
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("predict_cli", os.path.join(root, "predict.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    assert len(cli.COCO_NAMES) == 81
    rng = np.random.default_rng(5)
    path, out = str(tmp_path / "im.jpg"), str(tmp_path / "det.npz")
    Image.fromarray(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)).save(path, quality=95)
    with pytest.raises(SystemExit):
        cli.main([path])                                   # neither -model nor --random-weights: refuses to guess
    res = cli.main(["--random-weights", "--backbone", "resnet50", "--save", out, path])
    printed = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    z = np.load(out)
    n = z["class_ids"].shape[0]
    assert z["masks"].shape == (n, 300, 400) and z["boxes"].shape == (n, 4) and z["scores"].shape == (n,)
    assert len(res) == n and (len(printed) == n if n else printed == ["no instances"])
    if n:
        assert z["masks"].dtype == np.bool_ or z["masks"].dtype == np.uint8
        assert (z["class_ids"] > 0).all() and (z["class_ids"] < 81).all()
        assert (z["boxes"][:, 0] >= 0).all() and (z["boxes"][:, 2] <= 300).all() and (z["boxes"][:, 3] <= 400).all()
        assert (np.diff(z["scores"]) <= 1e-7).all()        # by descending score, as the reference returns them
    # a grey image file is read as RGB (grey2rgb in the reference)
    gpath = str(tmp_path / "g.png")
    Image.fromarray(rng.integers(0, 256, (64, 80), dtype=np.uint8)).save(gpath)
    assert cli.read_image(gpath).shape == (64, 80, 3)
