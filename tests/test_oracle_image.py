"""CPU: the oracle's restatement of the image pre-/post-processing (oracle/resample_ref.c + oracle.py) against the
golden vectors made with Pillow and the reference's own functions (tests/golden/make_golden.py image), against
Pillow live when it is importable, and the host-side bookkeeping of maskrcnn_amd.image against the same vectors."""
import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope="module")
def gold():
    return load_golden("image")


def test_resize_golden(oracle, gold):
    for i, (h, w, c, oh, ow) in enumerate(gold["resize_cases"].tolist()):
        got = oracle.pil_resize_u8(gold[f"resize_{i}_in"], oh, ow)
        assert np.array_equal(got, gold[f"resize_{i}_out"]), (i, h, w, c, oh, ow)


def test_f2l_golden(oracle, gold):
    assert np.array_equal(oracle.f32_to_l8(gold["f2l_in"]), gold["f2l_out"])


def test_resize_image_mold_golden(oracle, gold):
    mean = gold["mean_pixel"]
    for i, (h, w, min_dim, max_dim) in enumerate(gold["mold_cases"].tolist()):
        img, window, scale, padding = oracle.resize_image(gold[f"mold_{i}_in"], min_dim, max_dim, True)
        assert tuple(window) == tuple(gold[f"mold_{i}_window"].tolist())
        assert float(scale) == float(gold[f"mold_{i}_scale"])
        assert np.array_equal(np.array(padding), gold[f"mold_{i}_padding"])
        molded = oracle.mold_image(img, mean)[0].numpy()
        assert np.array_equal(molded, gold[f"mold_{i}_out"]), i


def test_resize_plan_matches_reference_bookkeeping(gold):
    from maskrcnn_amd import image as imagelib
    for i, (h, w, min_dim, max_dim) in enumerate(gold["mold_cases"].tolist()):
        nh, nw, window, scale, padding = imagelib.resize_plan(h, w, min_dim, max_dim, True)
        assert tuple(window) == tuple(gold[f"mold_{i}_window"].tolist())
        assert float(scale) == float(gold[f"mold_{i}_scale"])
        assert np.array_equal(np.array(padding), gold[f"mold_{i}_padding"])
        assert (nh, nw) == (window[2] - window[0], window[3] - window[1])


def test_full_masks_golden(oracle, gold):
    h, w = gold["fm_canvas"].tolist()
    got = oracle.full_masks(torch.from_numpy(gold["fm_class_id"]), torch.from_numpy(gold["fm_boxes"]),
                            torch.from_numpy(gold["fm_masks"]), h, w)
    assert got.dtype == torch.bool and np.array_equal(got.numpy(), gold["fm_out"])
    assert gold["fm_out"].any(axis=(1, 2)).sum() >= 8       # the cases are not degenerate


def test_decode_golden(oracle, gold):
    window = tuple(gold["dec_window"].tolist())
    full = torch.from_numpy(gold["fm_out"])
    boxes = torch.from_numpy(gold["fm_boxes"])
    for tag in ("up", "down"):
        scale = float(gold[f"dec_{tag}_scale"])
        assert np.array_equal(oracle.decode_boxes(boxes, scale, window).numpy(), gold[f"dec_{tag}_boxes"])
        assert np.array_equal(oracle.decode_masks(full, scale, window).numpy(), gold[f"dec_{tag}_masks"])
    from maskrcnn_amd import image as imagelib
    for tag in ("up", "down"):
        scale = float(gold[f"dec_{tag}_scale"])
        assert np.array_equal(imagelib.decode_boxes(boxes, scale, window).numpy(), gold[f"dec_{tag}_boxes"])


def test_resample_against_pillow_live(oracle):
    """The third-party dependency itself, when this environment has it (it does in the build image)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for _ in range(200):
        h, w = (int(v) for v in rng.integers(1, 90, 2))
        oh, ow = (int(v) for v in rng.integers(1, 140, 2))
        c = int(rng.choice([1, 3]))
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        src = a[:, :, 0] if c == 1 else a
        ref = np.array(Image.fromarray(src).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(oracle.pil_resize_u8(src, oh, ow), ref), (h, w, c, oh, ow)
    a = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)          # the usual detect() case: 480x640 -> 768x1024
    assert np.array_equal(oracle.pil_resize_u8(a, 768, 1024), np.array(Image.fromarray(a).resize((1024, 768), Image.BILINEAR)))
    f = (rng.random((64, 64), dtype=np.float32) * 400 - 70).astype(np.float32)
    assert np.array_equal(oracle.f32_to_l8(f), np.array(Image.fromarray(f).convert("L")))


def test_coefficients_sum_to_one(oracle):
    for in_size, out_size in ((28, 100), (28, 5), (1024, 768), (640, 1024), (3, 1)):
        bounds, kk = oracle.resample_coeffs(in_size, out_size)
        assert (np.abs(kk.sum(axis=1) - (1 << 22)) <= kk.shape[1]).all()
        assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= in_size).all()


def test_config1_car_image_geometry(oracle):
    """BASELINE configs[0]: the reference's sample image images/car58a54312d.jpg through detect()'s pre-processing
    (utils.resize_image + mold_image): 1200 x 1920 → 640 x 1024 → 1024^2 canvas, window (192, 0, 832, 1024)."""
    import hashlib

    from maskrcnn_amd import image as imagelib
    z = load_golden("config1")
    a = z["image"]
    assert a.shape == (1200, 1920, 3)
    img, window, scale, padding = oracle.resize_image(a, int(z["min_dim"]), int(z["max_dim"]), True)
    assert tuple(window) == tuple(z["window"].tolist()) == (192, 0, 832, 1024)
    assert float(scale) == float(z["scale"]) and np.array_equal(np.array(padding), z["padding"])
    assert np.array_equal(img[192:832], z["resized"]) and not img[:192].any() and not img[832:].any()
    molded = oracle.mold_image(img, z["mean_pixel"])[0].contiguous()
    assert hashlib.sha256(molded.numpy().tobytes()).hexdigest() == str(z["molded_sha256"])
    assert np.array_equal(molded.numpy()[:, ::16, ::16], z["molded_sample"])
    assert imagelib.resize_plan(1200, 1920, 800, 1024, True)[:4] == (640, 1024, (192, 0, 832, 1024), float(z["scale"]))
