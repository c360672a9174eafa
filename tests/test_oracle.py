"""CPU tests: pin the oracle (oracle/) against the golden vectors generated from the reference itself
(tests/golden/make_golden.py), against oracle/_ref live when present, and against the known answers
written in the reference's pdb comments (utils.py:148-151,248-289; data.py:464-471; model.py:165-166)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden


def _cases(z):
    i = 0
    while f"c{i}_tag" in z.files:
        yield i, str(z[f"c{i}_tag"])
        i += 1


def test_nms_golden(oracle):
    z = load_golden("nms")
    n = 0
    for i, tag in _cases(z):
        dets = torch.from_numpy(z[f"c{i}_dets"])
        keep = oracle.nms(dets, float(z[f"c{i}_thr"]))
        assert keep.dtype == torch.int64
        assert np.array_equal(keep.numpy(), z[f"c{i}_keep"]), tag
        n += 1
    assert n >= 30


def test_nms_strided_and_empty(oracle):
    z = load_golden("nms_strided")
    wide = torch.from_numpy(z["wide"])
    assert np.array_equal(oracle.nms(wide[:, 1:6], float(z["thr"])).numpy(), z["keep"])
    assert oracle.nms(torch.zeros(0, 5), 0.5).numel() == 0
    with pytest.raises(NotImplementedError):
        oracle.nms(torch.zeros(3, 5, dtype=torch.int32), 0.5)


def test_nms_class_aware_equals_per_class_loop(oracle):
    """The single class-aware pass == union of per-class nms calls (model.py:1454-1475)."""
    g = torch.Generator().manual_seed(3)
    n = 400
    b = torch.rand(n, 2, generator=g) * 200
    d = torch.cat([b, b + torch.rand(n, 2, generator=g) * 80 + 5, torch.randperm(n, generator=g)[:, None] / n], 1)
    cls = torch.randint(1, 7, (n,), generator=g, dtype=torch.int32)
    want = []
    for c in cls.unique():
        ix = torch.nonzero(cls == c)[:, 0]
        want.append(ix[oracle.nms(d[ix], 0.3)])
    want = torch.cat(want).sort()[0]
    got = oracle.nms(d, 0.3, class_ids=cls)
    assert torch.equal(got, want)


def test_nms_vs_compiled_reference(oracle, ref_ext):
    g = torch.Generator().manual_seed(11)
    for n in (7, 100, 777):
        c = torch.rand(n, 2, generator=g) * 300
        hw = torch.rand(n, 2, generator=g) * 120 + 1
        d = torch.cat([c, c + hw, torch.randperm(n, generator=g)[:, None].float() / n], 1)
        for thr in (0.1, 0.5, 0.9):
            assert torch.equal(oracle.nms(d, thr), ref_ext.nms(d, thr))
            assert torch.equal(oracle.nms(d.double(), thr), ref_ext.nms(d.double(), thr))


def test_crop_forward_golden(oracle):
    z = load_golden("crop_forward")
    n = 0
    for i, tag in _cases(z):
        extrap, ch, cw = z[f"c{i}_args"]
        got = oracle.crop_forward(torch.from_numpy(z[f"c{i}_image"]), torch.from_numpy(z[f"c{i}_boxes"]),
                                  torch.from_numpy(z[f"c{i}_ind"]), float(extrap), int(ch), int(cw))
        want = z[f"c{i}_crops"]
        assert tuple(got.shape) == want.shape, tag
        assert np.array_equal(got.numpy(), want), tag  # bit-exact: same op order, no FMA contraction
        n += 1
    assert n >= 14


def test_crop_backward_golden(oracle):
    z = load_golden("crop_backward")
    got = oracle.crop_backward(torch.from_numpy(z["grads"]), torch.from_numpy(z["boxes"]),
                               torch.from_numpy(z["ind"]), z["grads_image"].shape)
    assert np.array_equal(got.numpy(), z["grads_image"])


def test_crop_errors(oracle):
    img = torch.zeros(1, 2, 4, 4)
    with pytest.raises(RuntimeError):
        oracle.crop_forward(img, torch.zeros(1, 4), torch.tensor([1], dtype=torch.int32), 0.0, 2, 2)
    with pytest.raises(RuntimeError):
        oracle.crop_forward(img, torch.zeros(1, 4), torch.tensor([0], dtype=torch.int64), 0.0, 2, 2)
    with pytest.raises(RuntimeError):
        oracle.crop_forward(img.half(), torch.zeros(1, 4), torch.tensor([0], dtype=torch.int32), 0.0, 2, 2)


def test_roi_align_golden(oracle):
    z = load_golden("roi_align")
    fms = [torch.from_numpy(z[f"fm{i}"]) for i in range(4)]
    boxes = torch.from_numpy(z["boxes"])
    for pool in (7, 14):
        got = oracle.roi_align(boxes, fms, pool, z["image_shape"])
        assert np.array_equal(got.numpy(), z[f"pooled{pool}"])


def test_roi_levels_reference_sweep(oracle):
    """tests/golden/roi_levels.npz = the level the reference's own model.roi_align assigned (build container) to every
    box of the +-4 / +-16 / +-64 ulp sweeps around the k = 2.5 / 3.5 / 4.5 boundaries + 400 ordinary boxes, both image
    shapes. (a) the formula of model.py:331-338 with every fp32 operation correctly rounded — machine-independent, and
    what the HIP kernel evaluates — equals the reference on EVERY box; (b) the oracle's torch expression (this host's
    MKL log2) equals it on every ordinary box, and is at most one level off on boundary boxes (identical in the build
    container; VML's last bit differs between CPUs)."""
    z = load_golden("roi_levels")
    for hh, ww in ((1024, 1024), (832, 1344)):
        r, ref = z[f"rois_{hh}x{ww}"], z[f"levels_{hh}x{ww}"]
        area = float(hh * ww)
        h32, w32 = (r[:, 2] - r[:, 0]).astype(np.float32), (r[:, 3] - r[:, 1]).astype(np.float32)
        hw32 = (h32 * w32).astype(np.float32)
        denom = (np.float64(224.0) / np.sqrt(np.float64(np.float32(area))).astype(np.float32)).astype(np.float32)
        with np.errstate(divide="ignore"):
            ratio = (np.sqrt(hw32.astype(np.float64)).astype(np.float32).astype(np.float64) / np.float64(denom)).astype(np.float32)
            k = (np.float32(4.0) + np.log2(ratio.astype(np.float64)).astype(np.float32)).astype(np.float32)
        k = np.where(np.isfinite(k), k, np.float32(-100.0))
        exact = np.clip(np.rint(k), 2, 5).astype(np.int32)
        assert np.array_equal(exact, ref), f"{(exact != ref).sum()} boxes: reference != correctly rounded formula"
        got = oracle.roi_levels(torch.from_numpy(r), (hh, ww, 3)).numpy()
        assert np.array_equal(got[-400:], ref[-400:])
        assert np.abs(got - ref).max() <= 1 and (got != ref).mean() <= 0.15
        assert all((ref == l).sum() > 100 for l in (2, 3, 4, 5))


def test_anchors_known_answers(oracle):
    z = load_golden("anchors_boxes")
    cfg = oracle.Cfg()
    a = oracle.create_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, cfg.BACKBONE_SHAPES,
                                      cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
    assert a.shape == (261888, 4) == tuple(z["shape"])  # model.py:1019, utils.py:288-289
    assert np.array_equal(a[z["idx"]], z["rows_f64"])
    assert hashlib.sha256(a.astype(np.float32).tobytes()).hexdigest() == str(z["sha256_f32"])
    # pdb-comment known answers, independent of our oracle build
    np.testing.assert_allclose(a[:3], [[-22.627417, -11.3137085, 22.627417, 11.3137085],
                                       [-16., -16., 16., 16.],
                                       [-11.3137085, -22.627417, 11.3137085, 22.627417]], rtol=0, atol=1e-6)
    np.testing.assert_allclose(a[-3:], [[597.96132803, 778.98066402, 1322.03867197, 1141.01933598],
                                        [704., 704., 1216., 1216.],
                                        [778.98066402, 597.96132803, 1141.01933598, 1322.03867197]],
                               rtol=0, atol=1e-6)
    counts = [len(oracle.create_anchors(s, cfg.RPN_ANCHOR_RATIOS, sh, st, 1)) for s, sh, st in
              zip(cfg.RPN_ANCHOR_SCALES, cfg.BACKBONE_SHAPES, cfg.BACKBONE_STRIDES)]
    assert counts == [196608, 49152, 12288, 3072, 768]  # utils.py:248-285


def test_box_math_golden(oracle):
    z = load_golden("anchors_boxes")
    boxes, deltas = torch.from_numpy(z["boxes"]), torch.from_numpy(z["deltas"])
    refined = oracle.boxes_refine(boxes, deltas)
    assert np.array_equal(refined.numpy(), z["refined"])
    assert np.array_equal(oracle.boxes_clamp(refined, [10, 20, 300, 350]).numpy(), z["clamped"])
    assert np.array_equal(oracle.boxes_scale(refined, [0.1, 0.1, 0.2, 0.2]).numpy(), z["scaled"])


def _sd(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def test_graph_small_golden(oracle):
    z = load_golden("graph_small")
    for i in range(int(z["n_bottleneck"])):
        sd = {"blk." + k: v for k, v in _sd(z, f"b{i}_sd_").items()}
        y = oracle.bottleneck(torch.from_numpy(z[f"b{i}_x"]), sd, "blk", int(z[f"b{i}_stride"]))
        assert np.array_equal(y.numpy(), z[f"b{i}_y"]), i
    for k, s, h, w, top, bottom, left, right in z["same_pad"]:
        y = oracle.same_pad(torch.ones(1, 1, int(h), int(w)), int(k), int(s))
        nz = torch.nonzero(y[0, 0])
        assert (int(nz[:, 0].min()), int(nz[:, 1].min())) == (top, left)
        assert (y.size(2) - h - top, y.size(3) - w - left) == (bottom, right)
    # the (0,1,0,1) stem-maxpool pad, SURVEY Appendix A.12
    assert [int(v) for v in z["same_pad"][1][4:]] == [0, 1, 0, 1]
    sd = {"rpn." + k: v for k, v in _sd(z, "rpn_sd_").items()}
    logits, probs, bbox = oracle.rpn_forward(torch.from_numpy(z["rpn_x"]), sd)
    assert np.array_equal(logits.numpy(), z["rpn_logits"])
    assert np.array_equal(probs.numpy(), z["rpn_probs"])
    assert np.array_equal(bbox.numpy(), z["rpn_bbox"])


def test_refine_golden(oracle):
    z = load_golden("refine")
    cfg = oracle.Cfg(256, 256)
    anchors = oracle.anchors_for(cfg)
    assert tuple(anchors.shape) == tuple(z["anchors_shape"])
    rois, dets = oracle.rpn_refine(torch.from_numpy(z["rpn_class"]), torch.from_numpy(z["rpn_bbox"]),
                                   anchors, cfg, return_dets=True)
    assert np.array_equal(dets.numpy(), z["rpn_dets"])
    assert np.array_equal(oracle.nms(dets, cfg.RPN_NMS_THRESHOLD).numpy(), z["rpn_keep"])
    assert np.array_equal(rois.numpy(), z["rois"])
    cls, sc, bx, log = oracle.mrn_refine(torch.from_numpy(z["rois"]), torch.from_numpy(z["probs"]),
                                         torch.from_numpy(z["deltas"]), tuple(z["window"]), cfg,
                                         return_dets=True)
    assert len(log) == int(z["n_class_calls"])
    for i, (_, d, k) in enumerate(log):
        assert np.array_equal(d.numpy(), z[f"cls{i}_dets"])
        assert np.array_equal(k.numpy(), z[f"cls{i}_keep"])
    assert np.array_equal(cls.numpy(), z["det_class_ids"])
    assert np.array_equal(sc.numpy(), z["det_scores"])
    assert np.array_equal(bx.numpy(), z["det_boxes"])
