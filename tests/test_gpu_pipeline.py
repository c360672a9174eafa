"""GPU parity tests (-m gpu) of the whole inference step against the CPU oracle (oracle.predict and its
stages), on a reduced configuration the oracle finishes in seconds (256x256 images, ResNet-50-FPN,
synthetic weights). Stage by stage, each stage fed with the HIP path's own upstream output so that a
1-ulp exp() difference upstream cannot flip a downstream discrete decision:
  trunk activations  : <= 1e-4 * max(1, max|activation|)  (fp32 MFMA vs CPU summation order, ~60 layers)
  NMS keep sets      : bit-exact on the dets actually handed to NMS
  detections         : identical class ids / boxes / scores (scores 1e-6)
  masks              : <= 1e-4 abs
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["f32", "f16x3", "f32+f16x3"])
def setup(request):
    """Both contraction modes are held to the SAME bars: "f16x3" (fp16-operand MFMA, error-compensated
    3-product split) claims fp32-grade accuracy, so it has to pass the fp32 path's tests unchanged."""
    precision = request.param
    assert torch.cuda.is_available()
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    dev = torch.device("cuda:0")
    cfg = InferenceConfig(image_height=256, image_width=256, backbone="resnet50", pre_nms_limit=300,
                          proposal_count=200, detection_max_instances=20)
    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    # give the heads some spread so that several classes / detections appear with random weights
    g = torch.Generator().manual_seed(5)
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_class.bias"] = torch.randn(81, generator=g) * 0.5
    sd["classifier.linear_bbox.weight"] = torch.randn(324, 1024, generator=g) * 0.02
    sd["rpn.conv_bbox.bias"] = torch.randn(12, generator=g) * 0.3
    b = 2
    images = torch.randint(0, 256, (b, 256, 256, 3), generator=g).float() - torch.tensor(cfg.mean_pixel)
    images = images.permute(0, 3, 1, 2).contiguous()
    windows = torch.tensor([[0., 0., 256., 256.], [32., 0., 224., 256.]])
    # Random weights saturate the softmaxes (all scores == 1.0 → order undefined) and blow the box deltas
    # up. Calibrate the four head layers so scores are distinct and boxes sane; this only rescales the
    # synthetic weights, both paths then see the same state_dict.
    for _ in range(8):
        net = MaskRCNNInference(sd, cfg, dev, precision=precision)
        det, mid = net.predict(images.to(dev), windows.to(dev), with_masks=True, return_intermediates=True)
        sc = mid["rpn_scores"].double().clamp(1e-7, 1 - 1e-7)
        sat = torch.log(sc / (1 - sc)).std().item()  # std of the fg-bg logit difference
        dstd, lstd, bstd = mid["rpn_deltas"].std().item(), mid["logits"].std().item(), mid["bbox"].std().item()
        done = True
        if sat > 2.0:
            sd["rpn.conv_class.weight"] = sd["rpn.conv_class.weight"] * (1.0 / sat)
            done = False
        if dstd > 1.0:
            sd["rpn.conv_bbox.weight"] = sd["rpn.conv_bbox.weight"] * (0.5 / dstd)
            done = False
        if lstd > 3.0:
            sd["classifier.linear_class.weight"] = sd["classifier.linear_class.weight"] * (2.0 / lstd)
            done = False
        if bstd > 1.0:
            sd["classifier.linear_bbox.weight"] = sd["classifier.linear_bbox.weight"] * (0.5 / bstd)
            done = False
        if done:
            break
    torch.cuda.synchronize()
    return dict(cfg=cfg, sd=sd, net=net, images=images, windows=windows, det=det, mid=mid, b=b)


def _ocfg(oracle, cfg):
    return oracle.Cfg(cfg.image_height, cfg.image_width, PRE_NMS_LIMIT=cfg.pre_nms_limit,
                      RPN_NMS_MAX_ROIS_NUM=cfg.proposal_count,
                      DETECTION_MAX_INSTANCES=cfg.detection_max_instances)


def test_trunk_and_rpn_activations(setup, oracle):
    s = setup
    for b in range(s["b"]):
        fms = oracle.fpn_forward(s["images"][b:b + 1], s["sd"], "resnet50")
        for lvl, (want, got) in enumerate(zip(fms, s["mid"]["feature_maps"])):
            got = got[b].permute(2, 0, 1).cpu()
            tol = 1e-4 * max(1.0, want.abs().max().item())
            err = (got - want[0]).abs().max().item()
            assert err <= tol, f"P{lvl + 2}: max|err| {err:.3e} > {tol:.3e} (max|act| {want.abs().max().item():.1f})"
        _, rpn_class, rpn_bbox = oracle.rpn_detect(fms, s["sd"])
        assert rpn_class.shape[1] == s["mid"]["rpn_scores"].shape[1]
        assert (s["mid"]["rpn_scores"][b].cpu() - rpn_class[0, :, 1]).abs().max().item() <= 1e-4
        tol = 1e-4 * max(1.0, rpn_bbox.abs().max().item())
        assert (s["mid"]["rpn_deltas"][b].cpu() - rpn_bbox[0]).abs().max().item() <= tol


def test_trunk_unit_scale_input_abs_1e4(setup, oracle):
    """The 1e-4 bar of BASELINE.json is ABSOLUTE: on an image scaled to unit range (pixels / 128) every pyramid level of
    the ~60-layer trunk is within 1e-4 abs of the oracle, in the exact-fp32 mode and in the f16x3 mode alike. (With
    uint8-range pixels the activations reach ~200 and the same relative error is ~3e-4 abs: test above.)"""
    s = setup
    x = (s["images"] / 128.0).contiguous()
    got = s["net"].backbone(x.to(s["net"].device))
    torch.cuda.synchronize()
    for b in range(s["b"]):
        want = oracle.fpn_forward(x[b:b + 1], s["sd"], "resnet50")
        for lvl, (w_, g_) in enumerate(zip(want, got)):
            err = (g_[b].permute(2, 0, 1).cpu() - w_[0]).abs().max().item()
            assert err <= 1e-4, f"P{lvl + 2}: max|err| {err:.3e} > 1e-4 abs (max|act| {w_.abs().max().item():.2f})"


def test_proposals_stage(setup, oracle):
    s = setup
    ocfg = _ocfg(oracle, s["cfg"])
    anchors = oracle.anchors_for(ocfg)
    assert torch.equal(anchors, s["net"].anchors.cpu())
    for b in range(s["b"]):
        scores = s["mid"]["rpn_scores"][b].cpu()
        rpn_class = torch.stack([1 - scores, scores], 1).unsqueeze(0)
        rois, dets = oracle.rpn_refine(rpn_class, s["mid"]["rpn_deltas"][b].cpu().unsqueeze(0), anchors,
                                       ocfg, return_dets=True)
        got_dets = s["mid"]["rpn_dets"][b].cpu()
        assert torch.allclose(got_dets, dets, rtol=0, atol=1e-3)  # exp() ulp differences only
        # bit-exact NMS on the dets the HIP path actually used
        keep = oracle.nms(got_dets, ocfg.RPN_NMS_THRESHOLD)[:ocfg.RPN_NMS_MAX_ROIS_NUM]
        n = int(s["mid"]["roi_counts"][b])
        assert n == keep.numel()
        want = got_dets[keep, :4] / torch.tensor([256., 256., 256., 256.])
        assert torch.equal(s["mid"]["rois"][b, :n].cpu(), want)
        assert bool((s["mid"]["rois"][b, n:] == 0).all())


def test_classifier_and_detections_stage(setup, oracle):
    s = setup
    ocfg = _ocfg(oracle, s["cfg"])
    p = s["mid"]["rois"].size(1)
    total = 0
    for b in range(s["b"]):
        n = int(s["mid"]["roi_counts"][b])
        rois = s["mid"]["rois"][b, :n].cpu()
        fms = [f[b:b + 1].permute(0, 3, 1, 2).cpu().contiguous() for f in s["mid"]["feature_maps"][:4]]
        logits, probs, bbox = oracle.classifier_forward(fms, rois, s["sd"], ocfg)
        got_logits = s["mid"]["logits"][b * p:b * p + n].cpu()
        got_bbox = s["mid"]["bbox"][b * p:b * p + n].cpu()
        assert (got_logits - logits).abs().max().item() <= 1e-4 * max(1.0, logits.abs().max().item())
        assert (got_bbox - bbox).abs().max().item() <= 1e-4 * max(1.0, bbox.abs().max().item())
        # detections from the HIP path's own head outputs
        gp = torch.softmax(got_logits, dim=1)
        cls, sc, bx = oracle.mrn_refine(rois, gp, got_bbox, tuple(s["windows"][b].tolist()), ocfg)
        k = int(s["det"].counts[b])
        if cls is None:
            assert k == 0
            continue
        assert k == cls.size(1)
        total += k
        assert torch.equal(s["det"].class_ids[b, :k].cpu(), cls[0])
        assert torch.equal(s["det"].boxes[b, :k].cpu(), bx[0])
        assert torch.allclose(s["det"].scores[b, :k].cpu(), sc[0], rtol=0, atol=1e-6)
        assert bool((s["det"].class_ids[b, k:] == 0).all())
    assert total > 0, "test configuration produced no detections"


def test_mask_stage(setup, oracle):
    s = setup
    ocfg = _ocfg(oracle, s["cfg"])
    for b in range(s["b"]):
        k = int(s["det"].counts[b])
        if k == 0:
            continue
        fms = [f[b:b + 1].permute(0, 3, 1, 2).cpu().contiguous() for f in s["mid"]["feature_maps"][:4]]
        boxes = s["det"].boxes[b, :k].cpu()
        want = oracle.mask_forward(fms, boxes / 256.0, s["sd"], ocfg)        # [k,81,28,28]
        got = s["det"].masks[b, :k].permute(0, 3, 1, 2).cpu()
        assert (got - want).abs().max().item() <= 1e-4


def test_batch_independence_and_determinism(setup):
    """Images are independent (frozen BN): running image 1 alone reproduces its slice of the batch."""
    s = setup
    dev = s["net"].device
    det = s["net"].predict(s["images"][1:2].to(dev), s["windows"][1:2].to(dev))
    k = int(det.counts[0])
    assert k == int(s["det"].counts[1])
    assert torch.equal(det.class_ids[0], s["det"].class_ids[1])
    assert torch.equal(det.boxes[0], s["det"].boxes[1])
    assert torch.equal(det.scores[0], s["det"].scores[1])


# ------------------------------------------------------------------------------------------------------
# non-square images, ResNet-101 (BASELINE config 5 shape class), hipGraph capture
# ------------------------------------------------------------------------------------------------------
def _small_net(precision, h, w, arch, seed=11):
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    dev = torch.device("cuda:0")
    cfg = InferenceConfig(image_height=h, image_width=w, backbone=arch, pre_nms_limit=200,
                          proposal_count=100, detection_max_instances=10)
    sd = modules.synthetic_state_dict(arch, seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(seed)
    images = (torch.randint(0, 256, (1, h, w, 3), generator=g).float() - torch.tensor(cfg.mean_pixel))
    images = images.permute(0, 3, 1, 2).contiguous()
    windows = torch.tensor([[0., 0., float(h), float(w)]])
    return cfg, sd, MaskRCNNInference(sd, cfg, dev, precision=precision), images, windows, dev


def test_config5_shape_class_r101_nonsquare(oracle):
    """ResNet-101-FPN on a non-square image (the 832x1344 class of BASELINE config 5, reduced to 192x320):
    exact-fp32 and f16x3 trunks meet the fp32 bar; the plain-fp16 MFMA path ("fp16 MFMA path" of config 5)
    is held to its own stated tolerance, 2e-2 of the activation range (11 significand bits, ~100 layers)."""
    want = None
    for precision, rel in (("f32", 1e-4), ("f16x3", 1e-4), ("f32+f16x3", 1e-4), ("f16", 2e-2)):
        cfg, sd, net, images, windows, dev = _small_net(precision, 192, 320, "resnet101")
        if want is None:
            want = oracle.fpn_forward(images, sd, "resnet101")
            assert [tuple(f.shape[2:]) for f in want] == [(48, 80), (24, 40), (12, 20), (6, 10), (3, 5)]
        det, mid = net.predict(images.to(dev), windows.to(dev), return_intermediates=True)
        for lvl, (w_, g_) in enumerate(zip(want, mid["feature_maps"])):
            err = (g_[0].permute(2, 0, 1).cpu() - w_[0]).abs().max().item()
            tol = rel * max(1.0, w_.abs().max().item())
            assert err <= tol, f"{precision} P{lvl + 2}: {err:.3e} > {tol:.3e}"
        assert tuple(mid["rpn_scores"].shape) == (1, 3 * (48 * 80 + 24 * 40 + 12 * 20 + 6 * 10 + 3 * 5))
        assert tuple(det.boxes.shape) == (1, 10, 4) and tuple(det.masks.shape) == (1, 10, 28, 28, 81)
        assert bool((det.boxes[..., 2] <= 192).all()) and bool((det.boxes[..., 3] <= 320).all())


def test_step_is_hipgraph_capturable():
    """No allocation-dependent control flow, no host sync: the whole step captures into a hipGraph and the
    replay reproduces the eager result bit for bit."""
    cfg, sd, net, images, windows, dev = _small_net("f32", 128, 128, "resnet50")
    images, windows = images.to(dev), windows.to(dev)
    eager = net.predict(images, windows)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            net.predict(images, windows)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = net.predict(images, windows)
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured.class_ids, eager.class_ids)
    assert torch.equal(captured.boxes, eager.boxes)
    assert torch.equal(captured.scores, eager.scores)
    assert torch.equal(captured.masks, eager.masks)


@pytest.mark.parametrize("precision", ["f16", "f32"])
def test_concurrent_sub_batches_give_the_same_tensors(precision):
    """predict() with concurrent_sub_batches=2 (the "f16" mode's default: the batch as two halves on two HIP streams, joined before
    it returns) returns what one batch returns, bit for bit — detections and masks —, call after call (the side streams' tensors
    are handed to the caller's stream), also with injected proposals; an odd batch, a per-launch profiling pass and a hipGraph
    capture run the batch whole."""
    from maskrcnn_amd import modules, ops
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    dev = torch.device("cuda:0")
    h, w = 128, 192
    cfg = InferenceConfig(image_height=h, image_width=w, backbone="resnet50", pre_nms_limit=200, proposal_count=100,
                          detection_max_instances=10)
    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(21)
    images = (torch.randint(0, 256, (4, h, w, 3), generator=g).float() - torch.tensor(cfg.mean_pixel)).permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0., 0., float(h), float(w)]] * 4, device=dev)
    one = MaskRCNNInference(sd, cfg, dev, precision=precision, concurrent_sub_batches=1)
    two = MaskRCNNInference(sd, cfg, dev, precision=precision, concurrent_sub_batches=2)
    assert MaskRCNNInference(sd, cfg, dev, precision=precision).sub_batches == (2 if precision == "f16" else 1)
    want = one.predict(images, windows)
    for _ in range(3):
        got = two.predict(images, windows)
        for f in ("class_ids", "scores", "boxes", "counts", "masks"):
            assert torch.equal(getattr(got, f), getattr(want, f)), f
    assert len(two._side_streams) == 2
    got3 = two.predict(images[:3], windows[:3])                            # odd batch: whole
    assert torch.equal(got3.masks, want.masks[:3]) and torch.equal(got3.boxes, want.boxes[:3])
    ops.CONV_PROFILE = []
    try:
        two.predict(images, windows)
        n_split = len(ops.CONV_PROFILE)
        ops.CONV_PROFILE = []
        one.predict(images, windows)
        assert n_split == len(ops.CONV_PROFILE)                            # profiled whole, launch for launch
    finally:
        ops.CONV_PROFILE = None


def test_full_size_trunk_and_roialign_one_image(oracle):
    """BASELINE full size (1024x1024, ResNet-50-FPN, 1000 RoIs), one image: the CPU oracle needs a few seconds.
    Trunk activations within 1e-4 of the activation range; pyramid RoIAlign of 1000 seeded RoIs on the HIP
    path's own feature maps bit-exact with the oracle's roi_align on the same maps."""
    from maskrcnn_amd import ops
    cfg, sd, net, images, windows, dev = _small_net("f32", 1024, 1024, "resnet50", seed=0)
    fms = net.backbone(images.to(dev))
    torch.cuda.synchronize()
    want = oracle.fpn_forward(images, sd, "resnet50")
    assert [tuple(f.shape) for f in want] == [(1, 256, 256, 256), (1, 256, 128, 128), (1, 256, 64, 64),
                                              (1, 256, 32, 32), (1, 256, 16, 16)]   # model.py:165-166
    for lvl, (w_, g_) in enumerate(zip(want, fms)):
        err = (g_[0].permute(2, 0, 1).cpu() - w_[0]).abs().max().item()
        tol = 1e-4 * max(1.0, w_.abs().max().item())
        assert err <= tol, f"P{lvl + 2}: {err:.3e} > {tol:.3e}"
    g = torch.Generator().manual_seed(1234)  # SURVEY §8d proposal injection recipe
    c = torch.rand(1000, 2, generator=g)
    hw = torch.exp(torch.rand(1000, 2, generator=g) * (math.log(0.6) - math.log(0.02)) + math.log(0.02))
    rois = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1)
    got = ops.roi_align_pyramid(fms[:4], rois.to(dev), 7, 1024.0 * 1024.0, rois_per_image=1000)
    maps_cpu = [f.permute(0, 3, 1, 2).cpu().contiguous() for f in fms[:4]]
    ok = (rois[:, 2] > rois[:, 0]) & (rois[:, 3] > rois[:, 1])
    want_p = oracle.roi_align(rois[ok], maps_cpu, 7, (1024, 1024, 3))
    assert torch.equal(got.permute(0, 3, 1, 2).cpu()[ok], want_p)


def test_head_skips_empty_roi_slots_without_changing_a_valid_row(monkeypatch):
    """Round 4: RoIAlign 7x7 and the classifier's three GEMMs skip the RoI slots beyond each image's proposal count (the
    reference's rois tensor holds only the boxes NMS kept, model.py:1366-1374; this pipeline has proposal_count slots + a count).
    Rows are independent in those kernels: every valid row, the detections and the masks are bit-identical with the skip on and
    off — also when the skipped rows' memory held NaNs before (nothing of an empty slot is ever read)."""
    from maskrcnn_amd import modules, ops
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    dev = torch.device("cuda:0")
    cfg = InferenceConfig(image_height=256, image_width=256, backbone="resnet50", pre_nms_limit=1000, proposal_count=1000,
                          detection_max_instances=20, rpn_nms_threshold=0.3)   # a stricter NMS: several hundred empty slots
    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(11)
    sd["rpn.conv_class.weight"] = sd["rpn.conv_class.weight"] * 0.05
    sd["rpn.conv_bbox.weight"] = sd["rpn.conv_bbox.weight"] * 0.05
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_bbox.weight"] = torch.randn(324, 1024, generator=g) * 0.02
    images = (torch.randint(0, 256, (3, 256, 256, 3), generator=g).float() - torch.tensor(cfg.mean_pixel))
    images = images.permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0., 0., 256., 256.]] * 3, device=dev)
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(modules, "SKIP_EMPTY_ROI_TILES", on)
        net = MaskRCNNInference(sd, cfg, dev)
        if on:   # poison the allocator's free blocks: what the skipped rows will be "left untouched" as
            junk = [torch.full((1000 * 3, 1024), float("nan"), device=dev) for _ in range(4)]
            del junk
        outs[on] = net.predict(images, windows, return_intermediates=True)
        torch.cuda.synchronize()
    (d1, m1), (d0, m0) = outs[True], outs[False]
    counts = m1["roi_counts"].tolist()
    # 256 empty slots in a row always contain a whole 128-row tile
    assert counts == m0["roi_counts"].tolist() and min(counts) <= 1000 - 256, counts
    p = m1["rois"].size(1)
    for b, n in enumerate(counts):
        assert torch.equal(m1["logits"][b * p:b * p + n], m0["logits"][b * p:b * p + n])
        assert torch.equal(m1["bbox"][b * p:b * p + n], m0["bbox"][b * p:b * p + n])
        assert bool((m1["logits"][b * p + n:(b + 1) * p] == 0).all())
    assert torch.equal(d1.class_ids, d0.class_ids) and torch.equal(d1.boxes, d0.boxes) and torch.equal(d1.scores, d0.scores)
    assert torch.equal(d1.counts, d0.counts) and torch.equal(d1.masks, d0.masks)
    assert int(d1.counts.sum()) > 0 and not bool(torch.isnan(d1.masks).any())


def test_predict_splits_oversized_batches_into_equal_sub_batches():
    """pipeline.max_batch_per_launch: past it (31 images at 1024^2) a tensor would exceed the kernels' 2^30-element limit and a layer
    would fall to another kernel, i.e. image i of a batch would no longer equal image i alone. predict() runs such a batch as
    equal sub-batches instead: forced here with a limit of 2 on five images — every output equals the one-launch result bit for
    bit, and return_intermediates (whose tensors cannot be concatenated across sub-batches) is refused."""
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    dev = torch.device("cuda:0")
    cfg = InferenceConfig(image_height=128, image_width=192, backbone="resnet50", pre_nms_limit=200, proposal_count=100,
                          detection_max_instances=10)
    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(4)
    sd["rpn.conv_class.weight"] = sd["rpn.conv_class.weight"] * 0.05
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_bbox.weight"] = torch.randn(324, 1024, generator=g) * 0.02
    images = (torch.randint(0, 256, (5, 128, 192, 3), generator=g).float() - torch.tensor(cfg.mean_pixel))
    images = images.permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0., 0., 128., 192.]] * 5, device=dev)
    net = MaskRCNNInference(sd, cfg, dev)
    assert net.max_batch >= 5
    whole = net.predict(images, windows)
    net.max_batch = 2
    split = net.predict(images, windows)
    torch.cuda.synchronize()
    for f in ("class_ids", "scores", "boxes", "counts", "masks"):
        assert torch.equal(getattr(whole, f), getattr(split, f)), f
    assert tuple(split.masks.shape) == (5, 10, 28, 28, 81) and int(split.counts.sum()) > 0
    with pytest.raises(AssertionError):
        net.predict(images, windows, return_intermediates=True)


def test_small_images_large_batch_is_split_by_the_roi_head_tensors():
    """for small images the RoI heads' tensors reach the kernels' 2^30-element limit before the trunk's do — 64 x 64
    images with 1000 proposals: the pooled crops [B*1000, 7, 7, 256] pass it at B = 86 while the trunk alone would allow 8191.
    max_batch_per_launch takes the minimum over every batch-scaled tensor, so predict() splits such a batch instead of raising
    'tensor too large' from a launch; the split result equals the images run alone."""
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    dev = torch.device("cuda:0")
    cfg = InferenceConfig(image_height=64, image_width=64, backbone="resnet50", pre_nms_limit=1000, proposal_count=1000,
                          detection_max_instances=5)
    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(6)
    sd["rpn.conv_class.weight"] = sd["rpn.conv_class.weight"] * 0.05
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_bbox.weight"] = torch.randn(324, 1024, generator=g) * 0.02
    net = MaskRCNNInference(sd, cfg, dev)
    assert net.max_batch == 85
    n = 87
    images = (torch.randint(0, 256, (n, 64, 64, 3), generator=g).float() - torch.tensor(cfg.mean_pixel))
    images = images.permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0., 0., 64., 64.]] * n, device=dev)
    det = net.predict(images, windows)
    torch.cuda.synchronize()
    assert tuple(det.masks.shape) == (n, 5, 28, 28, 81) and int(det.counts.sum()) > 0
    for i in (0, 43, 44, 86):   # either side of the split point
        one = net.predict(images[i:i + 1], windows[i:i + 1])
        for f in ("class_ids", "scores", "boxes", "counts", "masks"):
            assert torch.equal(getattr(det, f)[i:i + 1], getattr(one, f)), (i, f)
