"""GPU parity tests (-m gpu) at BASELINE.json's FULL sizes, against the CPU oracle / torch-CPU fp32:

  config 3  ResNet-50-FPN, 1024 x 1024, 1000 proposals per image: every stage of the step (trunk, RPN scores/deltas,
            proposals incl. the bit-exact NMS keep set over 1000 boxes, classifier on the ~1000 RoIs, detections,
            masks) vs oracle.* on the same inputs; batch-8 independence (image i alone == slice i of the batch)
  config 5  ResNet-101-FPN, 832 x 1344 (1333 x 800 padded to /64): exact-fp32 vs the oracle, the two fp16-operand MFMA
            modes vs their stated tolerances
  per layer every distinct 3x3 shape of SURVEY App. B at FULL spatial size, unit-scale data, vs torch-CPU: 1e-4 ABS
  levels    +-4 ulp sweeps around the k = 2.5 / 3.5 / 4.5 pyramid-level boundaries vs torch-CPU (model.py:331-338)

Floating-point bars are written where they are asserted. Every comparison also records the measured max |error| and
the activation range max |reference| in gpurun_out/parity_fullsize.json (copied to profiles/ per round), so the distance
from the 1e-4 ABSOLUTE bar of BASELINE.json's north_star is on record, not just pass/fail: the pipeline-level bars are
1e-4 * max(1, max|activation|) — equal to 1e-4 abs wherever activations are at unit scale, which the per-layer tests
and the unit-scale trunk run below make explicit.
"""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


@pytest.fixture(scope="module", autouse=True)
def _write_report():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    name = "parity_fullsize_sync_fuzz.json" if os.environ.get("MRCNN_SYNC_FUZZ_CHILD") == "1" else "parity_fullsize.json"
    with open(os.path.join(out, name), "w") as fh:
        json.dump(REPORT, fh, indent=1, sort_keys=True)


def _record(key, got, want, bar_rel=1e-4):
    """max|err|, max|ref| → REPORT; asserts err <= bar_rel * max(1, max|ref|) with both numbers in the message."""
    err = (got.double() - want.double()).abs().max().item()
    rng = want.abs().max().item()
    tol = bar_rel * max(1.0, rng)
    REPORT[key] = {"max_abs_err": err, "max_abs_ref": rng, "bar": tol, "meets_1e-4_abs": bool(err <= 1e-4)}
    assert err <= tol, f"{key}: max|err| {err:.3e} > {tol:.3e} (max|ref| {rng:.3e})"
    return err, rng


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import maskrcnn_amd  # noqa: F401
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------------------
# per layer, full spatial size, unit-scale data, 1e-4 ABS vs torch-CPU
# ------------------------------------------------------------------------------------------------------------
FULL_3X3 = [
    # (B, H, W, Cin, Cout, relu, affine)  — SURVEY App. B at 1024^2, batch 1 (mask head: 50 RoIs)
    (1, 256, 256, 64, 64, True, True),      # C2 conv2
    (1, 128, 128, 128, 128, True, True),    # C3 conv2
    (1, 64, 64, 256, 256, True, True),      # C4 conv2
    (1, 32, 32, 512, 512, True, True),      # C5 conv2
    (1, 256, 256, 256, 256, False, False),  # FPN P2 smoothing (77 GFLOP)
    (1, 128, 128, 256, 256, False, False),  # P3
    (1, 64, 64, 256, 256, False, False),    # P4
    (1, 32, 32, 256, 256, False, False),    # P5
    (1, 256, 256, 256, 512, True, False),   # RPN conv_shared on P2 (155 GFLOP)
    (1, 128, 128, 256, 512, True, False),   # P3
    (1, 64, 64, 256, 512, True, False),     # P4
    (1, 32, 32, 256, 512, True, False),     # P5
    (1, 16, 16, 256, 512, True, False),     # P6
    (50, 14, 14, 256, 256, True, True),     # mask head conv1-4 on 50 detections
]


@pytest.mark.parametrize("case", FULL_3X3, ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_winograd_full_spatial_size_vs_torch_cpu(dev, case):
    """The Winograd F(2x2,3x3) kernel on every distinct 3x3 layer shape at its FULL spatial size: x ~ N(0,1), He-scaled
    weights (outputs at unit scale), against torch-CPU conv2d in fp32. Bar: 1e-4 absolute."""
    from maskrcnn_amd import ops
    b, h, w, cin, cout, relu, affine = case
    g = torch.Generator().manual_seed(1000 + h + cin + cout + b)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    scale = (torch.rand(cout, generator=g) + 0.5) if affine else None
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, None, padding=1)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1)
    u = ops.winograd_weights(wt.permute(0, 2, 3, 1).contiguous().to(dev))
    y = ops.conv3x3_winograd(x.permute(0, 2, 3, 1).contiguous().to(dev), u,
                             None if scale is None else scale.to(dev), shift.to(dev), relu).cpu()
    err = (y - ref).abs().max().item()
    rng = ref.abs().max().item()
    REPORT["layer3x3/" + "x".join(str(v) for v in case[:5])] = {"max_abs_err": err, "max_abs_ref": rng, "bar": 1e-4,
                                                                "meets_1e-4_abs": bool(err <= 1e-4)}
    assert err <= 1e-4, f"max|err| {err:.3e} > 1e-4 abs (max|ref| {rng:.2f})"


@pytest.mark.parametrize("case", [c for c in FULL_3X3 if c[1] % 4 == 0 and c[2] % 4 == 0 and c[4] % 64 == 0 and c[1] >= 64],
                         ids=lambda c: "x".join(str(v) for v in c[:5]))
def test_winograd4_full_spatial_size_vs_torch_cpu(dev, case):
    """The Winograd F(4x4,3x3) kernel on every layer shape the pipeline routes to it (modules.WINOGRAD4*: FPN smoothing,
    RPN shared conv, Bottleneck conv2, maps of 64 x 64 and more) at FULL spatial size, same data and the same 1e-4
    absolute bar as the F(2x2) test above."""
    from maskrcnn_amd import ops
    b, h, w, cin, cout, relu, affine = case
    g = torch.Generator().manual_seed(1000 + h + cin + cout + b)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    scale = (torch.rand(cout, generator=g) + 0.5) if affine else None
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, None, padding=1)
    if scale is not None:
        ref = ref * scale.view(1, -1, 1, 1)
    ref = ref + shift.view(1, -1, 1, 1)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1)
    u4 = ops.winograd4_weights(wt.permute(0, 2, 3, 1).contiguous().to(dev))
    xk = ops.nhwc_to_kblocked(x.permute(0, 2, 3, 1).contiguous().to(dev))
    y = ops.conv3x3_winograd4(xk, u4, None if scale is None else scale.to(dev), shift.to(dev), relu).cpu()
    err = (y - ref).abs().max().item()
    rng = ref.abs().max().item()
    REPORT["layer3x3_f4/" + "x".join(str(v) for v in case[:5])] = {"max_abs_err": err, "max_abs_ref": rng, "bar": 1e-4,
                                                                   "meets_1e-4_abs": bool(err <= 1e-4)}
    assert err <= 1e-4, f"max|err| {err:.3e} > 1e-4 abs (max|ref| {rng:.2f})"


FULL_1X1 = [
    # (B, H, W, Cin, Cout, stride, relu, residual)  — the direct kernel's bottleneck layers at full spatial size
    (1, 256, 256, 64, 256, 1, True, True),      # C2 conv3 + residual
    (1, 256, 256, 256, 64, 1, True, False),     # C2 conv1
    (1, 256, 256, 256, 128, 2, True, False),    # C3 conv1 stride 2
    (1, 128, 128, 128, 512, 1, True, True),     # C3 conv3 + residual
    (1, 64, 64, 1024, 256, 1, True, False),     # C4 conv1
    (1, 32, 32, 512, 2048, 1, True, True),      # C5 conv3 + residual
    (1, 256, 256, 512, 18, 1, False, False),    # RPN heads on P2
    (8, 32, 32, 2048, 256, 1, False, False),    # P5 lateral at batch 8: the under-filled long-K case (128x32 tiles, round 4)
    (8, 28, 28, 256, 81, 1, False, False),      # 64 < Cout <= 96: the 128x96 tile (mask head conv5's shape class, round 4)
]


@pytest.mark.parametrize("case", FULL_1X1, ids=lambda c: "x".join(str(v) for v in c[:6]))
def test_direct_1x1_full_spatial_size_vs_torch_cpu(dev, case):
    from maskrcnn_amd import ops
    b, h, w, cin, cout, stride, relu, res = case
    g = torch.Generator().manual_seed(2000 + h + cin + cout)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * math.sqrt(2.0 / cin)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, wt, None, stride=stride) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    residual = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + residual
    ref = F.relu(ref) if relu else ref
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    y = ops.conv_bn_act(nhwc(x), nhwc(wt), scale.to(dev), shift.to(dev), stride, (0, 0, 0, 0), relu,
                        nhwc(residual) if res else None).permute(0, 3, 1, 2).cpu()
    err = (y - ref).abs().max().item()
    REPORT["layer1x1/" + "x".join(str(v) for v in case[:6])] = {"max_abs_err": err, "max_abs_ref": ref.abs().max().item(),
                                                                "bar": 1e-4, "meets_1e-4_abs": bool(err <= 1e-4)}
    assert err <= 1e-4, f"max|err| {err:.3e} > 1e-4 abs"


# ------------------------------------------------------------------------------------------------------------
# config 3 at full size, stage by stage
# ------------------------------------------------------------------------------------------------------------
def _calibrated(cfg, sd, images, windows, dev, precision="f32"):
    """Random weights saturate the softmaxes (all scores 1.0 → the order is undefined) and blow the box deltas up:
    rescale the four head layers (weights only) until scores are distinct and boxes sane. Both paths then see the same
    state dict."""
    from maskrcnn_amd.pipeline import MaskRCNNInference
    for _ in range(8):
        net = MaskRCNNInference(sd, cfg, dev, precision=precision)
        det, mid = net.predict(images.to(dev), windows.to(dev), with_masks=True, return_intermediates=True)
        sc = mid["rpn_scores"].double().clamp(1e-7, 1 - 1e-7)
        sat = torch.log(sc / (1 - sc)).std().item()
        dstd, lstd, bstd = mid["rpn_deltas"].std().item(), mid["logits"].std().item(), mid["bbox"].std().item()
        done = True
        for key, cur, target, limit in (("rpn.conv_class.weight", sat, 1.0, 2.0), ("rpn.conv_bbox.weight", dstd, 0.5, 1.0),
                                        ("classifier.linear_class.weight", lstd, 2.0, 3.0),
                                        ("classifier.linear_bbox.weight", bstd, 0.5, 1.0)):
            if cur > limit:
                sd[key] = sd[key] * (target / cur)
                done = False
        if done:
            break
    torch.cuda.synchronize()
    return net, det, mid


@pytest.fixture(scope="module", params=["f32", "f32+f16x3"])
def full(dev, request):
    """Both modes are held to the same bars: "f32" (the default and the headline) and "f32+f16x3" (bench.py's alt mode: the
    long-K GEMM-shaped layers on the error-compensated fp16x3 split)."""
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    cfg = InferenceConfig(image_height=1024, image_width=1024, backbone="resnet50", pre_nms_limit=1000,
                          proposal_count=1000, detection_max_instances=50)
    sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(5)
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_class.bias"] = torch.randn(81, generator=g) * 0.5
    sd["classifier.linear_bbox.weight"] = torch.randn(324, 1024, generator=g) * 0.02
    sd["rpn.conv_bbox.bias"] = torch.randn(12, generator=g) * 0.3
    b = 2
    g0 = torch.Generator().manual_seed(0)   # seed 0, uint8-range pixels minus MEAN_PIXEL
    images = torch.randint(0, 256, (b, 1024, 1024, 3), generator=g0).float() - torch.tensor(cfg.mean_pixel)
    images = images.permute(0, 3, 1, 2).contiguous()
    windows = torch.tensor([[0., 0., 1024., 1024.], [192., 0., 832., 1024.]])   # full frame; config 1's window
    net, det, mid = _calibrated(cfg, sd, images, windows, dev, request.param)
    return dict(cfg=cfg, sd=sd, net=net, images=images, windows=windows, det=det, mid=mid, b=b, precision=request.param,
                tag="config3" if request.param == "f32" else "config3[" + request.param + "]")


def _ocfg(oracle, cfg):
    return oracle.Cfg(cfg.image_height, cfg.image_width, PRE_NMS_LIMIT=cfg.pre_nms_limit,
                      RPN_NMS_MAX_ROIS_NUM=cfg.proposal_count, DETECTION_MAX_INSTANCES=cfg.detection_max_instances)


def test_full_size_trunk_and_rpn(full, oracle):
    """fpn (model.py:133-168) and rpn_detect (:1294-1304) at 1024^2 on two images. Bars: 1e-4 * max(1, max|act|) on the
    trunk and the deltas; 1e-4 abs on the fg scores (probabilities)."""
    s = full
    for b in range(s["b"]):
        fms = oracle.fpn_forward(s["images"][b:b + 1], s["sd"], "resnet50")
        assert [tuple(f.shape) for f in fms] == [(1, 256, 256, 256), (1, 256, 128, 128), (1, 256, 64, 64),
                                                 (1, 256, 32, 32), (1, 256, 16, 16)]   # model.py:165-166
        for lvl, (want, got) in enumerate(zip(fms, s["mid"]["feature_maps"])):
            _record(f"{s['tag']}/img{b}/P{lvl + 2}", got[b].permute(2, 0, 1).cpu(), want[0])
        _, rpn_class, rpn_bbox = oracle.rpn_detect(fms, s["sd"])
        assert rpn_class.shape[1] == s["mid"]["rpn_scores"].shape[1] == 261888       # utils.py:288
        err, _ = _record(f"{s['tag']}/img{b}/rpn_fg_score", s["mid"]["rpn_scores"][b].cpu(), rpn_class[0, :, 1])
        assert err <= 1e-4
        _record(f"{s['tag']}/img{b}/rpn_deltas", s["mid"]["rpn_deltas"][b].cpu(), rpn_bbox[0])


def test_full_size_trunk_unit_scale_input(full, oracle):
    """The same trunk on an image scaled to unit range (pixels / 128): how far from 1e-4 ABSOLUTE the ~60-layer trunk
    lands when activations start at unit scale. Asserted: 1e-4 * max(1, max|act|); recorded: err, range, whether 1e-4
    abs was met."""
    s = full
    x = (s["images"][:1] / 128.0).contiguous()
    got = s["net"].backbone(x.to(s["net"].device))
    torch.cuda.synchronize()
    want = oracle.fpn_forward(x, s["sd"], "resnet50")
    for lvl, (w_, g_) in enumerate(zip(want, got)):
        _record(f"{s['tag']}/unit_scale_input/P{lvl + 2}", g_[0].permute(2, 0, 1).cpu(), w_[0])


def _fp64_truth(oracle, image, sd, arch):
    """The trunk in float64 on the host (same graph as oracle.fpn_forward, every tensor and parameter double): the TRUE result
    to ~1e-13, against which both fp32 implementations are measured."""
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items() if k.startswith("fpn.")}
    with torch.no_grad():
        return oracle.fpn_forward(image.double(), sd64, arch)


FP64_MAX_BAR = 2.5e-6   # max|HIP - fp64| <= 2.5e-6 * max|fp64| per pyramid level = 21 ulps (2^-23) of the level's range
FP64_RMS_BAR = 5.0e-7   # rms(HIP - fp64) <= 5e-7 * max|fp64|


def _fp64_rows(key, truth, got, want32):
    """Per level: max / rms of |HIP - fp64| and |oracle_fp32 - fp64| → REPORT. Asserted — a bar that does not depend on the host
    the oracle runs on (round 4's was a ratio to torch-CPU's own error, i.e. fitted to one box's oneDNN blocking): the HIP trunk's
    distance from the float64 truth is at most FP64_MAX_BAR (2.5e-6) of the level's activation range, its rms at most
    FP64_RMS_BAR (5e-7) of it. Measured on the MI355X (profiles/r04_fp64_truth.jsonl, r05_parity_fullsize.json): 1.5 - 2.1e-6
    (13 - 17 ulps) max and 2.3 - 3.5e-7 rms in exact fp32 over both configs, 1.6 - 1.9e-6 / 2.9 - 4.0e-7 for the f32+f16x3 alt
    mode. The reference's own arithmetic (torch-CPU fp32) is recorded beside it: 6 - 9 ulps of the range (1.6e-4 abs at |act| 210,
    7.2e-4 at 650: ABOVE north_star's 1e-4 absolute on its own), so the HIP trunk is 1.8 - 2.2 x as far from the truth as the
    oracle on this host — the length of the sequential fp32 accumulation over K in the MFMA loops (the exact direct kernel
    everywhere is 25 - 35 ulps; Winograd shortens the sums), attributed per layer by tools/fp64_truth.py --attribute."""
    for lvl, (t, g_, w_) in enumerate(zip(truth, got, want32)):
        eh, eo = (g_.double() - t[0]).abs(), (w_[0].double() - t[0]).abs()
        hip_err, ora_err = eh.max().item(), eo.max().item()
        hip_rms, ora_rms = eh.pow(2).mean().sqrt().item(), eo.pow(2).mean().sqrt().item()
        rng = t.abs().max().item()
        REPORT[f"{key}/P{lvl + 2}"] = {"max_abs_hip_minus_fp64": hip_err, "max_abs_oracle_fp32_minus_fp64": ora_err,
                                       "rms_hip_minus_fp64": hip_rms, "rms_oracle_fp32_minus_fp64": ora_rms,
                                       "max_abs_fp64": rng, "hip_over_oracle_max": hip_err / max(ora_err, 1e-30),
                                       "hip_over_oracle_rms": hip_rms / max(ora_rms, 1e-30),
                                       "hip_max_over_range": hip_err / rng, "hip_rms_over_range": hip_rms / rng,
                                       "hip_max_in_ulps_of_range": hip_err / (rng * 2.0 ** -23),
                                       "bar_max_over_range": FP64_MAX_BAR, "bar_rms_over_range": FP64_RMS_BAR,
                                       "oracle_fp32_meets_1e-4_abs": bool(ora_err <= 1e-4), "hip_meets_1e-4_abs": bool(hip_err <= 1e-4)}
        assert hip_err <= max(FP64_MAX_BAR * rng, 1e-4), (f"{key}/P{lvl + 2}: max|HIP - fp64| {hip_err:.3e} > {FP64_MAX_BAR} x "
                                                         f"max|fp64| {rng:.1f} (oracle fp32: {ora_err:.3e})")
        assert hip_rms <= max(FP64_RMS_BAR * rng, 2e-5), (f"{key}/P{lvl + 2}: rms(HIP - fp64) {hip_rms:.3e} > {FP64_RMS_BAR} x "
                                                         f"max|fp64| {rng:.1f} (oracle fp32: {ora_rms:.3e})")


def test_full_size_trunk_against_fp64_truth(full, oracle):
    """What the relative pipeline bar rests on (north_star says 1e-4 ABSOLUTE; at |act| ~ 200 the HIP trunk is 3.5e-4 from the
    fp32 oracle): the same image through the trunk in FLOAT64 on the host. If torch-CPU fp32 — the reference's own arithmetic —
    is itself further than 1e-4 from the true result on this data (it is: 1.6e-4), a 1e-4 ABSOLUTE bar against it measures
    agreement of rounding sequences, not accuracy; the bar asserted here is the HIP trunk's own distance from the truth in units
    of the activation range (_fp64_rows: host-independent). Both distances are recorded per level."""
    s = full
    img = s["images"][:1]
    truth = _fp64_truth(oracle, img, s["sd"], "resnet50")
    want32 = oracle.fpn_forward(img, s["sd"], "resnet50")
    got = [m[0].permute(2, 0, 1).cpu() for m in s["mid"]["feature_maps"]]
    _fp64_rows(f"{s['tag']}/fp64_truth/img0", truth, got, want32)


def test_full_size_proposals(full, oracle):
    """rpn_refine (model.py:1307-1382) with 1000 proposals: decoded boxes vs the oracle (expf ulp only), the NMS keep
    set over the 1000 boxes the HIP path used BIT-EXACT, rois == keep-gathered boxes / [H,W,H,W] exactly."""
    s = full
    ocfg = _ocfg(oracle, s["cfg"])
    anchors = oracle.anchors_for(ocfg)
    assert torch.equal(anchors, s["net"].anchors.cpu())
    for b in range(s["b"]):
        scores = s["mid"]["rpn_scores"][b].cpu()
        rpn_class = torch.stack([1 - scores, scores], 1).unsqueeze(0)
        rois, dets = oracle.rpn_refine(rpn_class, s["mid"]["rpn_deltas"][b].cpu().unsqueeze(0), anchors, ocfg,
                                       return_dets=True)
        got_dets = s["mid"]["rpn_dets"][b].cpu()
        assert got_dets.shape == (1000, 5)
        assert torch.equal(got_dets[:, 4], dets[:, 4])                      # same top-1000 scores, same order
        # With 261 888 anchors some of the top-1000 fp32 scores TIE. The reference orders ties with ATen's unstable
        # sort (unspecified), this library by lower anchor index: rows are compared within each group of equal scores
        # as sets (sorted by their coordinates), everything else row by row.
        def canon(t):
            a = t.numpy()
            return torch.from_numpy(a[np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0], -a[:, 4]))])
        ties = int(1000 - torch.unique(dets[:, 4]).numel())
        d = (canon(got_dets)[:, :4] - canon(dets)[:, :4]).abs()
        REPORT[f"{s['tag']}/img{b}/proposal_boxes"] = {"max_abs_err_px": d.max().item(), "tied_scores": ties,
                                                    "boxes_not_bit_identical": int((d.max(1).values > 0).sum())}
        assert d.max().item() <= 1e-3                                        # pixels; exp() ulp differences only
        keep = oracle.nms(got_dets, ocfg.RPN_NMS_THRESHOLD)[:ocfg.RPN_NMS_MAX_ROIS_NUM]
        n = int(s["mid"]["roi_counts"][b])
        assert n == keep.numel() and n > 100, n
        want = got_dets[keep, :4] / torch.tensor([1024., 1024., 1024., 1024.])
        assert torch.equal(s["mid"]["rois"][b, :n].cpu(), want)
        assert bool((s["mid"]["rois"][b, n:] == 0).all())
        REPORT[f"{s['tag']}/img{b}/proposals_kept"] = n


def test_full_size_classifier_and_detections(full, oracle):
    """RoIAlign 7x7 + classifier (model.py:759-800) on every valid RoI (~800 of 1000 slots), then mrn_refine
    (:1389-1487) on the HIP path's own head outputs: detections identical."""
    s = full
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
        _record(f"{s['tag']}/img{b}/classifier_logits", got_logits, logits)
        _record(f"{s['tag']}/img{b}/classifier_bbox", got_bbox, bbox)
        gp = torch.softmax(got_logits, dim=1)
        cls, sc, bx = oracle.mrn_refine(rois, gp, got_bbox, tuple(s["windows"][b].tolist()), ocfg)
        k = int(s["det"].counts[b])
        if cls is None:
            assert k == 0
            continue
        assert k == cls.size(1)
        total += k
        assert torch.equal(s["det"].class_ids[b, :k].cpu(), cls[0])
        same = (s["det"].boxes[b, :k].cpu() == bx[0]).all(1)
        REPORT[f"{s['tag']}/img{b}/detections"] = {"count": k, "boxes_identical": int(same.sum()),
                                                "max_box_diff_px": (s["det"].boxes[b, :k].cpu() - bx[0]).abs().max().item()}
        assert torch.equal(s["det"].boxes[b, :k].cpu(), bx[0])
        assert torch.allclose(s["det"].scores[b, :k].cpu(), sc[0], rtol=0, atol=1e-6)
        assert bool((s["det"].class_ids[b, k:] == 0).all())
    assert total > 0, "test configuration produced no detections"


def test_full_size_masks(full, oracle):
    """RoIAlign 14x14 + mask head (model.py:875-920) on the detections: sigmoid outputs within 1e-4 abs."""
    s = full
    ocfg = _ocfg(oracle, s["cfg"])
    for b in range(s["b"]):
        k = int(s["det"].counts[b])
        if k == 0:
            continue
        fms = [f[b:b + 1].permute(0, 3, 1, 2).cpu().contiguous() for f in s["mid"]["feature_maps"][:4]]
        boxes = s["det"].boxes[b, :k].cpu()
        want = oracle.mask_forward(fms, boxes / 1024.0, s["sd"], ocfg)        # [k,81,28,28]
        got = s["det"].masks[b, :k].permute(0, 3, 1, 2).cpu()
        err, _ = _record(f"{s['tag']}/img{b}/masks", got, want)
        assert err <= 1e-4


def test_full_size_batch8_independence(full):
    """configs[2] runs 8 images per step: image i alone reproduces slice i of the batch BIT FOR BIT, at every stage
    (frozen BN, no cross-image state; every kernel's per-output summation order is independent of the batch)."""
    s = full
    net, dev = s["net"], s["net"].device
    g = torch.Generator().manual_seed(8)
    images = torch.randint(0, 256, (8, 1024, 1024, 3), generator=g).float() - torch.tensor(s["cfg"].mean_pixel)
    images = images.permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0., 0., 1024., 1024.]] * 8, device=dev)
    det8, mid8 = net.predict(images, windows, return_intermediates=True)
    for i in (0, 5):
        det1, mid1 = net.predict(images[i:i + 1], windows[i:i + 1], return_intermediates=True)
        for lvl, (a, c) in enumerate(zip(mid1["feature_maps"], mid8["feature_maps"])):
            assert torch.equal(a[0], c[i]), f"image {i}: P{lvl + 2} differs between batch 1 and batch 8"
        assert torch.equal(mid1["rpn_scores"][0], mid8["rpn_scores"][i])
        assert torch.equal(mid1["rpn_dets"][0], mid8["rpn_dets"][i])
        assert torch.equal(mid1["rois"][0], mid8["rois"][i]) and int(mid1["roi_counts"][0]) == int(mid8["roi_counts"][i])
        p = mid8["rois"].size(1)
        assert torch.equal(mid1["logits"], mid8["logits"][i * p:(i + 1) * p])
        assert torch.equal(det1.class_ids[0], det8.class_ids[i]) and torch.equal(det1.boxes[0], det8.boxes[i])
        assert torch.equal(det1.scores[0], det8.scores[i]) and int(det1.counts[0]) == int(det8.counts[i])
        assert torch.equal(det1.masks[0], det8.masks[i])
    REPORT[s["tag"] + "/batch8_independence"] = "bit-identical (images 0 and 5; trunk, RPN, proposals, classifier, detections, masks)"


def test_full_size_kernel_routing_is_the_documented_one_and_independent_of_the_batch(full):
    """modules.py's routing — which kernel every conv of the step takes — as a table: at 1024^2 the R50-FPN step is 75 conv
    launches (direct implicit GEMM 45 incl. the streaming 1x1 kernel, F(4x4) Winograd 19, F(2x2) linear tiles 4 =
    the mask head, F(2x2) spatial tiles 6, stem 1), and the SAME sequence of (kernel, N, K) for one image as for eight: a layer's
    kernel is chosen by the image size, never by the batch (which is what makes image i of a batch equal image i alone)."""
    from maskrcnn_amd import ops
    s = full
    if s["precision"] != "f32":
        pytest.skip("the routing table is the f32 mode's")
    net, dev = s["net"], s["net"].device
    g = torch.Generator().manual_seed(3)
    images = (torch.randint(0, 256, (8, 1024, 1024, 3), generator=g).float() - torch.tensor(s["cfg"].mean_pixel))
    images = images.permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0., 0., 1024., 1024.]] * 8, device=dev)
    seqs = {}
    for b in (1, 8):
        ops.CONV_PROFILE = []
        try:
            net.predict(images[:b], windows[:b])
            torch.cuda.synchronize()
            prof = ops.CONV_PROFILE
        finally:
            ops.CONV_PROFILE = None
        seqs[b] = [(r[5] if len(r) > 5 else "direct", r[3][1], r[3][2]) for r in prof]
    assert seqs[1] == seqs[8], [(i, a, c) for i, (a, c) in enumerate(zip(seqs[1], seqs[8])) if a != c][:5]
    counts = {}
    for tag, _, _ in seqs[8]:
        counts[tag] = counts.get(tag, 0) + 1
    REPORT["config3/kernel_routing"] = counts
    assert counts == {"stem": 1, "direct": 45, "winograd4": 19, "winograd": 4, "winograd_spatial": 6}, counts


# ------------------------------------------------------------------------------------------------------------
# config 5 at full size: ResNet-101-FPN, 832 x 1344
# ------------------------------------------------------------------------------------------------------------
def test_config5_full_size_r101_832x1344(dev, oracle):
    """BASELINE configs[4] geometry (1333 x 800 padded to multiples of 64 → 832 x 1344), ResNet-101-FPN, one image:
    exact-fp32 and f16x3 vs the oracle at 1e-4 * max(1, max|act|); the plain-fp16 MFMA path ("fp16 MFMA path") at its
    stated tolerance, 2e-2 of the activation range, against the oracle AND against the exact-fp32 HIP path."""
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    h, w = 832, 1344
    cfg = InferenceConfig(image_height=h, image_width=w, backbone="resnet101", pre_nms_limit=1000, proposal_count=1000)
    sd = modules.synthetic_state_dict("resnet101", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(55)
    images = (torch.randint(0, 256, (1, h, w, 3), generator=g).float() - torch.tensor(cfg.mean_pixel))
    images = images.permute(0, 3, 1, 2).contiguous()
    windows = torch.tensor([[16., 5., 816., 1338.]], device=dev)          # 800 x 1333 inside the padded canvas
    want = oracle.fpn_forward(images, sd, "resnet101")
    assert [tuple(f.shape[2:]) for f in want] == [(208, 336), (104, 168), (52, 84), (26, 42), (13, 21)]
    f32_maps = None
    for precision, rel in (("f32", 1e-4), ("f16x3", 1e-4), ("f32+f16x3", 1e-4), ("f16", 2e-2)):
        net = MaskRCNNInference(sd, cfg, dev, precision=precision)
        det, mid = net.predict(images.to(dev), windows, return_intermediates=True)
        torch.cuda.synchronize()
        maps = [m[0].permute(2, 0, 1).cpu() for m in mid["feature_maps"]]
        for lvl, (w_, g_) in enumerate(zip(want, maps)):
            _record(f"config5/{precision}/P{lvl + 2}_vs_oracle", g_, w_[0], rel)
            if f32_maps is not None:
                _record(f"config5/{precision}/P{lvl + 2}_vs_f32_hip", g_, f32_maps[lvl], rel)
        if precision == "f32":
            f32_maps = maps
        a = 3 * sum(hh * ww for hh, ww in ((208, 336), (104, 168), (52, 84), (26, 42), (13, 21)))
        assert tuple(mid["rpn_scores"].shape) == (1, a)
        assert tuple(det.boxes.shape) == (1, 50, 4) and tuple(det.masks.shape) == (1, 50, 28, 28, 81)
        assert bool((det.boxes[..., 2] <= 832).all()) and bool((det.boxes[..., 3] <= 1344).all())
        del net


def test_config5_f16_kernel_routing_is_independent_of_the_batch(dev):
    """The routing table of the "f16" mode at configs[4]'s geometry (R101-FPN, 832 x 1344; VERDICT r4 item 8: the f32 table was
    the only one pinned): the SAME sequence of (kernel, N, K) for one image as for two — a layer's kernel is a function of the image
    size, never of the batch — and the documented launch counts: the pipelined kernel for every 1x1 / 3x3 layer with 64-multiple
    channels (and, round 5, the RPN's P6 level), the 128 x 128 / 256 x 64 tile kernel for the 64-channel C2 layers and the
    18-channel heads of the small levels, one stem launch."""
    from maskrcnn_amd import modules, ops
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    h, w = 832, 1344
    cfg = InferenceConfig(image_height=h, image_width=w, backbone="resnet101", pre_nms_limit=1000, proposal_count=1000)
    sd = modules.synthetic_state_dict("resnet101", seed=0, bn_seed=1)
    net = MaskRCNNInference(sd, cfg, dev, precision="f16")
    g = torch.Generator().manual_seed(3)
    images = (torch.randint(0, 256, (2, h, w, 3), generator=g).float() - torch.tensor(cfg.mean_pixel)).permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0., 0., float(h), float(w)]] * 2, device=dev)
    seqs = {}
    for b in (1, 2):
        ops.CONV_PROFILE = []
        try:
            net.predict(images[:b], windows[:b])
            torch.cuda.synchronize()
            prof = ops.CONV_PROFILE
        finally:
            ops.CONV_PROFILE = None
        seqs[b] = [(r[5] if len(r) > 5 else "direct", r[3][1], r[3][2]) for r in prof]
    assert seqs[1] == seqs[2], [(i, a, c) for i, (a, c) in enumerate(zip(seqs[1], seqs[2])) if a != c][:5]
    counts = {}
    for tag, _, _ in seqs[2]:
        counts[tag] = counts.get(tag, 0) + 1
    REPORT["config5/f16/kernel_routing"] = counts
    assert counts == F16_ROUTING_832x1344, counts


# conv launches of one "f16" step at 832 x 1344, R101-FPN (measured on the MI355X, round 5): the three C2 blocks are one
# launch each (csrc/bottleneck_f16.hip) — before that 4 + 6 of the per-layer launches (f16p 116, f16 11) — and the mask head's deconv +
# conv5 are one (csrc/mask_tail_f16.hip)
F16_ROUTING_832x1344 = {"stem": 1, "f16blk": 3, "f16tail": 1, "f16p": 112, "f16": 3}


def _iou_matrix(a, b):
    """[n,4] x [m,4] pixel boxes (y1,x1,y2,x2) → IoU [n,m] (plain areas, no +1)."""
    a, b = a.double(), b.double()
    tl = torch.maximum(a[:, None, :2], b[None, :, :2])
    br = torch.minimum(a[:, None, 2:], b[None, :, 2:])
    inter = (br - tl).clamp(min=0).prod(-1)
    area = lambda t: ((t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])).clamp(min=0)
    return inter / (area(a)[:, None] + area(b)[None, :] - inter).clamp(min=1e-12)


def test_config5_end_to_end_fp32_exact_and_fp16_detections(dev, oracle):
    """BASELINE configs[4] END TO END (model.py:1389-1487 is where a trunk error becomes a different answer), R101-FPN,
    832 x 1344, batch 2, calibrated heads (distinct scores, sane boxes):
      * fp32 mode: trunk vs the float64 truth (as configs[2]); the NMS keep set over the proposals it used bit-exact;
        detections IDENTICAL to oracle.mrn_refine on the HIP path's own head outputs; masks 1e-4 abs vs oracle.mask_forward;
      * fp16 mode (the config's "fp16 MFMA path"): the fp32 detections with score > 0.5 have an fp16 detection of the SAME class
        with IoU >= 0.9 — all but at most one in eight (recorded: how many, the worst IoU, the score differences); the fp16 mask
        head on the fp32 path's boxes within 3e-2 abs of the fp32 masks."""
    from maskrcnn_amd import modules, ops
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    h, w, nb = 832, 1344, 2
    cfg = InferenceConfig(image_height=h, image_width=w, backbone="resnet101", pre_nms_limit=1000, proposal_count=1000,
                          detection_max_instances=50)
    sd = modules.synthetic_state_dict("resnet101", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(5)
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_class.bias"] = torch.randn(81, generator=g) * 0.5
    sd["classifier.linear_bbox.weight"] = torch.randn(324, 1024, generator=g) * 0.02
    sd["rpn.conv_bbox.bias"] = torch.randn(12, generator=g) * 0.3
    g0 = torch.Generator().manual_seed(55)
    images = (torch.randint(0, 256, (nb, h, w, 3), generator=g0).float() - torch.tensor(cfg.mean_pixel))
    images = images.permute(0, 3, 1, 2).contiguous()
    windows = torch.tensor([[16., 5., 816., 1338.], [0., 0., 832., 1344.]])
    net, det, mid = _calibrated(cfg, sd, images, windows, dev, "f32")
    ocfg = _ocfg(oracle, cfg)
    scale4 = torch.tensor([float(h), float(w), float(h), float(w)])
    # ---- fp32 mode
    truth = _fp64_truth(oracle, images[:1], sd, "resnet101")
    want32 = oracle.fpn_forward(images[:1], sd, "resnet101")
    _fp64_rows("config5/f32/fp64_truth/img0", truth, [m[0].permute(2, 0, 1).cpu() for m in mid["feature_maps"]], want32)
    del truth, want32
    p = mid["rois"].size(1)
    total = 0
    for b in range(nb):
        got_dets = mid["rpn_dets"][b].cpu()
        keep = oracle.nms(got_dets, ocfg.RPN_NMS_THRESHOLD)[:ocfg.RPN_NMS_MAX_ROIS_NUM]
        n = int(mid["roi_counts"][b])
        assert n == keep.numel() and n > 100, n                                   # NMS keep set: bit-exact
        assert torch.equal(mid["rois"][b, :n].cpu(), got_dets[keep, :4] / scale4)
        rois = mid["rois"][b, :n].cpu()
        got_logits, got_bbox = mid["logits"][b * p:b * p + n].cpu(), mid["bbox"][b * p:b * p + n].cpu()
        fms = [f[b:b + 1].permute(0, 3, 1, 2).cpu().contiguous() for f in mid["feature_maps"][:4]]
        logits, _, bbox = oracle.classifier_forward(fms, rois, sd, ocfg)
        _record(f"config5/f32/img{b}/classifier_logits", got_logits, logits)
        _record(f"config5/f32/img{b}/classifier_bbox", got_bbox, bbox)
        cls, sc, bx = oracle.mrn_refine(rois, torch.softmax(got_logits, dim=1), got_bbox, tuple(windows[b].tolist()), ocfg)
        k = int(det.counts[b])
        assert cls is not None and k == cls.size(1) and k > 0
        total += k
        assert torch.equal(det.class_ids[b, :k].cpu(), cls[0])
        assert torch.equal(det.boxes[b, :k].cpu(), bx[0])                          # detections identical
        assert torch.allclose(det.scores[b, :k].cpu(), sc[0], rtol=0, atol=1e-6)
        REPORT[f"config5/f32/img{b}/detections"] = {"count": k, "boxes_identical": k, "proposals_kept": n}
        # the reference divides all four coordinates by h (model.py:1188); this library divides (y, x) by (h, w) — the oracle
        # is given the boxes the library pools
        want_m = oracle.mask_forward(fms, det.boxes[b, :k].cpu() / scale4, sd, ocfg)
        err, _ = _record(f"config5/f32/img{b}/masks", det.masks[b, :k].permute(0, 3, 1, 2).cpu(), want_m)
        assert err <= 1e-4
    # ---- fp16 mode on the same weights and images
    net16 = MaskRCNNInference(sd, cfg, dev, precision="f16")
    det16, mid16 = net16.predict(images.to(dev), windows.to(dev), return_intermediates=True)
    torch.cuda.synchronize()
    matched = strong = 0
    worst_iou, worst_ds = 1.0, 0.0
    for b in range(nb):
        k, k16 = int(det.counts[b]), int(det16.counts[b])
        sel = (det.scores[b, :k] > 0.5).nonzero().flatten().cpu()
        if sel.numel() == 0:
            continue
        iou = _iou_matrix(det.boxes[b, :k].cpu()[sel], det16.boxes[b, :k16].cpu())
        same = det.class_ids[b, :k].cpu()[sel][:, None] == det16.class_ids[b, :k16].cpu()[None, :]
        best, arg = (iou * same).max(1) if k16 else (torch.zeros(sel.numel(), dtype=torch.float64), None)
        strong += sel.numel()
        matched += int((best >= 0.9).sum())
        worst_iou = min(worst_iou, best.min().item())
        if k16:
            ds = (det.scores[b, :k].cpu()[sel] - det16.scores[b, :k16].cpu()[arg]).abs()
            worst_ds = max(worst_ds, ds[best >= 0.9].max().item() if bool((best >= 0.9).any()) else 0.0)
    REPORT["config5/f16/detections_vs_f32"] = {"f32_detections_score_gt_0.5": strong, "matched_same_class_iou_ge_0.9": matched,
                                               "worst_best_iou": worst_iou, "max_abs_score_diff_of_matched": worst_ds,
                                               "f16_counts": det16.counts.tolist(), "f32_counts": det.counts.tolist()}
    assert strong > 0, "calibration left no confident detections"
    # the fp16 mask head on the boxes of the fp32 path (same boxes → the difference is the arithmetic's, not the box set's)
    d = det.boxes.size(1)
    mrois = (det.boxes / scale4.to(dev)).view(-1, 4).contiguous()
    mp = ops.roi_align_pyramid(mid16["feature_maps"][:4], mrois, cfg.mask_pool_size, float(h * w), rois_per_image=d,
                               out_kblocked=net16.mask.wants_kblocked(cfg.mask_pool_size), out_f16=net16.mask.wants_f16())
    m16 = net16.mask(mp)
    m16 = m16.view(nb, d, m16.size(1), m16.size(2), m16.size(3)).float()
    torch.cuda.synchronize()
    merr = 0.0
    for b in range(nb):
        k = int(det.counts[b])
        merr = max(merr, (m16[b, :k] - det.masks[b, :k]).abs().max().item())
    REPORT["config5/f16/masks_on_f32_boxes_max_abs_diff"] = merr
    # measured on the MI355X (profiles/r04_parity_fullsize.json): 2.3e-2 — the fp16 trunk's 2e-3-of-range feature error through
    # an fp16 mask head, on sigmoid outputs in [0, 1]
    assert merr <= FP16_MASK_BAR, merr
    # (two images are an anecdote — 10 of 11 in round 4; the RATE over three seeds x eight images is asserted by
    # test_config5_fp16_detection_agreement_rate below)
    assert matched >= strong - max(1, strong // 8), (f"fp16 path: {matched} of {strong} confident fp32 detections have a same-class "
                                                      f"fp16 detection with IoU >= 0.9 (worst {worst_iou:.3f})")


FP16_MASK_BAR = 3e-2   # fp16 mask head on the fp32 path's boxes vs the fp32 masks, sigmoid outputs in [0, 1] (measured 2.3e-2)


def test_config5_fp16_detection_agreement_rate(dev):
    """BASELINE configs[4]'s "fp16 MFMA path" END TO END as a RATE (round 4 rested on 11 detections of two images, one seed):
    R101-FPN, 832 x 1344, batch 8, three image seeds, calibrated heads. Every fp32 detection with score > 0.5 (the fp32 path's
    detections equal the oracle's: test above and configs[2]) is looked up among the fp16 path's detections of the same image
    and class. Asserted over all of them (>= 100): at least 95 % have a same-class fp16 detection with IoU >= 0.9 (measured:
    127 of 130, score differences of the matched <= 2e-3) and at most 3 % are without one at IoU >= 0.5. Measured: 2 of 130 —
    both BARELY confident in fp32 (scores 0.507 and 0.501 against the 0.5 cut) and both a different PROPOSAL, not a wrong box: the
    fp16 path's nearest same-class candidate before its per-class NMS overlaps them by IoU 0.55 / 0.61 with a score of 0.39 / 0.41,
    i.e. the RoI they came from did not survive the fp16 path's proposal stage (a near-tie in the RPN's top-k / NMS,
    model.py:1345-1366), and with RANDOM head weights a neighbouring RoI scores differently. Every unmatched case is recorded
    with these figures.
    Recorded: the IoU histogram of the best matches, the score-difference histogram of the matched pairs, the unmatched cases."""
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference
    h, w, nb = 832, 1344, 8
    cfg = InferenceConfig(image_height=h, image_width=w, backbone="resnet101", pre_nms_limit=1000, proposal_count=1000,
                          detection_max_instances=50)
    sd = modules.synthetic_state_dict("resnet101", seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(5)
    sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
    sd["classifier.linear_class.bias"] = torch.randn(81, generator=g) * 0.5
    sd["classifier.linear_bbox.weight"] = torch.randn(324, 1024, generator=g) * 0.02
    sd["rpn.conv_bbox.bias"] = torch.randn(12, generator=g) * 0.3
    windows = torch.tensor([[16., 5., 816., 1338.]] * (nb // 2) + [[0., 0., 832., 1344.]] * (nb - nb // 2))
    best_all, ds_all, per_seed, unmatched = [], [], {}, []
    fixed_best, fixed_ds, fixed_unmatched, roi_overlap, unmatched_roi_iou = [], [], [], [], []
    net = net16 = None
    for seed in (55, 56, 57):
        g0 = torch.Generator().manual_seed(seed)
        images = (torch.randint(0, 256, (nb, h, w, 3), generator=g0).float() - torch.tensor(cfg.mean_pixel))
        images = images.permute(0, 3, 1, 2).contiguous()
        if net is None:   # the heads are calibrated ONCE (first seed); both modes and all seeds then share the weights
            net, det, mid32 = _calibrated(cfg, sd, images, windows, dev, "f32")
            net16 = MaskRCNNInference(sd, cfg, dev, precision="f16")
        else:
            det, mid32 = net.predict(images.to(dev), windows.to(dev), return_intermediates=True)
        det16, mid16 = net16.predict(images.to(dev), windows.to(dev), return_intermediates=True)
        # (round 6) the same fp16 path on the fp32 path's PROPOSALS: its trunk, RoIAlign, heads, decode and NMS, with the one
        # thing the end-to-end comparison cannot hold fixed held fixed
        det16h = net16.predict(images.to(dev), windows.to(dev), rois_override=(mid32["rois"].contiguous(), mid32["roi_counts"].contiguous()))
        torch.cuda.synchronize()
        scale4 = torch.tensor([h, w, h, w], dtype=torch.float64)
        for b in range(nb):
            # proposal sets after the RPN's NMS: fraction of the fp32 path's RoIs with an fp16-path RoI at IoU >= 0.95
            n32, n16 = int(mid32["roi_counts"][b]), int(mid16["roi_counts"][b])
            r32, r16 = mid32["rois"][b, :n32].cpu().double() * scale4, mid16["rois"][b, :n16].cpu().double() * scale4
            roi_overlap.append(float((_iou_matrix(r32, r16).max(1).values >= 0.95).double().mean()) if n32 and n16 else 0.0)
            # heads on fixed proposals
            k, kh = int(det.counts[b]), int(det16h.counts[b])
            sel = (det.scores[b, :k] > 0.5).nonzero().flatten().cpu()
            if sel.numel() == 0:
                continue
            if kh == 0:
                fixed_best += [0.0] * sel.numel()
                continue
            iou = _iou_matrix(det.boxes[b, :k].cpu()[sel], det16h.boxes[b, :kh].cpu())
            same = det.class_ids[b, :k].cpu()[sel][:, None] == det16h.class_ids[b, :kh].cpu()[None, :]
            fb, fa = (iou * same).max(1)
            fixed_best += fb.tolist()
            fixed_ds += (det.scores[b, :k].cpu()[sel] - det16h.scores[b, :kh].cpu()[fa]).abs()[fb >= 0.9].tolist()
            for j in (fb < 0.9).nonzero().flatten().tolist():
                fixed_unmatched.append({"seed": seed, "image": b, "class": int(det.class_ids[b, sel[j]]), "f32_score": float(det.scores[b, sel[j]]),
                                        "best_same_class_iou": float(fb[j]), "any_class_iou": float(iou[j].max())})
        # the fp16 path's candidates BEFORE its per-class NMS: decoded boxes, argmax class and score of every proposal
        from maskrcnn_amd import ops as _ops
        cand, _, cand_cls = _ops.detection_decode(mid16["logits"], mid16["bbox"], mid16["rois"].contiguous(), mid16["roi_counts"],
                                                  windows.to(dev).float().contiguous(), cfg.rpn_bbox_std_dev, h, w, 0.0)
        cand, cand_cls = cand.cpu(), cand_cls.cpu()
        n_seed = 0
        for b in range(nb):
            k, k16 = int(det.counts[b]), int(det16.counts[b])
            sel = (det.scores[b, :k] > 0.5).nonzero().flatten().cpu()
            if sel.numel() == 0:
                continue
            n_seed += sel.numel()
            if k16 == 0:
                best_all += [0.0] * sel.numel()
                continue
            iou = _iou_matrix(det.boxes[b, :k].cpu()[sel], det16.boxes[b, :k16].cpu())
            same = det.class_ids[b, :k].cpu()[sel][:, None] == det16.class_ids[b, :k16].cpu()[None, :]
            best, arg = (iou * same).max(1)
            best_all += best.tolist()
            for j in (best < 0.5).nonzero().flatten().tolist():   # a vanished detection: is its candidate still there?
                dj = sel[j]
                box, cls_, sc_ = det.boxes[b, dj].cpu(), int(det.class_ids[b, dj]), float(det.scores[b, dj])
                nv = int(mid16["roi_counts"][b])
                ci = _iou_matrix(box[None], cand[b, :nv, :4])[0] * (cand_cls[b, :nv] == cls_)
                cb, ca = ci.max(0)
                # (round 6) the explanation, measured: does the fp32 path's RoI behind this detection exist in the fp16 path's
                # proposal set at all? (the detection's box is a refinement of ONE of the fp32 path's RoIs: the one whose decoded
                # candidate it is — found through the fp32 candidates)
                cand32, _, cls32 = _ops.detection_decode(mid32["logits"], mid32["bbox"], mid32["rois"].contiguous(), mid32["roi_counts"],
                                                         windows.to(dev).float().contiguous(), cfg.rpn_bbox_std_dev, h, w, 0.0)
                nv32 = int(mid32["roi_counts"][b])
                src = int((_iou_matrix(box[None], cand32[b, :nv32, :4].cpu())[0] * (cls32[b, :nv32].cpu() == cls_)).argmax())
                n16 = int(mid16["roi_counts"][b])
                scale4 = torch.tensor([h, w, h, w], dtype=torch.float64)
                src_iou = float(_iou_matrix(mid32["rois"][b, src].cpu().double()[None] * scale4, mid16["rois"][b, :n16].cpu().double() * scale4).max())
                unmatched_roi_iou.append(src_iou)
                unmatched.append({"seed": seed, "image": b, "class": cls_, "f32_score": sc_, "best_same_class_f16_detection_iou": float(best[j]),
                                  "source_roi_best_iou_in_f16_proposal_set": src_iou,
                                  "f16_candidate_iou": float(cb), "f16_candidate_score": float(cand[b, ca, 4]),
                                  "any_class_f16_detection_iou": float(_iou_matrix(box[None], det16.boxes[b, :k16].cpu()).max())})
            ds = (det.scores[b, :k].cpu()[sel] - det16.scores[b, :k16].cpu()[arg]).abs()
            ds_all += ds[best >= 0.9].tolist()
        per_seed[str(seed)] = n_seed
    best_t = torch.tensor(best_all, dtype=torch.float64)
    ds_t = torch.tensor(ds_all, dtype=torch.float64)
    n, n90, n50 = best_t.numel(), int((best_t >= 0.9).sum()), int((best_t >= 0.5).sum())
    iou_edges = [0.0, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95, 0.99, 1.0 + 1e-9]
    ds_edges = [0.0, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 1.0]
    hist = lambda t, edges: [int(((t >= lo) & (t < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
    REPORT["config5/f16/detection_agreement_rate"] = {
        "f32_detections_score_gt_0.5": n, "per_seed": per_seed, "matched_same_class_iou_ge_0.9": n90, "rate_iou_ge_0.9": n90 / max(n, 1),
        "with_a_match_iou_ge_0.5": n50, "worst_best_iou": float(best_t.min()) if n else None,
        "best_iou_histogram": {"edges": iou_edges[:-1] + [1.0], "counts": hist(best_t, iou_edges)},
        "abs_score_diff_of_matched_histogram": {"edges": ds_edges, "counts": hist(ds_t, ds_edges)},
        "max_abs_score_diff_of_matched": float(ds_t.max()) if ds_t.numel() else None,
        "without_a_match_iou_ge_0.5": unmatched}
    fb_t, fds_t, ov_t = torch.tensor(fixed_best, dtype=torch.float64), torch.tensor(fixed_ds, dtype=torch.float64), torch.tensor(roi_overlap, dtype=torch.float64)
    ov_edges = [0.0, 0.5, 0.8, 0.9, 0.95, 0.98, 0.99, 1.0 + 1e-9]
    REPORT["config5/f16/heads_on_fp32_proposals"] = {
        "what": "the fp16 path run on the fp32 path's RoIs (predict rois_override): every fp32 detection with score > 0.5 looked up among the same-class fp16 detections of its image",
        "f32_detections_score_gt_0.5": int(fb_t.numel()), "matched_same_class_iou_ge_0.9": int((fb_t >= 0.9).sum()),
        "worst_best_iou": float(fb_t.min()) if fb_t.numel() else None, "best_iou_histogram": {"edges": iou_edges[:-1] + [1.0], "counts": hist(fb_t, iou_edges)},
        "abs_score_diff_of_matched_histogram": {"edges": ds_edges, "counts": hist(fds_t, ds_edges)},
        "max_abs_score_diff_of_matched": float(fds_t.max()) if fds_t.numel() else None, "below_0.9": fixed_unmatched}
    REPORT["config5/f16/proposal_set_overlap"] = {
        "what": "per image (3 seeds x 8): fraction of the fp32 path's RoIs (after the RPN's NMS, <= 1000) that have an fp16-path RoI with IoU >= 0.95",
        "per_image": [round(v, 4) for v in roi_overlap], "min": float(ov_t.min()), "mean": float(ov_t.mean()),
        "histogram": {"edges": ov_edges[:-1] + [1.0], "counts": hist(ov_t, ov_edges)},
        "source_roi_iou_of_the_end_to_end_unmatched": unmatched_roi_iou}
    # (round 6) with the proposal set held fixed NOTHING may be missing: the heads' fp16 arithmetic moves a score by <= 1e-2 and a
    # box by a fraction of a pixel — it does not lose or displace a confident detection
    assert fb_t.numel() >= 100 and int((fb_t >= 0.9).sum()) == fb_t.numel(), (
        f"fp16 heads on the fp32 path's proposals: {int((fb_t < 0.9).sum())} of {fb_t.numel()} confident fp32 detections without a "
        f"same-class fp16 detection at IoU >= 0.9: {fixed_unmatched[:4]}")
    # the proposal sets themselves: what the end-to-end allowance below is made of
    assert float(ov_t.min()) >= 0.90 and float(ov_t.mean()) >= 0.95, f"fp16 vs fp32 proposal sets: per-image overlap min {float(ov_t.min()):.3f}, mean {float(ov_t.mean()):.3f}"
    assert n >= 100, f"only {n} confident fp32 detections over three seeds x eight images"
    assert n90 >= 0.95 * n, f"fp16 path: {n90} of {n} confident fp32 detections matched at IoU >= 0.9 ({n90 / n:.3f} < 0.95)"
    assert n - n50 <= 0.03 * n, f"fp16 path: {n - n50} of {n} confident fp32 detections have NO same-class fp16 detection with IoU >= 0.5"
    for u in unmatched:
        # (round 6) an end-to-end miss is a PROPOSAL-SET difference, measured: the fp32 path's RoI behind the detection has no
        # counterpart (IoU >= 0.95) among the fp16 path's RoIs — with it present, the fixed-proposal assertion above says the
        # heads would have found the detection. (Round 5 bounded these cases by their fp32 score instead.)
        assert u["source_roi_best_iou_in_f16_proposal_set"] < 0.95, u


# ------------------------------------------------------------------------------------------------------------
# pyramid level assignment at the boundaries
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1024, 1024), (832, 1344)], ids=["1024x1024", "832x1344"])
def test_level_boundaries_ulp_sweep(dev, oracle, shape):
    """k = 4 + log2(sqrt(h*w) / (224 / sqrt(H*W))), round-half-even, clamp [2,5] (model.py:331-338): boxes whose
    sqrt(area) sits within +-4 ulp (and +-16 / +-64 ulp) of every level boundary k = 2.5 / 3.5 / 4.5, square and at
    aspect ratios 1:2 ... 1:8, at two offsets, plus 400 ordinary boxes. Pinned by the REFERENCE: tests/golden/
    roi_levels.npz holds the boxes and the level the reference's own model.roi_align gave each of them in the build
    container (make_golden.py roi_levels: constant-valued maps, the crop's value is the level). The in-kernel level
    must equal that for EVERY box."""
    from maskrcnn_amd import ops
    from conftest import load_golden
    hh, ww = shape
    area = float(hh * ww)
    z = load_golden("roi_levels")
    rois = torch.from_numpy(z[f"rois_{hh}x{ww}"])
    ref = torch.from_numpy(z[f"levels_{hh}x{ww}"]).to(torch.int32)
    assert rois.size(0) > 3000
    fms = [torch.zeros(1, hh // s, ww // s, 8, device=dev) for s in (4, 8, 16, 32)]
    _, levels = ops.roi_align_pyramid(fms, rois.to(dev), 7, area, rois_per_image=rois.size(0), return_levels=True)
    bad = (levels.cpu() != ref).nonzero().flatten()
    # for the record: torch-CPU ON THIS HOST (the oracle's roi_levels = the reference's tensor expression through this
    # host's MKL VML log2, which differs between CPUs on a few boundary boxes) and the formula with every fp32 operation
    # correctly rounded, which is what the kernel evaluates — the reference fixture agrees with the latter on every box
    want = oracle.roi_levels(rois, (hh, ww, 3))
    exact = torch.from_numpy(_levels_correctly_rounded(rois.numpy(), area))
    REPORT[f"levels/{hh}x{ww}"] = {"boxes": int(rois.size(0)), "mismatches_vs_reference_fixture": int(bad.numel()),
                                  "reference_vs_correctly_rounded_formula": int((ref != exact).sum()),
                                  "torch_cpu_on_this_host_vs_reference_fixture": int((want != ref).sum()),
                                  "level_histogram": torch.bincount(ref.long(), minlength=6).tolist()}
    assert bad.numel() == 0, f"{bad.numel()} of {rois.size(0)} boxes differ from the reference, first: {rois[bad[:3]].tolist()}"
    hist = torch.bincount(ref.long(), minlength=6)
    assert all(int(hist[l]) > 0 for l in (2, 3, 4, 5)), hist.tolist()


def _levels_correctly_rounded(r, area):
    """model.py:331-338 with every fp32 operation correctly rounded (float64 evaluation, rounded to float32 after each
    operation): machine-independent."""
    h32, w32 = (r[:, 2] - r[:, 0]).astype(np.float32), (r[:, 3] - r[:, 1]).astype(np.float32)
    hw32 = (h32 * w32).astype(np.float32)
    denom = (np.float64(224.0) / np.sqrt(np.float64(np.float32(area))).astype(np.float32)).astype(np.float32)
    with np.errstate(divide="ignore"):
        ratio = (np.sqrt(hw32.astype(np.float64)).astype(np.float32).astype(np.float64) / np.float64(denom)).astype(np.float32)
        k = (np.float32(4.0) + np.log2(ratio.astype(np.float64)).astype(np.float32)).astype(np.float32)
    k = np.where(np.isfinite(k), k, np.float32(-100.0))
    return np.clip(np.rint(k), 2, 5).astype(np.int32)
