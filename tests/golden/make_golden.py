#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build container only).

Sources of truth used here (nothing of theirs is copied into the repo — only input/output vectors):
  * the reference's compiled CPU c++ext, built in place by oracle/build_ref.py → oracle/_ref
    (nms, crop_forward, crop_backward of c++ext/maskrcnn/csrc/vision.cpp:11-15);
  * the reference's Python modules (config.py, utils.py, data.py, model.py) imported from
    /root/reference with third-party modules that are absent from this image (skimage, torchvision)
    replaced by empty placeholders, and `maskrcnn` bound to oracle/_ref through a shim that does what
    c++ext/maskrcnn/__init__.py:21-45 does (that file's legacy autograd.Function cannot run on
    torch >= 1.5).

Usage:  python tests/golden/make_golden.py [nms crop roi_align roi_levels anchors graph refine schema image config1]
        (no argument: rewrite every fixture; deterministic)
"""
import contextlib
import ctypes
import hashlib
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MASKRCNN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import build_ref  # noqa: E402


@contextlib.contextmanager
def mute_stdout():
    """crop_cpu.cpp:163 printf()s on every call."""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        ctypes.CDLL(None).fflush(None)  # drain libc's buffer into /dev/null before restoring fd 1
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)


def load_reference():
    build_ref.build()
    refc = build_ref.load()
    assert refc is not None, "oracle/_ref missing"

    # placeholders for third-party packages this image lacks (never on the hot path)
    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    sk = placeholder("skimage")
    sk.io = placeholder("skimage.io")
    sk.color = placeholder("skimage.color")
    sk.measure = placeholder("skimage.measure", find_contours=None)
    tv = placeholder("torchvision")
    tv.datasets = placeholder("torchvision.datasets", CocoDetection=object)
    tv.transforms = placeholder("torchvision.transforms")
    import scipy
    if not hasattr(scipy, "misc"):
        scipy.misc = placeholder("scipy.misc")

    shim = placeholder("maskrcnn")
    shim._C = refc
    shim.nms = lambda dets, threshold: refc.nms(dets, threshold)  # __init__.py:21-22

    class CropFunction:  # call shape of __init__.py:25-45 (forward only)
        def __init__(self, crop_height, crop_width, extrapolation_value=0):
            self.h, self.w, self.e = crop_height, crop_width, extrapolation_value

        def __call__(self, image, boxes, box_ind):
            crops = torch.zeros_like(image)  # :36
            with mute_stdout():
                refc.crop_forward(image, boxes, box_ind, self.e, self.h, self.w, crops)  # :38
            return crops

    shim.CropFunction = CropFunction
    sys.path.insert(0, REF)
    import config as rconfig
    import data as rdata
    import model as rmodel
    import utils as rutils
    return refc, rconfig, rutils, rdata, rmodel


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def rand_dets(g, n, extent=1024.0, cluster=True, integral=False):
    """Boxes with plenty of overlap (clusters around a few centres) and DISTINCT scores."""
    if n == 0:
        return torch.zeros(0, 5)
    k = max(1, n // 12)
    centres = torch.rand(k, 2, generator=g) * extent
    which = torch.randint(0, k, (n,), generator=g)
    c = centres[which] + torch.randn(n, 2, generator=g) * (12.0 if cluster else extent)
    hw = torch.exp(torch.rand(n, 2, generator=g) * 2.5 + 2.0)  # ~7 … 90 px
    y1, x1 = c[:, 0] - hw[:, 0] / 2, c[:, 1] - hw[:, 1] / 2
    y2, x2 = c[:, 0] + hw[:, 0] / 2, c[:, 1] + hw[:, 1] / 2
    b = torch.stack([y1, x1, y2, x2], 1).clamp(0, extent)
    if integral:
        b = b.round()
    while True:
        s = torch.rand(n, generator=g)
        if torch.unique(s).numel() == n:
            break
    return torch.cat([b, s[:, None]], 1)


def gen_nms(refc):
    g = torch.Generator().manual_seed(1234)
    out = {}
    case = 0

    def add(dets, thr, tag):
        nonlocal case
        keep = refc.nms(dets, thr)
        out[f"c{case}_dets"] = dets.numpy()
        out[f"c{case}_thr"] = np.float32(thr)
        out[f"c{case}_keep"] = keep.numpy()
        out[f"c{case}_tag"] = np.array(tag)
        case += 1

    for n in (1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 500, 1000):
        for thr in (0.3, 0.7):
            add(rand_dets(g, n), thr, f"random_n{n}")
    add(rand_dets(g, 500, integral=True), 0.3, "integral_pixels_n500")  # detection call site :1432
    add(rand_dets(g, 1000, cluster=False), 0.7, "sparse_n1000")
    d = rand_dets(g, 300)
    add(d[torch.argsort(d[:, 4], descending=True)].contiguous(), 0.7, "presorted_n300")  # :1346
    add(rand_dets(g, 2000), 0.5, "random_n2000")
    # IoU == threshold exactly: [0,0,9,9] vs [0,0,9,4]: areas 100 / 50, inter 50 → 0.5 (>= suppresses)
    add(torch.tensor([[0., 0., 9., 9., 0.9], [0., 0., 9., 4., 0.8], [20., 20., 29., 29., 0.7]]), 0.5,
        "iou_equals_threshold")
    # degenerate boxes (y2<y1): SURVEY Appendix A.4
    add(torch.tensor([[5., 5., 3., 3., 0.9], [4., 4., 6., 6., 0.8], [4., 4., 6., 6., 0.7],
                      [0., 0., -1., -1., 0.6], [0., 0., -1., -1., 0.5]]), 0.3, "degenerate")
    # identical boxes, zero-size boxes, normalised-looking coordinates (+1 convention still applies)
    add(torch.tensor([[1., 1., 1., 1., 0.5], [1., 1., 1., 1., 0.6], [1., 1., 2., 2., 0.4]]), 0.7, "zero_size")
    u = torch.rand(40, 4, generator=g)
    add(torch.cat([torch.minimum(u[:, :2], u[:, 2:]), torch.maximum(u[:, :2], u[:, 2:]),
                   torch.linspace(0.99, 0.01, 40)[:, None]], 1), 0.7, "normalised_coords")
    add(rand_dets(g, 150).double(), 0.7, "float64_n150")  # dispatch :75
    save("nms", **out)
    # strided input: columns selected from a wider tensor (nms_cpu.cpp:20-24 handles any strides)
    wide = torch.randn(96, 9, generator=g)
    d = rand_dets(g, 96)
    wide[:, 1:6] = d
    view = wide[:, 1:6]
    save("nms_strided", wide=wide.numpy(), keep=refc.nms(view, 0.6).numpy(), thr=np.float32(0.6))


def rand_boxes(g, n, lo=0.02, hi=0.6, spill=0.0):
    c = torch.rand(n, 2, generator=g)
    hw = torch.exp(torch.rand(n, 2, generator=g) * (np.log(hi) - np.log(lo)) + np.log(lo))
    b = torch.cat([c - hw / 2, c + hw / 2], 1)
    if spill == 0.0:
        b = b.clamp(0, 1)
    return b


def gen_crop(refc):
    g = torch.Generator().manual_seed(4321)
    out = {}
    case = 0

    def add(image, boxes, ind, extrap, ch, cw, tag):
        nonlocal case
        crops = torch.zeros(1)
        with mute_stdout():
            refc.crop_forward(image, boxes, ind, extrap, ch, cw, crops)
        out[f"c{case}_image"] = image.numpy()
        out[f"c{case}_boxes"] = boxes.numpy()
        out[f"c{case}_ind"] = ind.numpy()
        out[f"c{case}_args"] = np.array([extrap, ch, cw], dtype=np.float64)
        out[f"c{case}_crops"] = crops.numpy()
        out[f"c{case}_tag"] = np.array(tag)
        case += 1

    img = torch.randn(1, 16, 32, 32, generator=g)
    z = lambda n: torch.zeros(n, dtype=torch.int32)
    add(img, rand_boxes(g, 12), z(12), 0.0, 7, 7, "pool7")
    add(img, rand_boxes(g, 12), z(12), 0.0, 14, 14, "pool14")
    add(img, rand_boxes(g, 6), z(6), 0.0, 28, 28, "pool28")
    add(img, rand_boxes(g, 9, spill=1.0) * 1.4 - 0.2, z(9), -1.5, 7, 7, "partly_outside_extrap")
    add(img, torch.tensor([[-2., -2., -1., -1.], [1.5, 1.5, 2., 2.], [0., 0., 1., 1.]]), z(3), 9.0, 5, 3,
        "fully_outside_and_full")
    add(img, rand_boxes(g, 8), z(8), 0.0, 1, 1, "centre_sample_1x1")  # double-precision branch :61,84
    add(img, rand_boxes(g, 8), z(8), 0.0, 1, 5, "h1_w5")
    add(img, rand_boxes(g, 8), z(8), 0.0, 4, 1, "h4_w1")
    # integer-coincident samples: box on exact pixel centres → floor == ceil, lerp 0
    add(img, torch.tensor([[0., 0., 1., 1.], [4 / 31, 8 / 31, 10 / 31, 14 / 31]]), z(2), 0.0, 7, 7,
        "integer_coincident")
    # reversed box (y2<y1): negative scale, still defined
    add(img, torch.tensor([[0.8, 0.9, 0.2, 0.1]]), z(1), 0.0, 7, 7, "reversed_box")
    img3 = torch.randn(3, 5, 17, 23, generator=g)  # non-square, odd sizes, multi-image batch
    add(img3, rand_boxes(g, 10), torch.tensor([0, 1, 2, 2, 1, 0, 0, 2, 1, 1], dtype=torch.int32), 0.0, 7, 7,
        "multi_image_nonsquare")
    add(torch.randn(1, 3, 1, 1, generator=g), rand_boxes(g, 3), z(3), 0.5, 2, 2, "image_1x1")
    add(torch.randn(1, 256, 8, 8, generator=g), rand_boxes(g, 4), z(4), 0.0, 7, 7, "c256")
    add(img, torch.zeros(0, 4), z(0), 0.0, 7, 7, "no_boxes")
    save("crop_forward", **out)

    # backward (exported-symbol parity)
    grads = torch.randn(6, 4, 7, 7, generator=g)
    boxes = rand_boxes(g, 6, spill=1.0) * 1.2 - 0.1
    ind = torch.tensor([0, 1, 1, 0, 1, 0], dtype=torch.int32)
    gi = torch.zeros(2, 4, 12, 10)
    refc.crop_backward(grads, boxes, ind, gi)
    save("crop_backward", grads=grads.numpy(), boxes=boxes.numpy(), ind=ind.numpy(),
         grads_image=gi.numpy())


def gen_roi_align(rmodel):
    """model.py:276-393 incl. level edge cases."""
    g = torch.Generator().manual_seed(99)
    image_shape = np.array([256, 256, 3])
    fms = [torch.randn(1, 8, s, s, generator=g) for s in (64, 32, 16, 8)]
    boxes = rand_boxes(g, 60, lo=0.03, hi=0.95)
    # exact level boundaries: sqrt(h*w) * sqrt(A) / 224 = 2^(k-4): k = 2.5, 3.5, 4.5 ± 1 ulp-ish, and k=4
    side = lambda k: (224.0 / 256.0) * (2.0 ** (k - 4))
    extra = []
    for k in (2.0, 2.5, 3.0, 3.5, 4.0, 4.5, 5.0, 1.0, 6.5):
        for eps in (-1e-3, 0.0, 1e-3):
            s = min(side(k) * (1 + eps), 1.0)
            extra.append([0.0, 0.0, s, s])
    boxes = torch.cat([boxes, torch.tensor(extra, dtype=torch.float32)], 0)
    out = {"image_shape": image_shape, "boxes": boxes.numpy()}
    for i, fm in enumerate(fms):
        out[f"fm{i}"] = fm.numpy()
    for pool in (7, 14):
        pooled = rmodel.roi_align([boxes.unsqueeze(0)] + [f.clone() for f in fms], pool, image_shape)
        out[f"pooled{pool}"] = pooled.numpy()
    save("roi_align", **out)


def level_sweep_boxes(hh, ww):
    """Boxes whose sqrt(area) sits within +-4 ulp (and +-16 / +-64 ulp) of every pyramid-level boundary k = 2.5 / 3.5 /
    4.5 of an hh x ww image, square and at seven aspect ratios, at two offsets (the sweep of
    tests/test_gpu_fullsize.py::test_level_boundaries_ulp_sweep; the boxes are stored, the test does not rebuild them)."""
    area = float(hh * ww)
    rows = []
    for k in (2.5, 3.5, 4.5):
        s0 = (224.0 / np.sqrt(area)) * 2.0 ** (k - 4.0)                  # sqrt(h*w) at the boundary
        for ratio in (1.0, 2.0, 0.5, 4.0, 0.125, 3.0, 1.7):
            h0 = torch.tensor(s0 * np.sqrt(ratio), dtype=torch.float32)
            w0 = torch.tensor(s0 / np.sqrt(ratio), dtype=torch.float32)
            if h0 > 1 or w0 > 1:
                continue
            for dh in list(range(-4, 5)) + [-64, -16, 16, 64]:
                for dw in (-2, -1, 0, 1, 2):
                    h, w = h0.clone(), w0.clone()
                    for _ in range(abs(dh)):
                        h = torch.nextafter(h, torch.tensor(2.0 if dh > 0 else 0.0))
                    for _ in range(abs(dw)):
                        w = torch.nextafter(w, torch.tensor(2.0 if dw > 0 else 0.0))
                    for y1, x1 in ((0.0, 0.0), (0.25, 0.125)):
                        rows.append([y1, x1, y1 + h.item(), x1 + w.item()])
    rois = torch.tensor(rows, dtype=torch.float32)
    return rois[(rois[:, 2] <= 1) & (rois[:, 3] <= 1)].contiguous()


def gen_roi_levels(rmodel):
    """The pyramid level the REFERENCE gives every box of the ulp sweep: model.roi_align (model.py:276-393, unmodified)
    on four constant-valued maps (level l holds the value l), pool size 1 — the crop of a box inside the image is the
    constant of the map it was taken from, i.e. the level `roi_level` (model.py:331-338) assigned in this container
    (torch-CPU log2 = MKL VML here). A random set of ordinary boxes rides along."""
    out = {}
    g = torch.Generator().manual_seed(7)
    for hh, ww in ((1024, 1024), (832, 1344)):
        rois = torch.cat([level_sweep_boxes(hh, ww), rand_boxes(g, 400, lo=0.01, hi=0.9)], 0)
        fms = [torch.full((1, 1, 4, 4), float(l)) for l in (2, 3, 4, 5)]
        pooled = rmodel.roi_align([rois.unsqueeze(0)] + fms, 1, np.array([hh, ww, 3]))
        lv = pooled.reshape(-1)
        assert lv.numel() == rois.size(0) and bool(((lv == lv.round()) & (lv >= 2) & (lv <= 5)).all())
        out[f"rois_{hh}x{ww}"] = rois.numpy()
        out[f"levels_{hh}x{ww}"] = lv.to(torch.int32).numpy()
        print(f"levels {hh}x{ww}: {rois.size(0)} boxes, histogram {torch.bincount(lv.long(), minlength=6).tolist()}")
    out["torch_version"] = np.array(torch.__version__)
    save("roi_levels", **out)


def gen_anchors_boxes(rconfig, rutils, rdata):
    cfg = rconfig.CocoInferenceConfig()
    a = rutils.create_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, cfg.BACKBONE_SHAPES,
                                      cfg.BACKBONE_STRIDES, cfg.RPN_ANCHOR_STRIDE)
    a32 = a.astype(np.float32)
    idx = np.concatenate([np.arange(0, 12), np.arange(196608 - 6, 196608 + 6),
                          np.arange(a.shape[0] - 12, a.shape[0]),
                          np.random.RandomState(0).randint(0, a.shape[0], 64)])
    g = torch.Generator().manual_seed(5)
    boxes = torch.rand(50, 4, generator=g) * 200
    boxes[:, 2:] += boxes[:, :2]
    deltas = torch.randn(50, 4, generator=g) * 0.3
    refined = rdata.boxes_refine(boxes.clone(), deltas.clone())
    clamped = refined.clone()
    rdata.boxes_clamp_(clamped, [10, 20, 300, 350])
    scaled = rdata.boxes_scale(refined, [0.1, 0.1, 0.2, 0.2])
    save("anchors_boxes", shape=np.array(a.shape), sha256_f32=np.array(hashlib.sha256(a32.tobytes()).hexdigest()),
         idx=idx, rows_f64=a[idx], boxes=boxes.numpy(), deltas=deltas.numpy(), refined=refined.numpy(),
         clamped=clamped.numpy(), scaled=scaled.numpy())


def randomize_bn_(module, seed):
    """γ~U(0.5,1.5), β~N(0,0.1), μ~N(0,0.1), σ²~U(0.5,1.5)."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


def gen_graph(rmodel):
    """Small-channel instances of the reference nn.Modules (weights stored) + SamePad/stem behaviour."""
    out = {}
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(77)
    # Bottleneck: (inplanes, planes, stride, has_downsample, H, W)
    cases = [(16, 8, 1, True, 12, 12), (32, 8, 1, False, 12, 12), (32, 16, 2, True, 12, 12),
             (32, 16, 2, True, 11, 13)]
    for i, (cin, planes, stride, ds, h, w) in enumerate(cases):
        down = None
        if ds:
            down = torch.nn.Sequential(torch.nn.Conv2d(cin, planes * 4, kernel_size=1, stride=stride),
                                       torch.nn.BatchNorm2d(planes * 4, eps=0.001, momentum=0.01))
        blk = rmodel.Bottleneck(cin, planes, stride, down)
        for m in blk.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.bias.data.normal_(0, 0.1, generator=g)
        randomize_bn_(blk, 100 + i)
        blk.eval()
        x = torch.randn(2, cin, h, w, generator=g)
        with torch.no_grad():
            y = blk(x.clone())
        out[f"b{i}_x"] = x.numpy()
        out[f"b{i}_y"] = y.numpy()
        out[f"b{i}_stride"] = np.int64(stride)
        for k, v in blk.state_dict().items():
            out[f"b{i}_sd_{k}"] = v.numpy()
    out["n_bottleneck"] = np.int64(len(cases))
    # SamePad2d pad amounts on even/odd sizes (model.py:64-87)
    pads = []
    for (k, s, h, w) in [(3, 1, 8, 8), (3, 2, 8, 8), (3, 2, 9, 8), (3, 2, 8, 9), (3, 2, 7, 7), (7, 2, 10, 12)]:
        y = rmodel.SamePad2d(k, s)(torch.ones(1, 1, h, w))
        # recover (top, bottom, left, right) pads from where the ones landed
        nz = torch.nonzero(y[0, 0])
        top, left = int(nz[:, 0].min()), int(nz[:, 1].min())
        pads.append([k, s, h, w, top, y.size(2) - h - top, left, y.size(3) - w - left])
    out["same_pad"] = np.array(pads)
    # RPN head on one small level (model.py:609-649)
    rpn = rmodel.RPN(3, 1, 32)
    for m in rpn.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.normal_(m.weight, 0, 0.02, generator=g)
            m.bias.data.normal_(0, 0.1, generator=g)
    rpn.eval()
    x = torch.randn(1, 32, 6, 5, generator=g)
    with torch.no_grad():
        logits, probs, bbox = rpn(x)
    out["rpn_x"] = x.numpy()
    out["rpn_logits"], out["rpn_probs"], out["rpn_bbox"] = logits.numpy(), probs.numpy(), bbox.numpy()
    for k, v in rpn.state_dict().items():
        out[f"rpn_sd_{k}"] = v.numpy()
    save("graph_small", **out)


def gen_refine(rconfig, rmodel):
    """rpn_refine / mrn_refine of the reference MaskRCNN object on a 256x256 configuration
    (16368 anchors) so inputs stay small. Also records the dets handed to nms at each call site."""

    class SmallCfg(rconfig.CocoInferenceConfig):
        GPU_COUNT = 0
        IMAGE_MIN_DIM = 256
        IMAGE_MAX_DIM = 256

    cfg = SmallCfg()
    tmp = tempfile.mkdtemp(prefix="golden_logs_")
    torch.manual_seed(0)
    net = rmodel.MaskRCNN(config=cfg, model_dir=tmp)
    g = torch.Generator().manual_seed(31)
    A = net.anchors.size(0)
    logits = torch.randn(1, A, 2, generator=g) * 2
    rpn_class = torch.softmax(logits, dim=2)
    rpn_bbox = torch.randn(1, A, 4, generator=g) * 0.8
    captured = []
    import maskrcnn as shim
    orig = shim.nms

    def spy(dets, thr):
        k = orig(dets, thr)
        captured.append((dets.clone(), float(thr), k.clone()))
        return k

    shim.nms = spy
    rois = net.rpn_refine(rpn_class, rpn_bbox)
    out = dict(anchors_shape=np.array(net.anchors.shape), rpn_class=rpn_class.numpy(),
               rpn_bbox=rpn_bbox.numpy(), rois=rois.numpy(), rpn_dets=captured[0][0].numpy(),
               rpn_keep=captured[0][2].numpy())
    captured.clear()
    # detection refinement on the rois above with random class probabilities over 6 live classes
    n = rois.size(1)
    pl = torch.full((n, 81), -20.0)
    pl[:, :6] = torch.randn(n, 6, generator=g) * 3
    probs = torch.softmax(pl, dim=1)
    deltas = torch.randn(n, 81, 4, generator=g) * 0.5
    window = (16, 0, 240, 256)
    cls, sc, bx = net.mrn_refine(rois, probs, deltas, window)
    out.update(probs=probs.numpy(), deltas=deltas.numpy(), window=np.array(window), det_class_ids=cls.numpy(),
               det_scores=sc.numpy(), det_boxes=bx.numpy(), n_class_calls=np.int64(len(captured)))
    for i, (d, t, k) in enumerate(captured):
        out[f"cls{i}_dets"], out[f"cls{i}_keep"] = d.numpy(), k.numpy()
    shim.nms = orig
    save("refine", **out)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def gen_schema(rconfig, rmodel):
    """state_dict layout of the reference MaskRCNN (R101, hard-coded at model.py:985) and of a
    ResNet('resnet50') trunk: key names + shapes only (no weights)."""

    class Cfg0(rconfig.CocoInferenceConfig):
        GPU_COUNT = 0

    tmp = tempfile.mkdtemp(prefix="golden_logs_")
    net = rmodel.MaskRCNN(config=Cfg0(), model_dir=tmp)
    sd = net.state_dict()
    keys = list(sd.keys())
    shapes = [list(sd[k].shape) + [-1] * (4 - sd[k].dim()) for k in keys]
    r50 = rmodel.ResNet("resnet50", stage5=True).state_dict()
    save("schema", keys=np.array(keys), shapes=np.array(shapes, dtype=np.int64),
         r50_trunk_keys=np.array(list(r50.keys())))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def gen_image(rconfig, rutils, rmodel, rdata):
    """Pre-/post-processing around predict. The pixel arithmetic is Pillow's (third-party,
    installed here: the version is stored in the fixture); the two wrappers the reference calls it through are
    absent from this image and are spelled as the PIL calls they make:
        scipy.misc.imresize(a, (h, w))            == Image.fromarray(a).resize((w, h), Image.BILINEAR)   utils.py:73
        transform.Resize((h, w))(img)             == img.resize((w, h), Image.BILINEAR)                  data.py:277,295
        transform.CenterCrop((th, tw))(img)       == img.crop((j, i, j+tw, i+th)), i = int(round((h-th)/2.)), ...
        transform.Pad((l, t, r, b))(img)          == zero canvas with img at (t, l)                      data.py:305
    Everything else (padding, windows, mold_image, box decoding) is executed by the reference's own functions."""
    import PIL
    from PIL import Image
    g = np.random.default_rng(2024)
    out = {"pillow_version": np.array(PIL.__version__)}

    # ---- plain resizes: up, down, mixed, 1-pixel, identity, RGB and L
    cases = [(7, 9, 1, 28, 28), (28, 28, 1, 5, 9), (28, 28, 1, 1, 1), (28, 28, 1, 28, 28), (28, 28, 1, 27, 29),
             (28, 28, 1, 200, 3), (60, 80, 3, 96, 128), (300, 200, 3, 128, 85), (33, 47, 3, 33, 47),
             (64, 64, 3, 17, 130), (1, 1, 3, 9, 5), (50, 3, 1, 2, 77)]
    out["resize_cases"] = np.array(cases, dtype=np.int64)
    for i, (h, w, c, oh, ow) in enumerate(cases):
        a = g.integers(0, 256, (h, w, c), dtype=np.uint8)
        if i % 3 == 0:
            a[a < 40] = 0
            a[a > 215] = 255
        src = a[:, :, 0] if c == 1 else a
        out[f"resize_{i}_in"] = src
        out[f"resize_{i}_out"] = np.array(Image.fromarray(src).resize((ow, oh), Image.BILINEAR))

    # ---- float -> 'L' (Image.fromarray(F).convert('L'), data.py:294)
    f = (g.random((28, 28), dtype=np.float32) * 300 - 20).astype(np.float32)
    f[0, :6] = [0.0, 0.999, 1.0, 254.999, 255.0, 127.5]
    out["f2l_in"] = f
    out["f2l_out"] = np.array(Image.fromarray(f).convert("L"))

    # ---- resize_image + mold_image + detect()'s transpose (utils.py:42-90, model.py:1750-1754,1108-1110)
    cfg = rconfig.CocoInferenceConfig()
    mold_cases = [(110, 128, 100, 128), (60, 80, 100, 128), (300, 200, 100, 128), (128, 128, 100, 128), (81, 47, 64, 128)]
    out["mold_cases"] = np.array(mold_cases, dtype=np.int64)
    for i, (h, w, min_dim, max_dim) in enumerate(mold_cases):
        a = g.integers(0, 256, (h, w, 3), dtype=np.uint8)
        scale = max(1, min_dim / min(h, w))                                    # utils.py:62-69, evaluated here
        if round(max(h, w) * scale) > max_dim:
            scale = max_dim / max(h, w)
        if scale != 1:
            resized = np.array(Image.fromarray(a).resize((round(w * scale), round(h * scale)), Image.BILINEAR))
        else:
            resized = a
        # the reference pads (and reports window/padding) itself; scale is 1 for an image already at its final size
        padded, window, s1, padding = rutils.resize_image(resized, min_dim=None, max_dim=max_dim, padding=True)
        assert s1 == 1
        molded = rmodel.mold_image(padded, cfg)                                # float64: MEAN_PIXEL is a float64 array
        molded = torch.from_numpy(molded.transpose(2, 0, 1)).float()           # model.py:1108
        out[f"mold_{i}_in"] = a
        out[f"mold_{i}_out"] = molded.numpy()
        out[f"mold_{i}_window"] = np.array(window, dtype=np.int64)
        out[f"mold_{i}_scale"] = np.array(scale, dtype=np.float64)
        out[f"mold_{i}_padding"] = np.array(padding, dtype=np.int64)
    out["mean_pixel"] = np.asarray(cfg.MEAN_PIXEL, dtype=np.float64)

    # ---- full_masks (data.py:287-314) on a 128 x 192 canvas
    H, W, C = 128, 192, 5
    boxes = np.array([[10, 20, 60, 90], [0, 0, 128, 192], [100, 150, 128, 192], [5, 7, 10, 16], [64, 64, 65, 65],
                      [30, 40, 58, 68], [31, 41, 58, 70], [0, 100, 3, 192], [90, 0, 128, 2], [17, 33, 117, 37]],
                     dtype=np.float32)
    n = boxes.shape[0]
    masks = 1.0 / (1.0 + np.exp(-g.normal(0, 2.5, (n, C, 28, 28)))).astype(np.float32)
    masks[1, :, :3, :3] = 0.0
    masks[1, :, -3:, -3:] = 1.0
    masks[2, :, 10:12, 10:12] = 0.5
    masks[2, :, 12:14, 10:12] = 127.5 / 255.0
    masks = masks.astype(np.float32)
    class_id = g.integers(1, C, n).astype(np.int64)
    full = []
    for i in range(n):
        m = torch.from_numpy(masks[i][class_id[i]]) * 255.0                    # data.py:291
        box = rdata.Box.fromlist(torch.from_numpy(boxes[i]).tolist())
        img = Image.fromarray(m.numpy()).convert("L")
        img = img.resize((int(box.width()), int(box.height())), Image.BILINEAR)   # :295
        top_pad, left_pad = int(box.top()), int(box.left())                    # :298-303
        canvas = np.zeros((H, W), dtype=np.uint8)
        canvas[top_pad:top_pad + img.height, left_pad:left_pad + img.width] = np.array(img)
        full.append(canvas > 127)                                              # :307-308
    out["fm_masks"], out["fm_class_id"], out["fm_boxes"] = masks, class_id, boxes
    out["fm_out"] = np.stack(full)
    out["fm_canvas"] = np.array([H, W], dtype=np.int64)

    # ---- decode_boxes (run by the reference) and decode_masks (data.py:264-284, 331-343)
    window = (16, 0, 112, 192)
    for tag, scale in (("up", 1.6), ("down", 0.512)):
        b = torch.from_numpy(boxes.copy())
        out[f"dec_{tag}_boxes"] = rdata.decode_boxes(b, scale, rdata.Box.fromlist(list(window))).numpy()
        th, tw = window[2] - window[0], window[3] - window[1]
        dec = []
        for i in range(n):
            img = Image.fromarray(out["fm_out"][i]).convert("L")               # bool array -> mode '1' -> 0/255
            i0, j0 = int(round((H - th) / 2.0)), int(round((W - tw) / 2.0))
            img = img.crop((j0, i0, j0 + tw, i0 + th))
            nh, nw = round(img.height * 1.0 / scale), round(img.width * 1.0 / scale)
            dec.append(np.array(img.resize((nw, nh), Image.BILINEAR)))
        out[f"dec_{tag}_masks"] = np.stack(dec)
        out[f"dec_{tag}_scale"] = np.array(scale, dtype=np.float64)
    out["dec_window"] = np.array(window, dtype=np.int64)
    save("image", **out)


def gen_config1(rconfig, rutils, rmodel):
    """BASELINE configs[0]: `predict.py images/car58a54312d.jpg` — the geometry detect() sees for the reference's own
    sample image (predict.py:55-60 → model.detect, model.py:1095-1110): 1200 x 1920 RGB → scale 1024/1920 → 640 x 1024
    → zero-padded to 1024 x 1024 with window (192, 0, 832, 1024). The image file is one of the reference's data files;
    its decoded pixels are stored as the input vector. Outputs: the resized uint8 image (Pillow BILINEAR — what
    scipy.misc.imresize did, utils.py:73), and from the reference's own resize_image / mold_image on it: window,
    padding, and the molded fp32 [3,1024,1024] tensor as a sha256 plus a 16-pixel-stride sample (the full tensor is
    12.6 MB)."""
    from PIL import Image
    cfg = rconfig.CocoInferenceConfig()
    a = np.array(Image.open(os.path.join(REF, "images", "car58a54312d.jpg")).convert("RGB"))
    h, w = a.shape[:2]
    scale = max(1, cfg.IMAGE_MIN_DIM / min(h, w))                            # utils.py:62-69, evaluated here
    if round(max(h, w) * scale) > cfg.IMAGE_MAX_DIM:
        scale = cfg.IMAGE_MAX_DIM / max(h, w)
    resized = np.array(Image.fromarray(a).resize((round(w * scale), round(h * scale)), Image.BILINEAR))
    padded, window, s1, padding = rutils.resize_image(resized, min_dim=None, max_dim=cfg.IMAGE_MAX_DIM, padding=True)
    assert s1 == 1
    molded = rmodel.mold_image(padded, cfg)                                   # model.py:1750-1754 (float64 mean)
    molded = torch.from_numpy(molded.transpose(2, 0, 1)).float().contiguous()  # model.py:1108
    save("config1", image=a, resized=resized, window=np.array(window, dtype=np.int64),
         scale=np.array(scale, dtype=np.float64), padding=np.array(padding, dtype=np.int64),
         mean_pixel=np.asarray(cfg.MEAN_PIXEL, dtype=np.float64),
         min_dim=np.int64(cfg.IMAGE_MIN_DIM), max_dim=np.int64(cfg.IMAGE_MAX_DIM),
         molded_sha256=np.array(hashlib.sha256(molded.numpy().tobytes()).hexdigest()),
         molded_sample=molded.numpy()[:, ::16, ::16].copy(),
         pillow_version=np.array(__import__("PIL").__version__))


def main():
    only = set(sys.argv[1:])
    want = lambda n: not only or n in only
    refc, rconfig, rutils, rdata, rmodel = load_reference()
    if want("nms"):
        gen_nms(refc)
    if want("crop"):
        gen_crop(refc)
    if want("roi_align"):
        gen_roi_align(rmodel)
    if want("roi_levels"):
        gen_roi_levels(rmodel)
    if want("anchors"):
        gen_anchors_boxes(rconfig, rutils, rdata)
    if want("graph"):
        gen_graph(rmodel)
    if want("refine"):
        gen_refine(rconfig, rmodel)
    if want("schema"):
        gen_schema(rconfig, rmodel)
    if want("image"):
        gen_image(rconfig, rutils, rmodel, rdata)
    if want("config1"):
        gen_config1(rconfig, rutils, rmodel)


if __name__ == "__main__":
    main()
