"""CPU test, build container only: the oracle's graph-level restatement (oracle/oracle.py) against the REFERENCE
ITSELF — its model.py / data.py / utils.py imported from /root/reference and its compiled CPU extension
(oracle/_ref) — on a 256x256 configuration with the reference's hard-coded ResNet-101. Stage by stage and
bit for bit (both sides run the same torch-CPU kernels in the same order). Skipped where the reference tree is
absent (the GPU box): there the committed golden vectors pin the oracle."""
import importlib.util
import os
import shutil
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MASKRCNN_REFERENCE", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "c++ext")),
                                reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    # the reference modules are imported under their own top-level names (config, utils, data, model) and with
    # placeholder third-party modules and a `maskrcnn` shim: restore sys.modules / sys.path afterwards so the
    # rest of the test session still sees this repo's own `maskrcnn` package
    saved_modules, saved_path = dict(sys.modules), list(sys.path)
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden",
                                                                              "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    refc, rconfig, rutils, rdata, rmodel = mg.load_reference()

    class Cfg(rconfig.CocoInferenceConfig):
        GPU_COUNT = 0
        IMAGE_MIN_DIM = 256
        IMAGE_MAX_DIM = 256

    tmp = tempfile.mkdtemp(prefix="live_logs_")
    torch.manual_seed(0)
    net = rmodel.MaskRCNN(config=Cfg(), model_dir=tmp)
    mg.randomize_bn_(net, 1)
    g = torch.Generator().manual_seed(5)
    sd = net.state_dict()
    # spread the heads (random init saturates every softmax → ties → order undefined on both sides)
    sd["rpn.conv_class.weight"].mul_(0.02)
    sd["rpn.conv_bbox.weight"].mul_(0.02)
    sd["classifier.linear_class.weight"].copy_(torch.randn(81, 1024, generator=g) * 0.002)
    sd["classifier.linear_class.bias"].copy_(torch.randn(81, generator=g) * 0.5)
    sd["classifier.linear_bbox.weight"].copy_(torch.randn(324, 1024, generator=g) * 0.001)
    net.eval()
    yield net, {k: v.detach().clone() for k, v in net.state_dict().items()}, mg
    shutil.rmtree(tmp, ignore_errors=True)
    for name in list(sys.modules):
        if name not in saved_modules:
            del sys.modules[name]
    for name, mod in saved_modules.items():
        sys.modules[name] = mod
    sys.path[:] = saved_path


def test_oracle_predict_stages_equal_reference(ref, oracle):
    net, sd, mg = ref
    cfg = oracle.Cfg(256, 256)
    g = torch.Generator().manual_seed(9)
    image = torch.randint(0, 256, (1, 3, 256, 256), generator=g).float() - 110.0
    window = (16, 0, 240, 256)
    with torch.no_grad():
        r_fms = net.fpn(image)
        o_fms = oracle.fpn_forward(image, sd, "resnet101")
        for a, b in zip(r_fms, o_fms):
            assert torch.equal(a, b)
        _, r_cls, r_box = net.rpn_detect(r_fms)
        _, o_cls, o_box = oracle.rpn_detect(o_fms, sd)
        assert torch.equal(r_cls, o_cls) and torch.equal(r_box, o_box)
        assert torch.equal(net.anchors, oracle.anchors_for(cfg))
        r_rois = net.rpn_refine(r_cls, r_box)
        o_rois = oracle.rpn_refine(o_cls, o_box, oracle.anchors_for(cfg), cfg)
        assert torch.equal(r_rois, o_rois)
        with mg.mute_stdout():
            _, r_probs, r_bbox = net.mrn_detect([f.clone() for f in r_fms[:4]], r_rois)
        _, o_probs, o_bbox = oracle.classifier_forward(o_fms[:4], o_rois, sd, cfg)
        assert torch.equal(r_probs, o_probs) and torch.equal(r_bbox, o_bbox)
        r_ids, r_scores, r_boxes = net.mrn_refine(r_rois, r_probs, r_bbox, window)
        o_ids, o_scores, o_boxes = oracle.mrn_refine(o_rois, o_probs, o_bbox, window, cfg)
        assert r_ids is not None and r_ids.numel() > 0
        assert torch.equal(r_ids, o_ids) and torch.equal(r_scores, o_scores) and torch.equal(r_boxes, o_boxes)
        with mg.mute_stdout():
            r_masks = net.mask([f.clone() for f in r_fms[:4]], r_boxes.float() * 1.0 / 256)
        o_masks = oracle.mask_forward(o_fms[:4], o_boxes.float() * 1.0 / 256, sd, cfg)
        assert torch.equal(r_masks, o_masks)
