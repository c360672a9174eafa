#!/usr/bin/env python3
"""Times the REFERENCE's own CPU hot path in the build container: its model.py modules imported
from /root/reference, its compiled CPU extension (oracle/_ref), seeded random weights, 1024 x 1024 synthetic images, the
stages of MaskRCNN.predict (model.py:1140-1203) up to and including the mask head (data.full_masks needs torchvision:
absent). ResNet-101 is what MaskRCNN.build hard-codes (model.py:985); the ResNet-50 row swaps `fpn` for
FPN(*ResNet("resnet50", stage5=True).stages()) — the backbone of BASELINE configs[2].
Build container only (the reference does not travel to the GPU box). Prints one JSON line per backbone."""
import importlib.util
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    refc, rconfig, rutils, rdata, rmodel = mg.load_reference()

    class Cfg(rconfig.CocoInferenceConfig):
        GPU_COUNT = 0

    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
    for arch in ("resnet50", "resnet101"):
        torch.manual_seed(0)
        net = rmodel.MaskRCNN(config=Cfg(), model_dir=tempfile.mkdtemp(prefix="ref_time_"))
        if arch == "resnet50":
            net.fpn = rmodel.FPN(*rmodel.ResNet("resnet50", stage5=True).stages(), out_channels=256)
            net.initialize_weights()
        mg.randomize_bn_(net, 1)
        sd = net.state_dict()
        g = torch.Generator().manual_seed(5)
        sd["rpn.conv_class.weight"].mul_(0.02)
        sd["rpn.conv_bbox.weight"].mul_(0.02)
        sd["classifier.linear_class.weight"].copy_(torch.randn(81, 1024, generator=g) * 0.002)
        sd["classifier.linear_class.bias"].copy_(torch.randn(81, generator=g) * 0.5)
        sd["classifier.linear_bbox.weight"].copy_(torch.randn(324, 1024, generator=g) * 0.001)
        net.eval()
        window = (0, 0, 1024, 1024)
        stages = {k: 0.0 for k in ("fpn", "rpn_detect", "rpn_refine", "mrn_detect", "mrn_refine", "mask")}
        dets = 0
        with torch.no_grad():
            for i in range(n_img + 1):           # image 0 = warm-up (oneDNN primitive creation), not counted
                g = torch.Generator().manual_seed(1000 + i)
                image = torch.randint(0, 256, (1, 1024, 1024, 3), generator=g).float() - torch.tensor([123.7, 116.8, 103.9])
                image = image.permute(0, 3, 1, 2).contiguous()
                t = [time.perf_counter()]
                fms = net.fpn(image); t.append(time.perf_counter())
                _, cls, box = net.rpn_detect(fms); t.append(time.perf_counter())
                rois = net.rpn_refine(cls, box); t.append(time.perf_counter())
                with mg.mute_stdout():
                    _, probs, bbox = net.mrn_detect([f.clone() for f in fms[:4]], rois)
                t.append(time.perf_counter())
                ids, scores, boxes = net.mrn_refine(rois, probs, bbox, window); t.append(time.perf_counter())
                if ids is not None:
                    with mg.mute_stdout():
                        net.mask([f.clone() for f in fms[:4]], boxes.float() * 1.0 / 1024)
                t.append(time.perf_counter())
                if i:
                    for k, a, b in zip(stages, t, t[1:]):
                        stages[k] += (b - a) / n_img
                    dets += 0 if ids is None else int(ids.numel())
        total = sum(stages.values())
        print(json.dumps({"backbone": arch, "images": n_img, "s_per_image": round(total, 3), "images_per_s": round(1 / total, 3),
                          "stages_s": {k: round(v, 4) for k, v in stages.items()}, "proposals": int(rois.size(1)),
                          "mean_detections": dets / n_img, "threads": torch.get_num_threads(), "cpu": cpu[0], "cpus": len(cpu),
                          "torch": torch.__version__,
                          "path": "reference model.py (fpn, rpn_detect, rpn_refine, mrn_detect, mrn_refine, mask) + "
                                  "reference c++ext CPU nms / crop_forward"}), flush=True)


if __name__ == "__main__":
    main()
