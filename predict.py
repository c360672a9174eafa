#!/usr/bin/env python3
"""`python predict.py [-model state_dict.pth] image` — the reference's predict.py (predict.py:30-72 there) on the MI355X path:
load the checkpoint (the reference's `MaskRCNN.state_dict()` key layout loads unchanged), read the image, run
`MaskRCNNInference.detect` — resize + pad + mean-subtract, ResNet-101-FPN trunk, RPN, proposal NMS, RoIAlign, heads, per-class
NMS, mask head, full-size mask pasting, all on the GPU with one host synchronisation — and print one line per detection
(class id, COCO class name, box, score), as the reference does. Its matplotlib window (`utils.display_instances`) is out of
scope; `--save out.npz` stores class ids, scores, boxes and the full-size boolean masks instead.

The reference's weights (models/mask_rcnn_coco.pth, a manual download) are not available offline: without `-model` the script
refuses to guess, and `--random-weights` runs the same path on seeded random weights of the same architecture (plumbing /
timing runs; the detections mean nothing).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MS-COCO's 80 category names in the order of the class ids 1..80 the COCO-trained checkpoint predicts (0 = background)
COCO_NAMES = (
    'BG', 'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light',
    'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant',
    'bear', 'zebra', 'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard',
    'sports ball', 'kite', 'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle',
    'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange', 'broccoli',
    'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch', 'potted plant', 'bed', 'dining table', 'toilet',
    'tv', 'laptop', 'mouse', 'remote', 'keyboard', 'cell phone', 'microwave', 'oven', 'toaster', 'sink',
    'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear', 'hair drier', 'toothbrush',
)


def read_image(path):
    """RGB uint8 [h, w, 3] (skimage.io.imread + grey2rgb in the reference, predict.py:57-59)."""
    import numpy as np
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"), dtype=np.uint8)   # a writable copy


def main(argv=None):
    ap = argparse.ArgumentParser(description="Mask RCNN Predictor (MI355X path)")
    ap.add_argument("-model", type=str, default=None, help="trained model: a torch.save()d state_dict of the reference's MaskRCNN")
    ap.add_argument("--random-weights", action="store_true", help="seeded random weights instead of a checkpoint (plumbing runs)")
    ap.add_argument("--backbone", default="resnet101", choices=["resnet50", "resnet101"], help="the reference hard-codes resnet101")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3", "f32+f16x3", "f16"])
    ap.add_argument("--save", default=None, help="write class_ids / scores / boxes / masks to this .npz")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("image", type=str, help="image file")
    args = ap.parse_args(argv)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("predict.py needs a GPU (the product has no CPU path)")
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference

    if args.model:
        sd = torch.load(args.model, map_location="cpu")
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    elif args.random_weights:
        sd = modules.synthetic_state_dict(args.backbone)
    else:
        raise SystemExit("predict.py: give -model <state_dict.pth> (the reference's models/mask_rcnn_coco.pth) or --random-weights")
    cfg = InferenceConfig(backbone=args.backbone)          # CocoInferenceConfig: 1024 x 1024 canvas, 500 proposals, 50 detections
    net = MaskRCNNInference(sd, cfg, args.device, precision=args.precision)
    img = read_image(args.image)
    class_ids, scores, boxes, masks = net.detect([img])[0]
    out = []
    if class_ids is not None:
        ids, sc, bx = class_ids.tolist(), scores.tolist(), boxes.tolist()
        for j, b, s in zip(ids, bx, sc):
            name = COCO_NAMES[j] if 0 <= j < len(COCO_NAMES) else "?"
            print(j, name, [int(v) for v in b], round(float(s), 6))
            out.append((j, name, b, s))
    else:
        print("no instances")
    if args.save:
        import numpy as np
        if class_ids is None:
            np.savez_compressed(args.save, class_ids=np.zeros(0, np.int64), scores=np.zeros(0, np.float32),
                                boxes=np.zeros((0, 4), np.float32), masks=np.zeros((0,) + img.shape[:2], bool))
        else:
            np.savez_compressed(args.save, class_ids=class_ids.cpu().numpy(), scores=scores.cpu().numpy(),
                                boxes=boxes.cpu().numpy(), masks=masks.cpu().numpy())
    return out


if __name__ == "__main__":
    main()
