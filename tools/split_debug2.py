import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskrcnn_amd import modules, ops
from maskrcnn_amd.config import InferenceConfig
from maskrcnn_amd.pipeline import MaskRCNNInference
dev = torch.device("cuda:0")
cfg = InferenceConfig(image_height=1024, image_width=1024, backbone="resnet50", pre_nms_limit=1000, proposal_count=1000, detection_max_instances=50)
sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
g = torch.Generator().manual_seed(5)
sd["classifier.linear_class.weight"] = torch.randn(81, 1024, generator=g) * 0.05
sd["classifier.linear_class.bias"] = torch.randn(81, generator=g) * 0.5
sd["classifier.linear_bbox.weight"] = torch.randn(324, 1024, generator=g) * 0.02
sd["rpn.conv_bbox.bias"] = torch.randn(12, generator=g) * 0.3
g0 = torch.Generator().manual_seed(0)
images = torch.randint(0, 256, (2, 1024, 1024, 3), generator=g0).float() - torch.tensor(cfg.mean_pixel)
images = images.permute(0, 3, 1, 2).contiguous().to(dev)
windows = torch.tensor([[0., 0., 1024., 1024.], [192., 0., 832., 1024.]], device=dev)
net = MaskRCNNInference(sd, cfg, dev)
ref = [f.clone() for f in net.backbone(images)]
torch.cuda.synchronize()
for it in range(3):
    det, mid = net.predict(images, windows, return_intermediates=True)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(mid["feature_maps"], ref)):
        d = (a - b).abs()
        bad = (d > 0).nonzero()
        print("iter", it, "P%d" % (i + 2), "max diff", d.max().item(), "n_bad", bad.size(0), "first", bad[:2].tolist(), "last", bad[-2:].tolist())
    print("roi_counts", mid["roi_counts"].tolist(), "det counts", det.counts.tolist())
