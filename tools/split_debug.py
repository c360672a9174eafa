import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maskrcnn_amd import modules, ops
from maskrcnn_amd.config import InferenceConfig
dev = torch.device("cuda:0")
cfg = InferenceConfig(image_height=1024, image_width=1024, backbone="resnet50")
sd = modules.synthetic_state_dict("resnet50", seed=0, bn_seed=1)
g0 = torch.Generator().manual_seed(0)
images = torch.randint(0, 256, (2, 1024, 1024, 3), generator=g0).float() - torch.tensor(cfg.mean_pixel)
images = images.permute(0, 3, 1, 2).contiguous().to(dev)
bb = modules.FusedBackbone(sd, "resnet50", dev, precision="f32")
def stages(x):
    outs = {}
    st = bb.stem
    y = ops.stem_pool_f32(x.contiguous(), st.w.w, st.scale, st.shift, st.algo_cin)
    outs["stem"] = y
    for si, blocks in enumerate(bb.stages):
        for bi, blk in enumerate(blocks):
            y = blk(y)
            outs[f"C{si+2}.{bi}"] = y
    return outs
both = stages(images)
one = stages(images[1:2])
torch.cuda.synchronize()
for k in both:
    d = (both[k][1:2] - one[k]).abs().max().item()
    print(k, "batch2[1] vs alone:", d, "range", one[k].abs().max().item())
fm2 = bb(images); fm1 = bb(images[1:2])
for i, (a, b) in enumerate(zip(fm2, fm1)):
    print("P%d" % (i + 2), (a[1:2] - b).abs().max().item())
