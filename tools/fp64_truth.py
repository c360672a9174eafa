#!/usr/bin/env python3
"""How far each fp32 implementation of the trunk is from the TRUE (float64) result on the bench's own input range
(BASELINE configs[2]: R50-FPN, 1024^2, uint8-range pixels minus MEAN_PIXEL; optionally configs[4]: R101-FPN, 832 x 1344):
torch-CPU fp32 (the reference's arithmetic, = the oracle) and the HIP trunk in its three contraction forms — the default
(F(4x4) Winograd on the large maps, F(2x2) elsewhere), F(2x2) everywhere (MRCNN_WINOGRAD4=0) and the exact direct kernel
(MRCNN_WINOGRAD=0: bitwise an fmaf chain). One JSON line per (config, mode, level). Test infrastructure (imports oracle/)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config5", type=int, default=0)
    args = ap.parse_args()
    from maskrcnn_amd import modules
    from maskrcnn_amd.config import InferenceConfig
    from oracle import oracle
    dev = torch.device("cuda:0")
    cases = [("config3", "resnet50", 1024, 1024, 0)] + ([("config5", "resnet101", 832, 1344, 55)] if args.config5 else [])
    for name, arch, h, w, seed in cases:
        cfg = InferenceConfig(image_height=h, image_width=w, backbone=arch)
        sd = modules.synthetic_state_dict(arch, seed=0, bn_seed=1)
        g = torch.Generator().manual_seed(seed)
        img = (torch.randint(0, 256, (1, h, w, 3), generator=g).float() - torch.tensor(cfg.mean_pixel)).permute(0, 3, 1, 2).contiguous()
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items() if k.startswith("fpn.")}
        with torch.no_grad():
            truth = oracle.fpn_forward(img.double(), sd64, arch)
            want32 = oracle.fpn_forward(img, sd, arch)
        rows = {"torch_cpu_fp32": [w_[0] for w_ in want32]}
        for mode, flags in (("hip_default", {}), ("hip_f2x2_everywhere", {"WINOGRAD4": False, "WINOGRAD4_TRUNK": False}),
                            ("hip_direct", {"WINOGRAD": False, "WINOGRAD4": False, "WINOGRAD4_TRUNK": False})):
            saved = {k: getattr(modules, k) for k in ("WINOGRAD", "WINOGRAD4", "WINOGRAD4_TRUNK")}
            for k, v in flags.items():
                setattr(modules, k, v)
            try:
                bb = modules.FusedBackbone(sd, arch, dev, precision="f32")
                maps = bb(img.to(dev))
                torch.cuda.synchronize()
                rows[mode] = [m[0].permute(2, 0, 1).cpu() for m in maps]
                del bb
            finally:
                for k, v in saved.items():
                    setattr(modules, k, v)
        for mode, maps in rows.items():
            for lvl, (t, m) in enumerate(zip(truth, maps)):
                e = (m.double() - t[0]).abs()
                print(json.dumps({"config": name, "impl": mode, "level": f"P{lvl + 2}", "max_abs_err_vs_fp64": e.max().item(),
                                  "rms_err_vs_fp64": e.pow(2).mean().sqrt().item(), "max_abs_fp64": t.abs().max().item(),
                                  "err_in_ulps_of_max": e.max().item() / (t.abs().max().item() * 2.0 ** -23)}), flush=True)


if __name__ == "__main__":
    main()
