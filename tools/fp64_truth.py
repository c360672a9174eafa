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


def attribute(arch="resnet50", h=1024, w=1024, seed=0):
    """WHERE the HIP trunk's distance from the float64 truth comes from, layer by layer: every conv unit of the trunk (one HIP
    launch: conv + folded BN (+ ReLU) (+ residual); the stem with its pool) is given the float64 truth of ITS input rounded to fp32
    — the same tensor for all three implementations — and evaluated in float64 (the unit's true result on that input), by
    torch-CPU fp32 (the reference's arithmetic) and by the HIP kernel the pipeline routes it to. One JSON line per unit: the
    local error of both in ulps (2^-23) of the unit's output range, max and rms; then the five units with the largest HIP
    excess. The float64 chain itself continues on unrounded float64 tensors."""
    import torch.nn.functional as F
    from maskrcnn_amd import modules, ops
    from maskrcnn_amd.config import InferenceConfig
    from oracle import oracle
    dev = torch.device("cuda:0")
    cfg = InferenceConfig(image_height=h, image_width=w, backbone=arch)
    sd = modules.synthetic_state_dict(arch, seed=0, bn_seed=1)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items() if k.startswith("fpn.")}
    g = torch.Generator().manual_seed(seed)
    img = (torch.randint(0, 256, (1, h, w, 3), generator=g).float() - torch.tensor(cfg.mean_pixel)).permute(0, 3, 1, 2).contiguous()
    bb = modules.FusedBackbone(sd, arch, dev, precision="f32")
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    back = lambda t: t.permute(0, 3, 1, 2).cpu()
    rows = []

    def unit(name, kind, mnk, ref64, got32, gothip):
        rng = ref64.abs().max().item()
        ulp = rng * 2.0 ** -23
        eo, eh = (got32.double() - ref64).abs(), (gothip.double() - ref64).abs()
        row = {"unit": name, "kernel": kind, "K": mnk, "max_abs_out": rng,
               "oracle_fp32_max_ulps": eo.max().item() / ulp, "hip_max_ulps": eh.max().item() / ulp,
               "oracle_fp32_rms_ulps": eo.pow(2).mean().sqrt().item() / ulp, "hip_rms_ulps": eh.pow(2).mean().sqrt().item() / ulp}
        row["hip_excess_rms_ulps"] = row["hip_rms_ulps"] - row["oracle_fp32_rms_ulps"]
        rows.append(row)
        print(json.dumps(row), flush=True)

    def conv_ref(x, p, bn, stride=1, pad3=False, relu=False, res=None, sdx=None):
        """conv (+BN) (+residual) (+ReLU) with the oracle's own functions on sdx (fp32 or fp64 parameters)."""
        y = oracle._conv(oracle.same_pad(x, 3, 1) if pad3 else x, sdx, p, stride=stride)
        if bn:
            y = oracle._bn(y, sdx, bn)
        if res is not None:
            y = y + res
        return F.relu(y) if relu else y

    def run_unit(name, fc, x64, p, bn, stride=1, pad3=False, relu=False, res64=None, res_div=1):
        """fc: the FusedConv the pipeline uses for this layer. Returns the float64 chain's output (from the unrounded input)."""
        x32 = x64.float()
        r32 = None if res64 is None else res64.float()
        up = (lambda t: F.interpolate(t, scale_factor=2)) if res_div == 2 else (lambda t: t)
        ref = conv_ref(x32.double(), p, bn, stride, pad3, relu, None if r32 is None else up(r32.double()), sd64)
        o32 = conv_ref(x32, p, bn, stride, pad3, relu, None if r32 is None else up(r32), sd)
        xd = nhwc(x32)
        hh, ww = xd.size(1), xd.size(2)
        kind = "direct"
        if pad3 and fc.w.takes_winograd4(hh, ww, 1, (1, 1, 1, 1), None, fc.relu, 1):
            yh, kind = fc(ops.nhwc_to_kblocked(xd)), "winograd4"
        elif pad3 and fc.takes_winograd(hh, ww):
            yh, kind = fc(xd), "winograd2"
        else:
            yh = fc(xd, residual=None if r32 is None else nhwc(r32), res_div=res_div)
        torch.cuda.synchronize()
        k = fc.w.shape[1] * fc.w.shape[2] * fc.algo_cin
        unit(name, kind, k, ref, o32, back(yh))
        return conv_ref(x64, p, bn, stride, pad3, relu, None if res64 is None else up(res64), sd64)

    with torch.no_grad():
        # stem + pool: one launch
        x64 = img.double()
        ref = oracle.stem(img.double(), sd64, "fpn.C1")
        o32 = oracle.stem(img, sd, "fpn.C1")
        st = bb.stem
        if modules.STEM_POOL:
            yh = ops.stem_pool_f32(img.to(dev), st.w.w, st.scale, st.shift, st.algo_cin)
        else:
            yh = ops.stem_conv(img.to(dev), st.w.w, st.scale, st.shift, True, st.algo_cin, nchw=True)
            yh = ops.maxpool(yh, 3, 2, ops.same_pad(yh.size(1), yh.size(2), 3, 2))
        unit("C1 stem+pool", "stem", 147, ref, o32, back(yh))
        x64 = ref
        cs = []
        for si, (name, blocks) in enumerate(zip(("C2", "C3", "C4", "C5"), bb.stages)):
            for bi, blk in enumerate(blocks):
                c1, c2, c3, cd = blk.convs
                pre = f"fpn.{name}.{bi}"
                stride = c1.stride
                res64 = x64
                if cd is not None:
                    res64 = run_unit(f"{name}.{bi}.downsample", cd, x64, pre + ".downsample.0", pre + ".downsample.1", stride)
                h1 = run_unit(f"{name}.{bi}.conv1", c1, x64, pre + ".conv1", pre + ".bn1", stride, relu=True)
                h2 = run_unit(f"{name}.{bi}.conv2", c2, h1, pre + ".conv2", pre + ".bn2", pad3=True, relu=True)
                x64 = run_unit(f"{name}.{bi}.conv3+res", c3, h2, pre + ".conv3", pre + ".bn3", relu=True, res64=res64)
            cs.append(x64)
        c2_, c3_, c4_, c5_ = cs
        p5 = run_unit("P5 lateral", bb.lateral[5], c5_, "fpn.P5_conv1", None)
        p4 = run_unit("P4 lateral+up", bb.lateral[4], c4_, "fpn.P4_conv1", None, res64=p5, res_div=2)
        p3 = run_unit("P3 lateral+up", bb.lateral[3], c3_, "fpn.P3_conv1", None, res64=p4, res_div=2)
        p2 = run_unit("P2 lateral+up", bb.lateral[2], c2_, "fpn.P2_conv1", None, res64=p3, res_div=2)
        for k, pk in ((5, p5), (4, p4), (3, p3), (2, p2)):
            run_unit(f"P{k} smoothing", bb.smooth[k], pk, f"fpn.P{k}_conv2.1", None, pad3=True)
    top = sorted(rows, key=lambda r: -r["hip_excess_rms_ulps"])[:5]
    by_kind = {}
    for r in rows:
        a = by_kind.setdefault(r["kernel"], {"units": 0, "hip_rms_ulps_mean": 0.0, "oracle_fp32_rms_ulps_mean": 0.0})
        a["units"] += 1
        a["hip_rms_ulps_mean"] += r["hip_rms_ulps"]
        a["oracle_fp32_rms_ulps_mean"] += r["oracle_fp32_rms_ulps"]
    for a in by_kind.values():
        a["hip_rms_ulps_mean"] /= a["units"]
        a["oracle_fp32_rms_ulps_mean"] /= a["units"]
    print(json.dumps({"summary": "attribution", "arch": arch, "image": [h, w], "units": len(rows),
                      "top5_by_hip_excess_rms_ulps": [{k: r[k] for k in ("unit", "kernel", "K", "hip_rms_ulps", "oracle_fp32_rms_ulps",
                                                                             "hip_max_ulps", "oracle_fp32_max_ulps")} for r in top],
                      "by_kernel": by_kind}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config5", type=int, default=0)
    ap.add_argument("--attribute", type=int, default=0, help="per-layer attribution of the HIP trunk's distance from the truth")
    args = ap.parse_args()
    if args.attribute:
        attribute()
        return
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
