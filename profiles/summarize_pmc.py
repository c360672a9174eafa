#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of tools/profile_step.py (one counter group per pass, --kernel-trace,
--output-format csv) into the per-kernel JSON files committed under profiles/.

    summarize_pmc.py mfma    <counter_collection.csv> <meta.json>                  > profiles/rNN_mfma_util.json
    summarize_pmc.py traffic <fetch counter_collection.csv> <write ...csv> <meta>  > profiles/rNN_hbm_traffic.json

Every predict() of profile_step.py has the workload's batch size, so per-step figures are totals / predict calls.

mfma:    SQ_INSTS_MFMA (wave-level MFMA instructions, all SEs), GRBM_GUI_ACTIVE (busy cycles summed over the 8 XCDs),
         SQ_BUSY_CYCLES. MFMA utilisation of a kernel = insts x (issue cycles of its MFMA per SIMD) / (1024 SIMDs x kernel
         cycles), kernel cycles = GRBM_GUI_ACTIVE / 8. Issue cycles (MI355X_MICROARCH.md § cycle constants):
         v_mfma_f32_32x32x2_f32 = 64; v_mfma_f32_32x32x16_f16 = 32 (8 of them hold the vector issue port).
traffic: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
         streaming read (MI355X_MICROARCH.md § HBM) → doubled. WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import json
import re
import sys
from collections import defaultdict

SIMDS = 1024


def short(name):
    """'(anonymous namespace)::conv_igemm_f32<128, 64, 2, 2, 16, 2, 1>(...)' → 'conv_igemm_f32<128,64,2,2,16,2,1>'"""
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace(", ", ",")


def base(name):
    return re.sub(r"<.*$", "", short(name))


def mfma_cycles(kernel):
    # issue cycles per SIMD: v_mfma_f32_16x16x32_f16 (conv_f16p) 16, v_mfma_f32_32x32x16_f16 32, v_mfma_f32_32x32x2_f32 64
    return 16 if "conv_f16p" in kernel else 32 if "f16" in kernel else 64


def read(path):
    per = defaultdict(lambda: defaultdict(float))   # kernel → counter → sum
    disp = defaultdict(set)
    ns = defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in disp[k]:
            disp[k].add(r["Dispatch_Id"])
            ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return per, {k: len(v) for k, v in disp.items()}, ns


def group(d):
    """instantiation-level dict → also summed per base kernel name"""
    out = defaultdict(lambda: defaultdict(float))
    for k, v in d.items():
        for c, x in v.items():
            out[base(k)][c] += x
    return out


def cmd_mfma(path, meta):
    per, n, ns = read(path)
    steps = meta["predict_calls"]
    rows = {}
    agg = defaultdict(lambda: defaultdict(float))
    for k in per:
        c = per[k]
        if c.get("SQ_INSTS_MFMA", 0) <= 0:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        util = c["SQ_INSTS_MFMA"] * mfma_cycles(k) / (SIMDS * cyc) if cyc else None
        rows[k] = {"launches_per_step": n[k] / steps, "mfma_insts_per_step": c["SQ_INSTS_MFMA"] / steps,
                   "kernel_cycles_per_step": cyc / steps, "mfma_util": round(util, 4),
                   "ms_per_step_in_this_pmc_pass": round(ns[k] / steps / 1e6, 3),
                   "clock_ghz_in_this_pmc_pass": round(cyc / ns[k], 3) if ns[k] else None}
        b = base(k)
        agg[b]["insts_x_cycles"] += c["SQ_INSTS_MFMA"] * mfma_cycles(k)
        agg[b]["cycles"] += cyc
        agg[b]["launches"] += n[k]
    by_kernel = {b: {"launches_per_step": v["launches"] / steps, "kernel_cycles_per_step": v["cycles"] / steps,
                     "mfma_util": round(v["insts_x_cycles"] / (SIMDS * v["cycles"]), 4)} for b, v in agg.items()}
    tot_i = sum(v["insts_x_cycles"] for v in agg.values())
    tot_c = sum(v["cycles"] for v in agg.values())
    return dict(meta, definition="mfma_util = SQ_INSTS_MFMA x issue cycles (64: fp32 32x32x2; 32: fp16 32x32x16; 16: fp16 16x16x32) / (1024 SIMDs x "
                                 "GRBM_GUI_ACTIVE / 8); launches summed per kernel over one step",
                all_mfma_kernels={"mfma_util": round(tot_i / (SIMDS * tot_c), 4), "kernel_cycles_per_step": tot_c / steps},
                by_kernel=dict(sorted(by_kernel.items(), key=lambda kv: -kv[1]["kernel_cycles_per_step"])),
                by_instantiation=dict(sorted(rows.items(), key=lambda kv: -kv[1]["kernel_cycles_per_step"])))


def cmd_traffic(fetch_path, write_path, meta):
    fper, fn, _ = read(fetch_path)
    wper, wn, _ = read(write_path)
    steps = meta["predict_calls"]
    f, w = group(fper), group(wper)
    launches = defaultdict(float)
    for k, v in fn.items():
        launches[base(k)] += v
    out = {}
    for b in sorted(set(f) | set(w)):
        fb = f[b].get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / steps
        wb = w[b].get("WRITE_SIZE", 0.0) * 1024.0 / steps
        out[b] = {"launches_per_step": launches[b] / steps, "fetch_bytes_per_step_x2_corrected": fb,
                  "write_bytes_per_step": wb, "hbm_bytes_per_step": fb + wb}
        if out[b]["launches_per_step"] == int(out[b]["launches_per_step"]):
            out[b]["launches_per_step"] = int(out[b]["launches_per_step"])
    conv = [b for b in out if any(t in b for t in ("conv", "stem", "wino", "kblock", "bottleneck"))]
    total = sum(v["hbm_bytes_per_step"] for v in out.values())
    return dict(meta, corrections="KiB → bytes; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read)",
                hbm_bytes_per_step=total,
                conv_path_hbm_bytes_per_step=sum(out[b]["hbm_bytes_per_step"] for b in conv),
                per_kernel=dict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])))


def cmd_counters(paths, meta):
    """Any set of counter passes over tools/profile_step.py -> per kernel instantiation, per step: the raw counter sums plus the
    ratios the design discussion uses (L2 hit rate, fabric read bytes, LDS bank-conflict share)."""
    steps = meta["predict_calls"]
    rows = defaultdict(dict)
    launches = {}
    for path in paths:
        per, n, ns = read(path)
        for k, c in per.items():
            launches[k] = n[k] / steps
            for name, v in c.items():
                rows[k][name] = v / steps
    out = {}
    for k, c in rows.items():
        if not any(t in k for t in ("conv", "stem", "wino", "bottleneck", "roi_align", "crop", "maxpool")):
            continue
        e = {"launches_per_step": launches[k]}
        e.update({name: round(v, 1) for name, v in sorted(c.items())})
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if "TCC_EA0_RDREQ_sum" in c:   # 128-byte read requests from the L2s to the fabric (64-byte ones counted apart when asked for)
            e["fabric_read_GB_at_128B_per_request"] = round(c["TCC_EA0_RDREQ_sum"] * 128 / 1e9, 3)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            e["lds_bank_conflict_share_of_lds_active"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        out[k] = e
    key = lambda kv: -kv[1].get("TCC_EA0_RDREQ_sum", kv[1].get("GRBM_GUI_ACTIVE", 0))
    return dict(meta, definition="rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over tools/profile_step.py: "
                                 "sums over all launches of a kernel instantiation in one step; TCC_* are summed over the L2 channels of "
                                 "all eight XCDs, SQ_* over all SIMDs",
                by_instantiation=dict(sorted(out.items(), key=key)))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "mfma":
        res = cmd_mfma(sys.argv[2], json.load(open(sys.argv[3])))
    elif cmd == "traffic":
        res = cmd_traffic(sys.argv[2], sys.argv[3], json.load(open(sys.argv[4])))
    elif cmd == "counters":   # counters <meta.json> <counter_collection.csv> ...
        res = cmd_counters(sys.argv[3:], json.load(open(sys.argv[2])))
    else:
        raise SystemExit(__doc__)
    print(json.dumps(res, indent=1))
