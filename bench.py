#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): images/sec of the Mask R-CNN inference hot path at 1024x1024 with
1000 proposals per image on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]          (N=1 by default)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of synthetic images, per GPU:
trunk (C1-C5+FPN) → RPN → proposal decode + NMS → RoIAlign 7x7 → classifier head → per-class NMS →
RoIAlign 14x14 → mask head, followed — for N>1 — by the one RCCL all-gather of the
fixed-shape detections. Workload at every N: BASELINE.json configs[2] per GPU (ResNet-50-FPN, batch 8,
synthetic 1024x1024) → weak scaling; N=8 is configs[3] (batch 64 sharded over 8 GPUs). Inputs are resident
in HBM before the timed region. fp32 throughout (exact-fp32 MFMA).

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  `roofline`      the dominant kernel (largest summed duration in a step: the direct implicit-GEMM kernel conv_igemm_f32 or
                  the F(4x4) Winograd kernel conv3x3_wino4_f32): the multiply-adds it EXECUTES (2*M*N*K, / 2.25 for Winograd
                  F(2x2,3x3), / 4 for F(4x4,3x3)) / its summed launch durations — HIP events on the launch stream around every conv
                  launch, in an instrumented pass right after the timed region — against the fp32-MFMA peak, so
                  `frac` <= 1; the convolution-equivalent (algorithmic) rate and the whole conv path sit beside it;
                  FLOPs / bytes are those of the work that RAN: the classifier's row-group GEMMs skip 128-row tiles of empty RoI slots
                  and are booked on the rows of the tiles that executed (`heads_rows`); every `by_kernel` row carries
                  `executed_gflop_per_step`, so frac of any set of families = sum(GFLOP) / sum(ms) / peak from the line itself;
  `roofline_ops`  RoIAlign (BASELINE config 2 shape + the pipeline's pyramid call; HBM-bound) and NMS 8 x 1000
                  (latency-bound), measured in the same run after the timed region;
  `alt_configs`   never `value`: the headline step with SURVEY 8(d)'s 1000 VALID proposals per image injected behind the proposal
                  stage; MaskRCNNInference.detect() (uint8 images in, full-size masks out) beside predict() on the same batch;
                  BASELINE configs[4]'s geometry (R101-FPN, 832x1344, fp16 MFMA path) on this GPU;
  `per_rank_ms_per_step`  slowest / fastest rank's own K steps per repetition (a straggler shows the first time N > 1 runs);
  `cpu_baseline`  the CPU oracle's predict() on a bounded sample of the same workload, timed on this host's cores.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (spec)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured float4 copy)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz


TRAFFIC_PROFILE = "r05_hbm_traffic.json"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_sha16():
    """sha256 over maskrcnn_amd/csrc/*.{hip,hpp} (sorted by name): ties a committed PMC profile to the kernels it profiled."""
    import hashlib
    d = os.path.join(ROOT, "maskrcnn_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def calibrate_heads_(sd, make_net, images, windows):
    """Random weights saturate the RPN/classifier softmaxes and blow up box deltas, which makes every
    proposal degenerate. Rescale the four head layers (weights only, deterministic) until scores are
    spread and boxes sane, so the proposal/detection stages see realistic, diverse boxes."""
    net = None
    for _ in range(8):
        net = make_net(sd)
        _, mid = net.predict(images, windows, with_masks=False, return_intermediates=True)
        sc = mid["rpn_scores"].double().clamp(1e-7, 1 - 1e-7)
        lsd = torch.log(sc / (1 - sc)).std().item()
        dstd, cstd, bstd = mid["rpn_deltas"].std().item(), mid["logits"].std().item(), mid["bbox"].std().item()
        done = True
        for key, cur, target, limit in (("rpn.conv_class.weight", lsd, 1.0, 2.0),
                                        ("rpn.conv_bbox.weight", dstd, 0.5, 1.0),
                                        ("classifier.linear_class.weight", cstd, 2.0, 3.0),
                                        ("classifier.linear_bbox.weight", bstd, 0.5, 1.0)):
            if cur > limit:
                sd[key] = sd[key] * (target / cur)
                done = False
        if done:
            break
    return net


def measure_traffic_in_run(args, H, W):
    """HBM / fabric bytes per kernel of ONE step of this workload, measured in this run: two rocprofv3 counter passes
    (`--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`: separate passes, `--kernel-trace` only, as MI355X_MICROARCH.md prescribes) over
    `tools/profile_step.py` — the same step, weights and inputs — as CHILD processes, summarised by profiles/summarize_pmc.py
    (KiB -> bytes, FETCH_SIZE doubled on gfx950). Must run BEFORE this process touches the GPU (a child is fork + exec).
    Returns (summary dict | None, reason)."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler"
    tmp = tempfile.mkdtemp(prefix="mrcnn_pmc_", dir="/tmp")
    found = {}
    try:
        for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("mfma", "SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE")):
            d = os.path.join(tmp, name)
            os.makedirs(d)
            cmd = [exe, "--pmc", *ctr.split(), "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "profile_step.py"), "--steps", "1", "--batch", str(args.batch),
                   "--arch", args.arch, "--height", str(H), "--width", str(W), "--proposals", str(args.proposals),
                   "--precision", args.precision, "--meta", os.path.join(d, "meta.json")]
            t0 = time.perf_counter()
            with open(os.path.join(d, "log.txt"), "w") as lf:
                # its own session: on a timeout the WHOLE group goes (rocprofv3 is a wrapper; a profiled python left behind
                # would share the GPU with the timed region)
                proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=lf, stderr=lf,
                                        start_new_session=True)
                try:
                    rc = proc.wait(timeout=args.traffic_timeout)
                except subprocess.TimeoutExpired:
                    import signal
                    with contextlib.suppress(ProcessLookupError):
                        os.killpg(proc.pid, signal.SIGKILL)
                    proc.wait()
                    raise
            if rc != 0:
                with open(os.path.join(d, "log.txt")) as lf:
                    tail = lf.read()[-400:]
                return None, f"rocprofv3 --pmc {ctr} exited {rc}: {tail!r}"
            csvs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if not csvs or not os.path.exists(os.path.join(d, "meta.json")):
                return None, f"rocprofv3 --pmc {ctr}: no counter_collection.csv"
            found[name] = csvs[0]
            log(f"[bench] traffic pass {ctr}: {time.perf_counter() - t0:.1f} s")
        spec = importlib.util.spec_from_file_location("_summarize_pmc", os.path.join(ROOT, "profiles", "summarize_pmc.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        with open(os.path.join(tmp, "fetch", "meta.json")) as fh:
            meta = json.load(fh)
        out = mod.cmd_traffic(found["fetch"], found["write"], meta)
        with open(os.path.join(tmp, "mfma", "meta.json")) as fh:
            out["mfma"] = mod.cmd_mfma(found["mfma"], json.load(fh))
        return out, "measured"
    except subprocess.TimeoutExpired:
        return None, f"a counter pass exceeded {args.traffic_timeout} s"
    except Exception as e:  # a profiler problem never costs the GPU number
        return None, f"failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def conv_roofline(prof, args, H, W, modules, measured=None, measured_reason=None):
    """prof: ops.CONV_PROFILE rows (start event, end event, algorithmic FLOPs = 2*M*N*K of the convolution, (M,N,K),
    algorithmic bytes = every operand/result tensor once, kernel tag) of `args.roofline_steps` steps."""
    steps = args.roofline_steps
    times = [r[0].elapsed_time(r[1]) for r in prof]   # ms
    tag = [r[5] if len(r) > 5 else "direct" for r in prof]
    # multiply-adds the MFMA pipe really performs: Winograd F(2x2,3x3) needs 16 products per 2x2 outputs and channel
    # pair instead of 36
    executed = [r[6] if len(r) > 6 else (r[2] / 4.0 if t == "winograd4" else r[2] / 2.25 if t.startswith("winograd") else r[2])
                for r, t in zip(prof, tag)]
    peak = F32_MFMA_PEAK_TFLOPS if args.precision == "f32" else F16_MFMA_PEAK_TFLOPS

    def agg(sel):
        ms = sum(t for t, k in zip(times, sel) if k)
        n = sum(sel)
        if n == 0 or ms <= 0:
            return None
        ex = sum(e for e, k in zip(executed, sel) if k)
        al = sum(r[2] for r, k in zip(prof, sel) if k)
        return {"launches_per_step": n // steps, "ms_per_step": round(ms / steps, 3),
                "avg_launch_us": round(ms / n * 1e3, 2),
                # frac of any set of families = sum(executed_gflop_per_step) / sum(ms_per_step) / peak: recomputable from here
                "executed_gflop_per_step": round(ex / steps / 1e9, 3),
                "algorithmic_gflop_per_step": round(al / steps / 1e9, 3),
                "executed_tflops": round(ex / (ms * 1e-3) / 1e12, 2),
                "executed_frac": round(ex / (ms * 1e-3) / 1e12 / peak, 4),
                "algorithmic_tflops": round(al / (ms * 1e-3) / 1e12, 2),
                "algorithmic_bytes_per_step": int(sum(r[4] for r, k in zip(prof, sel) if k) / steps)}

    whole = agg([True] * len(prof))
    by_tag = {t: agg([x == t for x in tag]) for t in sorted(set(tag))}
    ranked = sorted(by_tag, key=lambda t: -by_tag[t]["ms_per_step"])
    # The dominant kernel FAMILY — and every family within 5 % of its summed time: with two co-dominant families (the direct
    # implicit GEMM and the F(4x4) Winograd kernel are 1-2 % apart in this step) `frac` is the fraction of BOTH together, and
    # each is listed under `co_dominant`, so the line cannot quote the better half
    dom_tags = [t for t in ranked if by_tag[t]["ms_per_step"] >= 0.95 * by_tag[ranked[0]]["ms_per_step"]]
    dominant = ranked[0]
    dom = agg([x in dom_tags for x in tag])
    kernel_of = {"winograd": "conv3x3_wino8_f32", "winograd_spatial": "conv3x3_wino8s_f32", "winograd4": "conv3x3_wino4_f32",
                 "direct": "conv_igemm_f32",
                 "stem": "stem7x7_s2_f32",
                 "f16": "conv_igemm_f16", "f16p": "conv_f16p", "f16blk": "bottleneck_c2_f16", "f16tail": "mask_tail_f16",
                 "rpn_fused": "conv_igemm_f32<heads>", "bottleneck": "bottleneck_fused_f32"}
    if args.dump_conv:
        per = len(prof) // steps
        rows = []
        for i in range(per):
            t = sum(times[i + r * per] for r in range(steps)) / steps
            f, mnk = prof[i][2], prof[i][3]
            rows.append({"i": i, "M": mnk[0], "N": mnk[1], "K": mnk[2], "ms": round(t, 4), "kernel": tag[i],
                         "tflops": round(f / (t * 1e-3) / 1e12, 1),
                         "executed_tflops": round(executed[i] / (t * 1e-3) / 1e12, 1), "gflop": round(f / 1e9, 2),
                         "algorithmic_MB": round(prof[i][4] / 1e6, 1),
                         "algorithmic_GBps": round(prof[i][4] / (t * 1e-3) / 1e9, 0)})
            if len(prof[i]) > 7 and isinstance(prof[i][7], dict):
                rows[-1].update(prof[i][7])
        with open(args.dump_conv, "w") as fh:
            json.dump(rows, fh, indent=0)
    # HBM traffic: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, summarised per kernel by
    # profiles/summarize_pmc.py into a file that records the mode it was taken in; anything else → null
    traffic, traffic_src, in_run, conv_traffic, counters = None, None, False, None, None
    # the kernels behind a tag: the direct implicit GEMM is two kernels since the streaming 1x1 kernel (same arithmetic)
    def members_of(t):
        return {"direct": ("conv_igemm_f32", "conv_pw_stream_f32")}.get(t, (kernel_of.get(t, t),))

    members = tuple(m for t in dom_tags for m in members_of(t))

    def summed(per_kernel, field, mem=None):
        rows = [per_kernel[m] for m in (mem or members) if m in per_kernel]
        return (sum(r[field] for r in rows), sum(r["launches_per_step"] for r in rows)) if rows else (None, None)

    if measured is not None:   # this run's own counter passes (measure_traffic_in_run)
        tb, tl = summed(measured.get("per_kernel", {}), "hbm_bytes_per_step")
        if tb is not None and tl == dom["launches_per_step"]:
            traffic, traffic_src, in_run = tb, "rocprofv3 --pmc passes run by this bench.py invocation", True
            conv_traffic = measured.get("conv_path_hbm_bytes_per_step")
            mf = measured.get("mfma", {})
            rows = [mf.get("by_kernel", {}).get(m) for m in members]
            rows = [r for r in rows if r]
            if rows:
                cyc = sum(r["kernel_cycles_per_step"] for r in rows)
                counters = {"mfma_util": round(sum(r["mfma_util"] * r["kernel_cycles_per_step"] for r in rows) / cyc, 4),
                            "all_mfma_kernels_mfma_util": mf["all_mfma_kernels"]["mfma_util"],
                            "by_kernel": {b: v["mfma_util"] for b, v in mf["by_kernel"].items()},
                            "definition": "SQ_INSTS_MFMA x 64 issue cycles (v_mfma_f32_32x32x2_f32) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), "
                                          "one rocprofv3 --pmc pass over tools/profile_step.py started by this bench.py run; a counter "
                                          "pass serialises dispatches and runs at its own clock, so this is the fraction of the kernels' "
                                          "OWN cycles the MFMA pipe was issuing, to set beside the time-derived `frac`"}
        else:
            measured_reason = f"launch count of the counter pass ({tl}) != this run's ({dom['launches_per_step']})"
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
    if traffic is None and os.path.exists(tpath):
        try:
            with open(tpath) as fh:
                tj = json.load(fh)
            same = (tj.get("precision") == args.precision and tj.get("batch") == args.batch
                    and tj.get("image") == [H, W] and tj.get("arch") == args.arch
                    and tj.get("proposals") == args.proposals and tj.get("winograd") == bool(modules.WINOGRAD)
                    and tj.get("stem_kernel") == bool(modules.STEM_KERNEL)
                    and tj.get("fused_bottleneck") == bool(getattr(modules, "FUSED_BOTTLENECK", False))
                    and tj.get("rpn_fused_heads") == bool(getattr(modules, "RPN_FUSED_HEADS", False))
                    and tj.get("winograd4") == bool(getattr(modules, "WINOGRAD4", False))
                    and tj.get("winograd4_trunk") == bool(getattr(modules, "WINOGRAD4_TRUNK", False))
                    # the byte counts are only this run's if the kernels are the ones that were profiled
                    and tj.get("kernel_source_sha16") == kernel_source_sha16())
            tb, tl = summed(tj.get("per_kernel", {}), "hbm_bytes_per_step")
            if same and tb is not None and tl == dom["launches_per_step"]:
                traffic = tb
                traffic_src = "profiles/" + TRAFFIC_PROFILE
        except (OSError, ValueError, KeyError):
            traffic = None
    # the heads' GEMMs over [image][RoI slot] rows: rows of the tiles that ran / all slots / valid rows (ops.rows_executed)
    rows_rows = [r[7] for r in prof if len(r) > 7 and isinstance(r[7], dict) and "rows_slots" in r[7]]
    heads_rows = None
    if rows_rows:
        per = max(1, len(rows_rows) // steps)
        heads_rows = {"launches_per_step": per,
                      "rows_slots": rows_rows[0]["rows_slots"], "rows_executed": rows_rows[0]["rows_executed"],
                      "rows_valid": rows_rows[0]["rows_valid"],
                      "note": "first row-group GEMM of a step (the classifier's K = 12544 layer); FLOPs / bytes of these launches are "
                              "booked on rows_executed (MFMA work that ran: whole 128-row tiles with at least one valid RoI), their "
                              "algorithmic FLOPs on rows_valid; tiles of empty RoI slots return at once (csrc/conv.hip)"}
    co = []
    for t in dom_tags:
        row = dict(family=t, kernel=" + ".join(members_of(t)), **by_tag[t])
        if measured is not None:
            tb, tl = summed(measured.get("per_kernel", {}), "hbm_bytes_per_step", members_of(t))
            if tb is not None and tl == by_tag[t]["launches_per_step"]:
                row["traffic"] = tb
                row["traffic_over_algorithmic"] = round(tb / max(1, by_tag[t]["algorithmic_bytes_per_step"]), 3)
        co.append(row)
    return {"bound": "mfma", "kernel": " + ".join(members), "co_dominant": co,
            "co_dominant_note": "every kernel family whose summed launch time is within 5 % of the largest; achieved / frac / "
                                "traffic / algorithmic_bytes of this object are over ALL of them together",
            "achieved": dom["executed_tflops"], "peak": peak, "unit": "TFLOP/s", "frac": dom["executed_frac"],
            "definition": "achieved = multiply-adds the kernel executes x 2 (Winograd F(2x2,3x3): 2*M*N*K / 2.25, F(4x4,3x3): / 4) / "
                          "summed launch durations of that kernel in one step (HIP events on the launch stream); "
                          "frac = achieved / fp32-MFMA peak",
            "launches_per_step": dom["launches_per_step"], "ms_per_step": dom["ms_per_step"],
            "avg_launch_us": dom["avg_launch_us"],
            "algorithmic_tflops": dom["algorithmic_tflops"],
            "algorithmic_speedup": round(dom["algorithmic_tflops"] / dom["executed_tflops"], 3),
            "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_in_this_run": in_run,
            "traffic_conv_path": conv_traffic, "mfma_counters": counters,
            "traffic_note": ("fabric bytes of this kernel's launches of one step: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + "
                             "--pmc WRITE_SIZE, two separate counter passes over tools/profile_step.py (the same step, weights "
                             "and inputs) started by this bench.py run before its timed region"
                             if in_run else
                             "fabric bytes of this kernel's launches of one step from the committed rocprofv3 passes (--pmc "
                             "FETCH_SIZE x2 + WRITE_SIZE, separate passes of tools/profile_step.py): NOT measured in this run"
                             f" ({measured_reason}); null unless the profile's mode flags, launch count and csrc/ source hash "
                             "equal this run's"),
            "algorithmic_bytes": dom["algorithmic_bytes_per_step"],
            "heads_rows": heads_rows,
            "conv_path": dict(whole, conv_gflop_per_image=round(sum(r[2] for r in prof) / steps / args.batch / 1e9, 1)),
            "by_kernel": by_tag}


def op_rooflines(dev, ops):
    """The two non-conv hot ops, HIP-event timed on the launch stream after the timed region:
    RoIAlign is HBM-bound (compulsory bytes = output + every touched map once + boxes), NMS latency-bound."""
    def timeit(fn, iters=20, reps=3):
        """Device time per call, us: `iters` calls captured in a hipGraph and replayed (the Python + ctypes cost of a call,
        ~20 us, would otherwise bound a 20-30 us kernel); best of `reps` replays, HIP events on the launch stream."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        del graph
        return best

    out = []
    g = torch.Generator().manual_seed(1234)
    # BASELINE configs[1]: 256 RoIs x 256 ch x 14x14 on P2 of one 1024^2 image, the NCHW drop-in entry point
    fm = torch.randn(1, 256, 256, 256, generator=g).to(dev)
    c = torch.rand(256, 2, generator=g)
    hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
    boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
    ind = torch.zeros(256, dtype=torch.int32, device=dev)
    algo = 256 * 256 * 14 * 14 * 4 + fm.numel() * 4 + 256 * 20
    us = timeit(lambda: ops.crop(fm, boxes, ind, 0.0, 14, 14))
    # context, not a target: a plain device-to-device copy moving the same number of bytes (half read, half written), timed
    # the same way — what this box's memory system gives a kernel of this size with no gather in it
    src = torch.empty(algo // 8, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    copy_us = timeit(lambda: dst.copy_(src))
    del src, dst
    out.append({"op": "crop_forward_nchw (configs[1]: 256 RoIs x 256 ch x 14x14 on P2)", "bound": "hbm",
                "kernel": "crop_forward_nchw_staged", "us": round(us, 2), "algorithmic_bytes": algo,
                "achieved": round(algo / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(algo / us / 1e3 / HBM_PEAK_GBS, 4),
                "device_copy_of_the_same_bytes_us": round(copy_us, 2),
                "frac_of_that_copy_rate": round(copy_us / us, 4),
                "timing": "device time per call, 20 calls replayed from a hipGraph (includes the ~1.5 us kernel boundary)"})
    del fm
    # the same call on the P3..P5 map sizes
    for hl in (128, 64, 32):
        fm = torch.randn(1, 256, hl, hl, generator=g).to(dev)
        algo_l = 256 * 256 * 14 * 14 * 4 + fm.numel() * 4 + 256 * 20
        us = timeit(lambda: ops.crop(fm, boxes, ind, 0.0, 14, 14))
        out.append({"op": f"crop_forward_nchw (256 RoIs x 256 ch x 14x14 on a {hl}x{hl} map)", "bound": "hbm",
                    "us": round(us, 2), "algorithmic_bytes": algo_l, "achieved": round(algo_l / us / 1e3, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo_l / us / 1e3 / HBM_PEAK_GBS, 4)})
        del fm
    # the pipeline's classifier-head call: 8 images x 1000 RoIs x 7x7 over the four NHWC levels, one launch
    fms = [torch.randn(8, 1024 // s, 1024 // s, 256, generator=g).to(dev) for s in (4, 8, 16, 32)]
    c = torch.rand(8000, 2, generator=g)
    hw = torch.exp(torch.rand(8000, 2, generator=g) * 3.4 - 3.9)
    rois = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
    us = timeit(lambda: ops.roi_align_pyramid(fms, rois, 7, 1024.0 * 1024.0, rois_per_image=1000))
    algo = 8000 * 49 * 256 * 4 + sum(f.numel() for f in fms) * 4 + 8000 * 16
    out.append({"op": "roi_align_pyramid_nhwc (8 x 1000 RoIs x 7x7, P2..P5)", "bound": "hbm", "us": round(us, 2),
                "algorithmic_bytes": algo, "achieved": round(algo / us / 1e3, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(algo / us / 1e3 / HBM_PEAK_GBS, 4)})
    del fms
    # NMS, 8 segments x 1000 boxes, threshold 0.7 (the proposal stage of one step)
    d8 = torch.cat([torch.rand(8, 1000, 2, generator=g) * 900, torch.zeros(8, 1000, 2),
                    torch.rand(8, 1000, 1, generator=g)], 2)
    d8[..., 2:4] = d8[..., :2] + torch.exp(torch.rand(8, 1000, 2, generator=g) * 2.5 + 2.0)
    d8 = d8.to(dev)
    us = timeit(lambda: ops.nms_batched(d8, 0.7))
    _, cnt = ops.nms_batched(d8, 0.7)
    out.append({"op": "nms_batched (8 segments x 1000 boxes, thr 0.7)", "bound": "latency", "us": round(us, 2),
                "boxes_per_s": round(8000 / us * 1e6), "kept_mean": float(cnt.float().mean()),
                "note": "moves 160 KB and does <= 4M IoU tests: neither HBM nor MFMA bound; 3 launches "
                        "(sort, pair mask over the chip, scan), kernel boundary ~1.5 us each"})
    return out


def cpu_baseline(sd, cfg, n_images, seed):
    """Oracle predict() (torch-CPU fp32 convs + the C restatement of nms/crop) on `n_images` images of the
    same workload, one at a time like the reference (batch 1, model.py:1321)."""
    from oracle import oracle
    ocfg = oracle.Cfg(cfg.image_height, cfg.image_width, PRE_NMS_LIMIT=cfg.pre_nms_limit,
                      RPN_NMS_MAX_ROIS_NUM=cfg.proposal_count,
                      DETECTION_MAX_INSTANCES=cfg.detection_max_instances)
    anchors = oracle.anchors_for(ocfg)
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor(cfg.mean_pixel)
    window = (0.0, 0.0, float(cfg.image_height), float(cfg.image_width))
    times = []
    with torch.no_grad():
        for i in range(n_images + 1):  # first image = warm-up (oneDNN primitive creation), not counted
            img = (torch.randint(0, 256, (1, cfg.image_height, cfg.image_width, 3), generator=g).float() - mean)
            img = img.permute(0, 3, 1, 2).contiguous()
            t0 = time.perf_counter()
            oracle.predict(img, window, sd, ocfg, cfg.backbone, anchors)
            times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / max(1, len(times) - 1)
    return {"value": 1.0 / t, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "seconds_per_image": [round(x, 3) for x in times[1:]], "warmup_image_seconds": round(times[0], 3),
            "sample": f"{n_images} image(s) of the same workload, batch 1, oracle.predict "
                      f"(torch-CPU fp32 convs + C nms/crop), {t:.2f} s/image, host cpus={os.cpu_count()}"}


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N-rank job as a CHILD process — `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same
    arguments>` — relay rank 0's JSON line on stdout and return the child's exit code. This process never touches the GPU (a
    process that has must not be replaced or forked into a launcher); `torch.cuda.device_count()` does not initialise it.
    `--dry-run-launch` prints the child command as one JSON line instead of running it."""
    import socket
    import subprocess
    with socket.socket() as s:   # a free port: two benches on one node must not collide on 29500
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in argv if a != "--dry-run-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MASTER_PORT", None)
    if args.dry_run_launch:
        print(json.dumps({"launch": cmd, "env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}}), flush=True)
        return 0
    rehearsal = os.environ.get("MRCNN_DIST_REHEARSAL") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not rehearsal:
        log(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s)")
        return 2
    log("[bench] launching: " + " ".join(cmd))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        for ln in proc.stdout:   # rank 0's line goes to our stdout, anything else a rank printed to stderr
            tgt = sys.stdout if ln.lstrip().startswith('{"metric"') else sys.stderr
            tgt.write(ln)
            tgt.flush()
        return proc.wait()
    except BaseException:
        import signal
        with contextlib.suppress(ProcessLookupError):
            os.killpg(proc.pid, signal.SIGTERM)
        proc.wait()
        raise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="with --gpus N > 1 outside torchrun: print the child command bench.py would start, and exit")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3, help="timed repetitions of --steps steps; value = the median one")
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (configs[2]: 8)")
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--size", type=int, default=1024, help="square image side (BASELINE metric: 1024)")
    ap.add_argument("--height", type=int, default=None, help="non-square images (config 5: 832 x 1344)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--proposals", type=int, default=1000)
    ap.add_argument("--cpu-images", type=int, default=2, help="CPU baseline sample size (0 = skip)")
    ap.add_argument("--roofline-steps", type=int, default=2)
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="also report (never as `value`) the throughput with this many batches in flight on alternating streams; "
                         "1 = skip")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3", "f16"],
                    help="contraction mode of the headline number (default: exact-fp32 MFMA)")
    ap.add_argument("--alt-precision", default="f16x3,f32+f16x3",
                    help="comma-separated contraction modes also timed after the headline (none | f32 | f16x3 | f16 | f32+f16x3); "
                         "the first is reported under alt_precision, all of them under alt_precisions")
    ap.add_argument("--dump-conv", default=None, help="write per-launch conv (M,N,K,ms,TFLOP/s) JSON here")
    ap.add_argument("--measure-traffic", type=int, default=1,
                    help="N=1 only: measure roofline.traffic in this run (two rocprofv3 --pmc child passes of tools/profile_step.py, "
                         "about a minute); 0 = report the committed profile's figure instead")
    ap.add_argument("--traffic-timeout", type=int, default=120, help="seconds allowed per counter pass")
    ap.add_argument("--alt-injected", type=int, default=1,
                    help="also time the headline step with SURVEY 8(d)'s injected proposals (every RoI slot valid), N=1 only")
    ap.add_argument("--alt-detect", type=int, default=1,
                    help="also time MaskRCNNInference.detect() (uint8 images in, full-size masks out) on configs[0]-sized images, N=1 only")
    ap.add_argument("--alt-config5", type=int, default=1,
                    help="also time BASELINE configs[4]'s geometry (R101-FPN, 832x1344, fp16 MFMA path) on this GPU, N=1 only")
    args = ap.parse_args()

    # `python bench.py --gpus N`, N > 1, not under torchrun: become the launcher (before anything here touches the GPU)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    # the counter passes are child processes: started before this process initialises the GPU
    measured, measured_reason = None, "--measure-traffic 0"
    if args.measure_traffic and int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and args.roofline_steps > 0:
        measured, measured_reason = measure_traffic_in_run(args, args.height or args.size, args.width or args.size)
        if measured is None:
            log(f"[bench] roofline.traffic not measured in this run: {measured_reason}")

    from maskrcnn_amd import dist as mdist
    rank, local, world = mdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch with "
                         f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    local = mdist.device_index(local)   # LOCAL_RANK, or cuda:0 for every rank of a one-GPU rehearsal (MRCNN_DIST_REHEARSAL=1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from maskrcnn_amd import modules, ops
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import MaskRCNNInference

    H, W = args.height or args.size, args.width or args.size
    cfg = InferenceConfig(image_height=H, image_width=W, backbone=args.arch,
                          pre_nms_limit=args.proposals, proposal_count=args.proposals)
    sd = modules.synthetic_state_dict(args.arch, seed=0, bn_seed=1)
    g = torch.Generator().manual_seed(rank)  # seed 0 on rank 0; each rank owns its own shard of the batch
    mean = torch.tensor(cfg.mean_pixel)
    images = (torch.randint(0, 256, (args.batch, H, W, 3), generator=g).float() - mean)
    images = images.permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0.0, 0.0, float(H), float(W)]] * args.batch, device=dev)

    make_net = lambda s, prec=args.precision: MaskRCNNInference(s, cfg, dev, precision=prec)
    # the head calibration uses ONE fixed batch on every rank, so all ranks end up with identical (replicated) weights;
    # it has the workload's batch size so that every launch of the run (and of a rocprofv3 trace of it) has the
    # workload's shape
    gc = torch.Generator().manual_seed(999)
    cal = (torch.randint(0, 256, (args.batch, H, W, 3), generator=gc).float() - mean).permute(0, 3, 1, 2).contiguous()
    net = calibrate_heads_(sd, make_net, cal.to(dev), windows)
    del cal

    def step():
        det = net.predict(images, windows, with_masks=True)
        return mdist.all_gather_detections(det.packed(), det.counts, global_batch=world * args.batch,
                                           max_detections=cfg.detection_max_instances), det

    runner = step
    if args.graph:
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = step()
        runner = lambda: (graph.replay(), captured)[1]

    for _ in range(args.warmup):
        out = runner()
    # Three timed repetitions of EXACTLY --steps steps, each bracketed by a barrier + synchronize on both sides and reduced
    # with MAX over ranks; `value` is the MEDIAN repetition (run-to-run spread of one repetition on this pool: ~4 %), the
    # fastest and slowest are reported beside it.
    reps, own_max, own_min = [], [], []
    for _ in range(max(1, args.reps)):
        mdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = runner()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0          # this rank's K steps, before it waits for the others
        mdist.barrier()
        reps.append(mdist.max_over_ranks(time.perf_counter() - t0, dev))
        own_max.append(mdist.max_over_ranks(own, dev))
        own_min.append(-mdist.max_over_ranks(-own, dev))   # a straggler shows as max >> min the first time SCALE runs
    elapsed = sorted(reps)[len(reps) // 2]
    (gathered, gcounts), det = out
    n_images = world * args.batch * args.steps
    value = n_images / elapsed

    mean_valid = round(float(net_last_counts(net, images, windows)), 1)
    # ---- context, never `value`: the same K steps with TWO batches in flight (consecutive steps on alternating HIP streams, as a
    # serving loop would double-buffer them). The tail of a step (top-k, NMS, the small pyramid levels) fills few CUs; a second
    # step's trunk runs beside it. Same timing discipline; single GPU only (the per-step all-gather stays on one stream).
    pipelined = None
    if world == 1 and args.in_flight > 1 and not args.graph:
        strs = [torch.cuda.Stream() for _ in range(args.in_flight)]
        torch.cuda.synchronize()
        sub_batches, net.sub_batches = net.sub_batches, 1   # whole batches in flight: the streams are already full of them

        def run_on(i):
            with torch.cuda.stream(strs[i % len(strs)]):
                return net.predict(images, windows, with_masks=True)

        for i in range(args.warmup):
            run_on(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            run_on(i)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        net.sub_batches = sub_batches
        pipelined = {"batches_in_flight": args.in_flight, "value": round(args.batch * args.steps / el2, 2), "unit": "images/s",
                     "ms_per_step_amortised": round(el2 / args.steps * 1e3, 3), "steps": args.steps,
                     "note": "throughput with consecutive steps on alternating streams; the per-step latency is NOT this figure, "
                             "`value` / `ms_per_step` above are one step at a time"}
    # ---- roofline pass: per-launch HIP events around every conv launch (same stream) ----------------
    roofline = None
    roofline_ops = None
    if rank == 0 and args.roofline_steps > 0:
        ops.CONV_PROFILE = []
        for _ in range(args.roofline_steps):
            net.predict(images, windows, with_masks=True)
        torch.cuda.synchronize()
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        roofline = conv_roofline(prof, args, H, W, modules, measured, measured_reason)
        roofline_ops = op_rooflines(dev, ops)

    # ---- other contraction modes, same weights/inputs/steps (every rank takes part) --------------------
    alts = []
    notes = {"f16x3": "fp16-operand MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate), error-compensated 3-product split for "
                      "every conv (no Winograd): fp32-grade, held to the same 1e-4 parity bar as f32 (tests/test_gpu_*.py)",
             "f32+f16x3": "the f32 mode's Winograd / stem / k-blocked layers unchanged (exact-fp32 MFMA); the long-K GEMM-shaped "
                          "layers (Bottleneck conv3 of C4/C5, downsample convs, classifier GEMMs, mask deconv + conv5) on the "
                          "fp16x3 split: same 1e-4 parity bar (tests/test_gpu_pipeline.py, test_gpu_fullsize.py)",
             "f16": "plain fp16 operands and fp16 activations in HBM (the fp16 MFMA path of BASELINE configs[4]); 2e-2 of range"}
    for mode in [m.strip() for m in args.alt_precision.split(",") if m.strip() not in ("", "none", args.precision)]:
        net_alt = make_net(sd, mode)

        def step_alt():
            d = net_alt.predict(images, windows, with_masks=True)
            return mdist.all_gather_detections(d.packed(), d.counts, global_batch=world * args.batch,
                                               max_detections=cfg.detection_max_instances)
        for _ in range(args.warmup):
            step_alt()
        mdist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_alt()
        torch.cuda.synchronize()
        mdist.barrier()
        el_alt = mdist.max_over_ranks(time.perf_counter() - t1, dev)
        alt = {"precision": mode, "value": round(n_images / el_alt, 3), "unit": "images/s",
               "ms_per_step": round(el_alt / args.steps * 1e3, 3),
               "note": notes.get(mode, "") + "; reported for information, not the headline"}
        if rank == 0 and args.roofline_steps > 0:
            ops.CONV_PROFILE = []
            for _ in range(args.roofline_steps):
                net_alt.predict(images, windows, with_masks=True)
            torch.cuda.synchronize()
            prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
            ms = sum(r[0].elapsed_time(r[1]) for r in prof)
            fl = sum(r[2] for r in prof)
            alt["conv_algorithmic_tflops"] = round(fl / (ms * 1e-3) / 1e12, 1)
            alt["conv_ms_per_step"] = round(ms / args.roofline_steps, 3)
        alts.append(alt)
        del net_alt
    alt = alts[0] if alts else None

    # ---- BASELINE configs[4] geometry on this GPU (ResNet-101-FPN, 832 x 1344 = 1333 x 800 padded to /64, plain-fp16 MFMA
    # path with fp16 activations in HBM), after the headline and never part of `value` -------------------------------
    alt_configs = None
    if rank == 0 and world == 1 and args.alt_injected:
        try:
            alt_configs = [injected_entry(net, images, windows, args, cfg, dev, ops)]
        except Exception as e:
            log(f"[bench] ERROR: injected-proposals entry failed: {e!r}")
            alt_configs = [{"config": "configs[2] with injected proposals", "error": repr(e)}]
    if rank == 0 and world == 1 and args.alt_detect:
        try:
            alt_configs = (alt_configs or []) + [detect_entry(net, images, windows, args, cfg, dev)]
        except Exception as e:
            log(f"[bench] ERROR: detect entry failed: {e!r}")
            alt_configs = (alt_configs or []) + [{"config": "detect() on uint8 images", "error": repr(e)}]
    sub_batches = net.sub_batches
    if rank == 0 and world == 1 and args.alt_config5:
        del net
        torch.cuda.empty_cache()
        try:
            alt_configs = (alt_configs or []) + [config5_entry(dev, args, ops, modules, InferenceConfig, MaskRCNNInference)]
        except Exception as e:
            log(f"[bench] ERROR: alt_configs failed: {e!r}")
            alt_configs = (alt_configs or []) + [{"config": "configs[4] geometry", "error": repr(e)}]

    cpu, cpu_failed = None, False
    if rank == 0 and world == 1 and args.cpu_images > 0:
        try:
            cpu = cpu_baseline(sd, cfg, args.cpu_images, seed=1000)
        except Exception as e:  # keep the GPU number, but never report a silent null: the error is in the line
            log(f"[bench] ERROR: cpu_baseline failed: {e!r}")
            cpu = {"value": None, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": "FAILED", "error": repr(e)}
            cpu_failed = True

    td = torch.distributed
    dist_backend = td.get_backend() if (td.is_available() and td.is_initialized()) else None
    rccl_ranks = td.get_world_size() if dist_backend == "nccl" else 0
    mdist.check_gather_errors()   # the device-side flag of poisoned detection blocks, read once, after every timed region
    if rank == 0:
        line = {
            "metric": f"images/sec at {H}x{W}, {args.proposals} proposals/img (Mask R-CNN inference hot path)",
            "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "timed_repetitions": {"count": len(reps), "steps_each": args.steps, "value_is": "median",
                                  "images_per_s": [round(n_images / t, 2) for t in reps],
                                  "min": round(n_images / max(reps), 2), "max": round(n_images / min(reps), 2)},
            "per_rank_ms_per_step": {"max": [round(t / args.steps * 1e3, 3) for t in own_max],
                                     "min": [round(t / args.steps * 1e3, 3) for t in own_min],
                                     "note": "slowest / fastest rank's OWN K steps (clock stopped after its synchronize, before the "
                                             "closing barrier) per timed repetition; `value` / `ms_per_step` are barrier to barrier, "
                                             "MAX over ranks"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dist_backend": dist_backend, "rccl_ranks": rccl_ranks,
            "dtype": {"f32": "f32", "f16x3": "f16x3 split operands, f32 accumulate",
                      "f16": "f16 operands, f32 accumulate"}[args.precision],
            "data": "synthetic (seeded uint8-range images minus MEAN_PIXEL; reference-init random weights, "
                    "randomised BN stats, head layers rescaled so proposals are non-degenerate)",
            "config": {"workload": f"configs[2] per GPU: full {args.arch}-FPN + RoIAlign + NMS inference, "
                                   f"batch={args.batch} synthetic {H}x{W}, "
                                   f"{args.proposals} proposals/img, {cfg.detection_max_instances} mask slots/img",
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}: image shards, replicated weights, one RCCL all-gather of "
                                      f"detections [{world * args.batch},{cfg.detection_max_instances},6]"
                                      + (" — REHEARSAL: all ranks on one GPU over gloo, not a measurement" if mdist.rehearsal() else ""),
                       "hipgraph": bool(args.graph),
                       "concurrent_sub_batches": sub_batches,   # 1 = the batch as one launch sequence (every mode but "f16")
                       "conv3x3": (("winograd F(4x4,3x3) on maps of >= 8 tiles of 16x32 pixels per image"
                                    + ("" if modules.WINOGRAD4_TRUNK else " (FPN smoothing and RPN only)")
                                    + ", F(2x2,3x3) elsewhere; fp32 arithmetic on the fp32 MFMA"
                                    if modules.WINOGRAD4 else "winograd F(2x2,3x3), fp32 arithmetic on the fp32 MFMA")
                                   if (args.precision == "f32" and modules.WINOGRAD) else "direct implicit GEMM"),
                       "mean_valid_proposals": mean_valid,
                       "mean_detections": round(float(det.counts.float().mean().item()), 1)},
            "roofline": roofline, "roofline_ops": roofline_ops, "cpu_baseline": cpu, "alt_precision": alt,
            "alt_precisions": alts,
            "alt_configs": alt_configs,
            "pipelined_throughput": pipelined,
        }
        print(json.dumps(line), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if cpu_failed:
        raise SystemExit(3)


def injected_proposals(batch, count, seed=1234):
    """SURVEY.md 8(d) "synthetic input - proposals": `count` seeded proposals per image — centres U(0,1)^2, h and w log-uniform in
    [0.02, 0.6], clipped to [0,1] — for throughput runs with random weights. → (rois [batch,count,4] normalised (y1,x1,y2,x2),
    counts int32 [batch] == count)."""
    import math
    g = torch.Generator().manual_seed(seed)
    c = torch.rand(batch, count, 2, generator=g)
    hw = torch.exp(torch.rand(batch, count, 2, generator=g) * (math.log(0.6) - math.log(0.02)) + math.log(0.02))
    rois = torch.cat([c - hw / 2, c + hw / 2], -1).clamp(0.0, 1.0).contiguous()
    return rois, torch.full((batch,), count, dtype=torch.int32)


def injected_entry(net, images, windows, args, cfg, dev, ops):
    """The headline step with every RoI slot live: the metric says "1000 proposals/img", the RPN of random weights leaves
    ~790 after NMS (config.mean_valid_proposals), and the heads skip the empty slots as the reference does (model.py:1366-1374).
    Here the proposal stage still runs, then the heads get SURVEY 8(d)'s injected proposals: proposal_count valid RoIs per image.
    Same timing discipline as the headline; never `value`."""
    p = min(cfg.proposal_count, cfg.pre_nms_limit)
    rois, counts = injected_proposals(images.size(0), p)
    ro = (rois.to(dev), counts.to(dev))
    for _ in range(args.warmup):
        net.predict(images, windows, with_masks=True, rois_override=ro)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        det = net.predict(images, windows, with_masks=True, rois_override=ro)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"config": f"configs[2] per GPU with {p} VALID proposals per image injected after the proposal stage (SURVEY 8d: seed 1234, "
                     "centres U(0,1)^2, h / w log-uniform in [0.02, 0.6]); the proposal stage itself still runs",
           "precision": args.precision, "value": round(images.size(0) * args.steps / el, 2), "unit": "images/s",
           "ms_per_step": round(el / args.steps * 1e3, 3), "steps": args.steps,
           "mean_detections": round(float(det.counts.float().mean().item()), 1)}
    if args.roofline_steps > 0:
        ops.CONV_PROFILE = []
        for _ in range(args.roofline_steps):
            net.predict(images, windows, with_masks=True, rois_override=ro)
        torch.cuda.synchronize()
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        rows = [r for r in prof if len(r) > 7]
        peak = F32_MFMA_PEAK_TFLOPS if args.precision == "f32" else F16_MFMA_PEAK_TFLOPS
        if rows:
            per = len(rows) // args.roofline_steps
            fc1 = [rows[i * per] for i in range(args.roofline_steps)]
            ms = sum(r[0].elapsed_time(r[1]) for r in fc1) / len(fc1)
            out["heads_rows"] = {k: fc1[0][7][k] for k in ("rows_slots", "rows_executed", "rows_valid")}
            out["classifier_fc1"] = {"ms": round(ms, 4), "executed_tflops": round(fc1[0][6] / (ms * 1e-3) / 1e12, 1),
                                     "executed_frac": round(fc1[0][6] / (ms * 1e-3) / 1e12 / peak, 4)}
        out["conv_ms_per_step"] = round(sum(r[0].elapsed_time(r[1]) for r in prof) / args.roofline_steps, 3)
    return out


def detect_entry(net, images, windows, args, cfg, dev):
    """SURVEY 8f-4 under the clock: MaskRCNNInference.detect() — resize + pad + mean-subtract (utils.resize_image, mold_image),
    the whole predict step, full-size mask pasting and the mapping back to the original frames (datalib.full_masks,
    decode_boxes / decode_masks) — on `batch` uint8 RGB images of configs[0]'s size (1200 x 1920, seeded noise; already in HBM,
    as the headline's inputs are). Beside it predict() alone on the molded batch of the SAME images: detect must stay within
    10 % of it. Needs the square 1024 canvas of the reference; never `value`."""
    if cfg.image_height != cfg.image_width or cfg.image_height != cfg.image_max_dim:
        return {"config": "detect() on uint8 images", "skipped": "needs the square IMAGE_MAX_DIM canvas"}
    from maskrcnn_amd import image as imagelib
    b = images.size(0)
    g = torch.Generator().manual_seed(58)
    raw = [torch.randint(0, 256, (1200, 1920, 3), generator=g, dtype=torch.uint8).to(dev) for _ in range(b)]
    molded, win, _ = imagelib.mold_inputs(raw, cfg, dev)
    for _ in range(args.warmup):
        net.predict(molded, win, with_masks=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net.predict(molded, win, with_masks=True)
    torch.cuda.synchronize()
    el_p = time.perf_counter() - t0
    for _ in range(args.warmup):
        res = net.detect(raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = net.detect(raw)
    torch.cuda.synchronize()
    el_d = time.perf_counter() - t0
    split = {"mold_ms": 0.0, "predict_ms": 0.0, "paste_decode_ms": 0.0}
    reps = 3
    for _ in range(reps):
        t = {}
        net.detect(raw, timings=t)
        for k in split:
            split[k] += t[k] / reps
    n_det = sum(0 if r[0] is None else int(r[0].numel()) for r in res)
    mask_bytes = sum(0 if r[3] is None else r[3].numel() * r[3].element_size() for r in res)
    return {"config": f"detect(): {b} uint8 RGB images of 1200 x 1920 (configs[0]'s size) -> class ids, scores, boxes and FULL-SIZE masks "
                      "in each image's own frame (model.py:1095-1138, utils.py:42-90, data.py:264-314); same weights as the headline",
            "value": round(b * args.steps / el_d, 2), "unit": "images/s", "ms_per_step": round(el_d / args.steps * 1e3, 3),
            "predict_only_on_the_same_molded_batch": {"value": round(b * args.steps / el_p, 2), "ms_per_step": round(el_p / args.steps * 1e3, 3)},
            "detect_over_predict": round(el_p / el_d, 4), "steps": args.steps,
            "split_ms": {k: round(v, 3) for k, v in split.items()},
            "split_note": "HIP events around the three stages of one detect() call (mean of 3); one host synchronisation per batch "
                          "(the detection counts) sits between predict and paste",
            "detections": n_det, "full_size_mask_bytes": mask_bytes}


def config5_entry(dev, args, ops, modules, InferenceConfig, MaskRCNNInference):
    """BASELINE configs[4] on ONE GPU: ResNet-101-FPN, 832 x 1344, batch 8, precision "f16" (fp16 operands and fp16
    activations in HBM, fp32 accumulate; parity bar of this mode: 2e-2 of the activation range, tests/test_gpu_fullsize.py
    ::test_config5_*). Same timing discipline as the headline (warm-up, --steps steps between synchronisations)."""
    H, W, batch = 832, 1344, 8
    cfg = InferenceConfig(image_height=H, image_width=W, backbone="resnet101", pre_nms_limit=args.proposals,
                          proposal_count=args.proposals)
    sd = modules.synthetic_state_dict("resnet101", seed=0, bn_seed=1)
    mean = torch.tensor(cfg.mean_pixel)
    g = torch.Generator().manual_seed(5)
    images = (torch.randint(0, 256, (batch, H, W, 3), generator=g).float() - mean).permute(0, 3, 1, 2).contiguous().to(dev)
    windows = torch.tensor([[0.0, 0.0, float(H), float(W)]] * batch, device=dev)
    net = calibrate_heads_(sd, lambda s_: MaskRCNNInference(s_, cfg, dev, precision="f16"), images, windows)
    for _ in range(args.warmup):
        net.predict(images, windows, with_masks=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        det = net.predict(images, windows, with_masks=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ops.CONV_PROFILE = []
    for _ in range(max(1, args.roofline_steps)):
        net.predict(images, windows, with_masks=True)
    torch.cuda.synchronize()
    prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
    ms = sum(r[0].elapsed_time(r[1]) for r in prof) / max(1, args.roofline_steps)
    fl = sum(r[2] for r in prof) / max(1, args.roofline_steps)
    # every launch against ITS OWN roofline: the larger of its MFMA time at the fp16 peak and its HBM time at 8 TB/s on the
    # algorithmic bytes (the stem's fp32 MFMAs are priced at the fp16 peak too, i.e. against it)
    floor_ms = sum(max(r[2] / (F16_MFMA_PEAK_TFLOPS * 1e12), r[4] / (HBM_PEAK_GBS * 1e9)) * 1e3 for r in prof) / max(1, args.roofline_steps)
    hbm_bound = sum(1 for r in prof if r[4] / (HBM_PEAK_GBS * 1e9) > r[2] / (F16_MFMA_PEAK_TFLOPS * 1e12)) // max(1, args.roofline_steps)
    # the same sum with every REFERENCE LAYER on its own floor, whatever launch computes it (a launch that fuses a Bottleneck is then
    # priced on its layers' floors — the yardstick of rounds 3-4, when every layer was a launch: fusing layers lowers the
    # per-launch floor sum with the time, so the per-launch fraction cannot show what a fusion bought)
    def layer_floor(r):
        layers = r[7]["layers"] if len(r) > 7 and isinstance(r[7], dict) and "layers" in r[7] else [(r[2], r[4])]
        return sum(max(f / (F16_MFMA_PEAK_TFLOPS * 1e12), b / (HBM_PEAK_GBS * 1e9)) for f, b in layers) * 1e3
    layer_floor_ms = sum(layer_floor(r) for r in prof) / max(1, args.roofline_steps)
    return {"config": "BASELINE configs[4] geometry on 1 GPU: ResNet-101-FPN, 832x1344 (1333x800 padded to /64), batch 8, "
                      f"{args.proposals} proposals/img, fp16 MFMA path (fp16 operands + fp16 activations in HBM, fp32 accumulate)",
            "precision": "f16", "value": round(batch * args.steps / el, 2), "unit": "images/s",
            "concurrent_sub_batches": net.sub_batches,
            "concurrent_sub_batches_note": "the timed step runs the batch as this many equal sub-batches on concurrent HIP streams, "
                                           "joined before predict() returns (same tensors bit for bit; the mode's default); the "
                                           "per-launch pass below times whole-batch launches one at a time",
            "ms_per_step": round(el / args.steps * 1e3, 3), "steps": args.steps,
            "tolerance": "this mode's bars, not the 1e-4 one (tests/test_gpu_fullsize.py): trunk 2e-2 of the activation range vs the fp32 "
                         "oracle; masks on the same boxes 3e-2 abs (sigmoid outputs in [0,1]) vs the fp32 masks; detections: >= 95 % of "
                         "the fp32 path's confident detections have a same-class fp16 detection at IoU >= 0.9, none below 0.5 "
                         "(three seeds x eight images)",
            "conv_ms_per_step": round(ms, 3), "conv_algorithmic_tflops": round(fl / (ms * 1e-3) / 1e12, 1),
            "conv_frac_of_f16_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 4),
            "conv_frac_of_per_launch_roofline": round(floor_ms / ms, 4),
            "conv_frac_of_per_layer_roofline": round(layer_floor_ms / ms, 4),
            "conv_launches": len(prof) // max(1, args.roofline_steps), "conv_launches_hbm_bound_at_peak": hbm_bound,
            "per_layer_roofline_note": "as conv_frac_of_per_launch_roofline, but a launch that computes several of the reference's "
                                       "layers (a whole C2 Bottleneck, csrc/bottleneck_f16.hip) is priced at the SUM of those layers' own "
                                       "floors: comparable with rounds 3-4, when every layer was a launch",
            "per_launch_roofline_note": "sum over launches of max(algorithmic FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) / summed "
                                        "launch durations: most 1x1 layers of this path have their HBM floor above their fp16-MFMA floor; "
                                        "the fp16-MFMA launches are power-bound on random data (the same kernel on all-zero operands "
                                        "runs 1.3x faster: tools/f16_dvfs_probe.py, DESIGN.md 5.2a), so the nominal 2.5 PFLOP/s is not "
                                        "reachable on real activations",
            "mean_detections": round(float(det.counts.float().mean().item()), 1)}


def net_last_counts(net, images, windows):
    _, mid = net.predict(images, windows, with_masks=False, return_intermediates=True)
    return mid["roi_counts"].float().mean().item()


if __name__ == "__main__":
    main()
