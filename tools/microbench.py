#!/usr/bin/env python3
"""Microbenchmarks of the two non-conv hot ops, HIP-event timed on the launch stream.

  * RoIAlign (BASELINE config 2): 256 RoIs x 256 ch x 14x14 on one FPN level, NCHW drop-in entry point and
    the NHWC pyramid kernel; roofline = compulsory bytes (output + touched map once + boxes) / time vs 8 TB/s.
  * NMS: µs per call at N = 500 / 1000 / 2000 (threshold 0.7) and the batched 8 x 1000 form; latency-bound.
  * image pre-/post-processing: mold_inputs of 8 camera-sized images and full_masks of 8 x 50
    detections on a 1024^2 canvas; HBM-bound byte work, roofline = bytes that must move / time vs 8 TB/s, with the
    CPU oracle (C restatement of Pillow's resample, 1 thread) and Pillow itself timed beside it.
Prints one JSON object per line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from maskrcnn_amd import ops  # noqa: E402

HBM_PEAK_GBS = 8000.0


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # µs


def cpu_reference(g):
    """BASELINE config 2 'vs c++ext CPU': the reference's own compiled CPU extension (oracle/_ref, test
    infrastructure) on this host, same inputs. Skipped when oracle/_ref is absent."""
    import contextlib
    import time
    try:
        from oracle import build_ref
        ref = build_ref.load()
    except Exception:
        ref = None
    if ref is None:
        return

    @contextlib.contextmanager
    def mute():  # crop_cpu.cpp:163 printf()s on every call
        import ctypes
        sys.stdout.flush()
        saved, devnull = os.dup(1), os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            yield
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)

    fm = torch.randn(1, 256, 256, 256, generator=g)
    c = torch.rand(256, 2, generator=g)
    hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
    boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1)
    ind = torch.zeros(256, dtype=torch.int32)
    crops = torch.zeros(256, 256, 14, 14)
    with mute():
        ref.crop_forward(fm, boxes, ind, 0.0, 14, 14, crops)
        t0 = time.perf_counter()
        for _ in range(3):
            ref.crop_forward(fm, boxes, ind, 0.0, 14, 14, crops)
        ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"op": "crop_forward_cpu_reference (oracle/_ref, 1 thread)", "level_hw": 256, "rois": 256,
                      "ms": round(ms, 1)}), flush=True)
    for n in (500, 1000):
        cc = torch.rand(n, 2, generator=g) * 1000
        d = torch.cat([cc, cc + torch.rand(n, 2, generator=g) * 80 + 4, torch.rand(n, 1, generator=g)], 1)
        ref.nms(d, 0.7)
        t0 = time.perf_counter()
        for _ in range(5):
            ref.nms(d, 0.7)
        print(json.dumps({"op": "nms_cpu_reference (oracle/_ref, 1 thread)", "n": n,
                          "ms": round((time.perf_counter() - t0) / 5 * 1e3, 2)}), flush=True)


def image_bench(dev):
    import time

    import numpy as np

    from maskrcnn_amd import image as imagelib
    from maskrcnn_amd.config import InferenceConfig
    cfg = InferenceConfig()
    rng = np.random.default_rng(1234)
    # ---- pre-processing: 8 images 480x640 -> 768x1024 -> 1024^2 canvas, fp32 CHW
    images = [rng.integers(0, 256, (480, 640, 3), dtype=np.uint8) for _ in range(8)]
    dimgs = [torch.from_numpy(a).to(dev) for a in images]
    us = timeit(lambda: imagelib.mold_inputs(dimgs, cfg, dev), iters=20)
    moved = 8 * (480 * 640 * 3 + 3 * 1024 * 1024 * 4)
    print(json.dumps({"op": "mold_inputs (resize 480x640->768x1024, pad, mean, CHW)", "images": 8, "us": round(us, 1),
                      "algorithmic_MB": round(moved / 1e6, 1), "GBps": round(moved / us / 1e3, 1),
                      "frac_of_8TBps": round(moved / us / 1e3 / HBM_PEAK_GBS, 4),
                      "launches": 8 * 4}), flush=True)
    # ---- post-processing: 400 detections (8 images x 50) pasted at full size
    n, c = 400, 81
    cy, cx = rng.random(n), rng.random(n)
    hh = np.exp(rng.random(n) * 3.4 - 3.9)
    ww = np.exp(rng.random(n) * 3.4 - 3.9)
    y1 = np.clip((cy - hh / 2) * 1024, 0, 1023).round()
    x1 = np.clip((cx - ww / 2) * 1024, 0, 1023).round()
    y2 = np.clip((cy + hh / 2) * 1024, y1 + 1, 1024).round()
    x2 = np.clip((cx + ww / 2) * 1024, x1 + 1, 1024).round()
    boxes = np.stack([y1, x1, y2, x2], 1).astype(np.float32)
    masks = (1.0 / (1.0 + np.exp(-rng.normal(0, 3, (n, 28, 28, c))))).astype(np.float32)
    ids = rng.integers(1, c, n).astype(np.int64)
    dm, di, db = torch.from_numpy(masks).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(boxes).to(dev)
    us = timeit(lambda: ops.paste_masks(dm, di, db, 1024, 1024, channels_last=True), iters=20)
    moved = n * 1024 * 1024 + n * 28 * 28 * 4
    print(json.dumps({"op": "full_masks (paste_masks, 28x28 -> box -> 1024^2 canvas)", "detections": n,
                      "us": round(us, 1), "algorithmic_MB": round(moved / 1e6, 1), "GBps": round(moved / us / 1e3, 1),
                      "frac_of_8TBps": round(moved / us / 1e3 / HBM_PEAK_GBS, 4),
                      "mean_box_area_px": float(((y2 - y1) * (x2 - x1)).mean())}), flush=True)
    # ---- CPU beside it: the oracle (test infrastructure) and, when importable, Pillow itself
    try:
        from oracle import oracle
    except Exception:
        return
    t0 = time.perf_counter()
    for a in images[:4]:
        img, *_ = oracle.resize_image(a, cfg.image_min_dim, cfg.image_max_dim, True)
        oracle.mold_image(img, cfg.mean_pixel)
    print(json.dumps({"op": "mold_inputs_cpu_oracle (1 thread)", "images": 4,
                      "ms_per_image": round((time.perf_counter() - t0) / 4 * 1e3, 2)}), flush=True)
    t0 = time.perf_counter()
    oracle.full_masks(torch.from_numpy(ids[:50]), torch.from_numpy(boxes[:50]),
                      torch.from_numpy(masks[:50]).permute(0, 3, 1, 2), 1024, 1024)
    print(json.dumps({"op": "full_masks_cpu_oracle (1 thread)", "detections": 50,
                      "ms_per_50": round((time.perf_counter() - t0) * 1e3, 2)}), flush=True)
    try:
        from PIL import Image
    except Exception:
        return
    t0 = time.perf_counter()
    for a in images[:4]:
        r = np.array(Image.fromarray(a).resize((1024, 768), Image.BILINEAR))
        p = np.pad(r, [(128, 128), (0, 0), (0, 0)])
        torch.from_numpy((p.astype(np.float32) - np.asarray(cfg.mean_pixel)).transpose(2, 0, 1)).float()
    print(json.dumps({"op": "mold_inputs_cpu_pillow+numpy (the reference's third-party path, 1 thread)", "images": 4,
                      "ms_per_image": round((time.perf_counter() - t0) / 4 * 1e3, 2)}), flush=True)
    t0 = time.perf_counter()
    for i in range(50):
        m = Image.fromarray(masks[i, :, :, ids[i]] * 255.0).convert("L")
        m = m.resize((int(x2[i] - x1[i]), int(y2[i] - y1[i])), Image.BILINEAR)
        canvas = np.zeros((1024, 1024), dtype=np.uint8)
        canvas[int(y1[i]):int(y1[i]) + m.height, int(x1[i]):int(x1[i]) + m.width] = np.array(m)
        torch.from_numpy(canvas) > 127
    print(json.dumps({"op": "full_masks_cpu_pillow+numpy (the reference's third-party path, 1 thread)",
                      "detections": 50, "ms_per_50": round((time.perf_counter() - t0) * 1e3, 2)}), flush=True)


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1234)
    if "--image-only" in sys.argv:
        return image_bench(dev)
    cpu_reference(torch.Generator().manual_seed(1234))
    for hl in (256, 128, 64, 32):
        fm = torch.randn(1, 256, hl, hl, generator=g).to(dev)
        c = torch.rand(256, 2, generator=g)
        hw = torch.rand(256, 2, generator=g) * 0.10 + 0.02
        boxes = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
        ind = torch.zeros(256, dtype=torch.int32, device=dev)
        out_bytes = 256 * 256 * 14 * 14 * 4
        algo = out_bytes + fm.numel() * 4 + 256 * 20
        us = timeit(lambda: ops.crop(fm, boxes, ind, 0.0, 14, 14))
        print(json.dumps({"op": "crop_forward_nchw", "level_hw": hl, "rois": 256, "us": round(us, 2),
                          "algorithmic_MB": round(algo / 1e6, 2), "GBps": round(algo / us / 1e3, 1),
                          "frac_of_8TBps": round(algo / us / 1e3 / HBM_PEAK_GBS, 4)}), flush=True)
    # NHWC pyramid kernel, 8 images x 1000 rois x 7x7 (the classifier-head call of the pipeline)
    fms = [torch.randn(8, 1024 // s, 1024 // s, 256, generator=g).to(dev) for s in (4, 8, 16, 32)]
    c = torch.rand(8000, 2, generator=g)
    hw = torch.exp(torch.rand(8000, 2, generator=g) * 3.4 - 3.9)
    rois = torch.cat([c - hw / 2, c + hw / 2], 1).clamp(0, 1).to(dev)
    for pool in (7, 14):
        n = 8000 if pool == 7 else 400
        us = timeit(lambda: ops.roi_align_pyramid(fms, rois[:n], pool, 1024.0 * 1024.0, rois_per_image=n // 8))
        out_bytes = n * pool * pool * 256 * 4
        print(json.dumps({"op": "roi_align_pyramid_nhwc", "pool": pool, "rois": n, "us": round(us, 2),
                          "output_MB": round(out_bytes / 1e6, 1),
                          "GBps_out_plus_4taps": round(out_bytes * 5 / us / 1e3, 1)}), flush=True)
    for n in (500, 1000, 2000, 4096):
        k = max(1, n // 10)
        centres = torch.rand(k, 2, generator=g) * 1024
        cc = centres[torch.randint(0, k, (n,), generator=g)] + torch.randn(n, 2, generator=g) * 12
        wh = torch.exp(torch.rand(n, 2, generator=g) * 2.5 + 2.0)
        d = torch.cat([cc - wh / 2, cc + wh / 2, torch.rand(n, 1, generator=g)], 1).to(dev).unsqueeze(0)
        us = timeit(lambda: ops.nms_batched(d, 0.7), iters=30)
        keep, cnt = ops.nms_batched(d, 0.7)
        print(json.dumps({"op": "nms", "segments": 1, "n": n, "threshold": 0.7, "kept": int(cnt[0]),
                          "us": round(us, 1)}), flush=True)
    d8 = torch.stack([torch.cat([torch.rand(1000, 2, generator=g) * 900,
                                 torch.rand(1000, 2, generator=g) * 900, torch.rand(1000, 1, generator=g)], 1)
                      for _ in range(8)])
    d8[..., 2:4] = d8[..., :2] + torch.exp(torch.rand(8, 1000, 2, generator=g) * 2.5 + 2.0)
    d8 = d8.to(dev)
    us = timeit(lambda: ops.nms_batched(d8, 0.7), iters=30)
    _, cnt = ops.nms_batched(d8, 0.7)
    print(json.dumps({"op": "nms", "segments": 8, "n": 1000, "threshold": 0.7,
                      "kept_mean": float(cnt.float().mean()), "us": round(us, 1)}), flush=True)
    image_bench(dev)


if __name__ == "__main__":
    main()
