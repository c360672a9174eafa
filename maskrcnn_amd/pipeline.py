"""Batched, host-sync-free counterpart of MaskRCNN.predict (reference model.py:1140-1203).

    trunk (C1-C5 + FPN) → RPN → proposal decode + NMS → RoIAlign 7x7 → classifier head
          → per-class NMS + top-k → RoIAlign 14x14 → mask head

Differences from the reference, all by design:
  * any batch size (the reference is batch-1 everywhere: model.py:296,1321,1349);
  * fixed shapes: every image carries `proposal_count` proposal slots and `detection_max_instances`
    detection slots, zero-padded, with int32 valid counts — so there is no nonzero()/unique()/item()
    host synchronisation anywhere and the whole step can be captured in a hipGraph;
  * the Python per-class loop (model.py:1454-1475) is one class-aware NMS launch per batch;
  * activations are channels-last (NHWC) between the NCHW boundary and the outputs.
Every kernel of the step — convs/GEMMs, RoIAlign, NMS, the softmax / box-decode glue, top-k and the selection
gathers — is in libmaskrcnn_hip.so; torch does device memory and streams.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch

from . import image as imagelib
from . import modules, ops
from .anchors import pyramid_anchors
from .config import InferenceConfig


@dataclass
class Detections:
    """Fixed-shape result for a batch of B images (D = detection_max_instances)."""
    class_ids: torch.Tensor   # int64 [B, D]   0 in unused slots
    scores: torch.Tensor      # fp32  [B, D]
    boxes: torch.Tensor       # fp32  [B, D, 4] pixel (y1,x1,y2,x2), integral-valued
    counts: torch.Tensor      # int32 [B] valid detections per image
    masks: torch.Tensor | None  # fp32 [B, D, 28, 28, num_classes] (NHWC) sigmoid masks

    def packed(self) -> torch.Tensor:
        """[B, D, 6] fp32 (class_id, score, y1, x1, y2, x2): the all-gather payload."""
        return torch.cat([self.class_ids.float().unsqueeze(-1), self.scores.unsqueeze(-1), self.boxes], -1)


def max_batch_per_launch(cfg: InferenceConfig) -> int:
    """Largest per-GPU batch one pass of the step takes: EVERY batch-scaled tensor stays under the kernels' 2^30-element limit
    (32-bit byte offsets), so that no layer's kernel choice depends on the batch and no launch is refused. Per image: the RPN's
    512-channel shared activation on P2, (H/4)(W/4) x 512 = 32 HW elements (the stem's output is 16 HW) — what bounds large
    images: 31 at 1024^2, 29 at 832 x 1344 —, and the RoI heads' tensors, which bound small images with many proposals: the
    pooled crops P x pool^2 x 256, the classifier's P x 1024 activations, the mask head's D x (2 mask_pool)^2 x 256 up-sampled
    map and its D x (2 mask_pool)^2 x classes output (256^2 with 1000 proposals: 85, not 511). predict() splits larger
    batches into equal sub-batches."""
    p = min(cfg.proposal_count, cfg.pre_nms_limit)
    d = min(cfg.detection_max_instances, p)
    up = (2 * cfg.mask_pool_size) ** 2
    per_image = max(32 * cfg.image_height * cfg.image_width,
                    p * cfg.pool_size * cfg.pool_size * 256, p * 1024, p * cfg.num_classes * 5,
                    d * up * 256, d * up * cfg.num_classes)
    return max(1, ((1 << 30) - 1) // per_image)


class MaskRCNNInference:
    def __init__(self, state_dict: dict, cfg: InferenceConfig | None = None, device="cuda:0",
                 precision: str = "f32", concurrent_sub_batches: int | None = None):
        """concurrent_sub_batches: predict() runs a batch as this many equal sub-batches on concurrent HIP streams (joined before
        it returns; the result is the same tensor bit for bit — image i of a batch equals image i alone). None = the measured
        default: 2 in the "f16" mode, whose launches are short and leave CUs idle at their edges (configs[4]: 875 → 918 images/s),
        1 otherwise (the fp32 step loses 2 % to the smaller launches); MRCNN_SUB_BATCHES overrides.
        precision: contraction mode of every conv/GEMM (modules.ConvWeight): "f32" exact-fp32 MFMA (default,
        the parity mode), "f16x3" fp16-operand MFMA with the error-compensated 3-product split (fp32-grade),
        "f16" plain fp16 operands (BASELINE config 5); "f32+f16x3" (modules.MIXED): fp32 Winograd / stem / k-blocked layers as in
        "f32", fp16x3 split for the long-K GEMM-shaped layers. Activations are fp32 in HBM in every mode but "f16"."""
        assert precision in modules.PRECISIONS + (modules.MIXED,), precision
        self.cfg = cfg or InferenceConfig()
        self.precision = precision
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MaskRCNNInference runs on the GPU only (no CPU path)")
        c = self.cfg
        self.backbone = modules.FusedBackbone(state_dict, c.backbone, self.device, precision=precision)
        self.rpn = modules.FusedRPN(state_dict, self.device, precision=precision)
        self.classifier = modules.FusedClassifier(state_dict, self.device, precision=precision)
        self.mask = modules.FusedMask(state_dict, self.device, precision=precision)
        self.anchors = pyramid_anchors(c).to(self.device)
        self.image_area = float(c.image_height * c.image_width)
        self.max_batch = max_batch_per_launch(c)
        env = os.environ.get("MRCNN_SUB_BATCHES")
        self.sub_batches = int(env) if env else (concurrent_sub_batches if concurrent_sub_batches is not None
                                                  else (2 if precision == "f16" else 1))
        self._side_streams, self._keep, self._keep_event = [], None, None

    # ---------------------------------------------------------------- stage 1: proposals
    def rpn_heads(self, fms):
        """rpn_detect (model.py:1294-1304) → fg scores [B,A], deltas [B,A,4], A = 261888 at 1024^2."""
        kb = getattr(self.backbone, "kblocked", None) or [None] * len(fms)
        # per level [B,H,W,18] head outputs, or the in-kernel head sums of the large levels → one launch
        return ops.rpn_scores_deltas([self.rpn(p, k) for p, k in zip(fms, kb)], self.rpn.b_head)

    def proposals(self, scores, deltas):
        """rpn_refine (model.py:1307-1382), batched. → rois [B,P,4] normalised (zero beyond count),
        counts int32 [B], plus the dets handed to NMS (for parity tests)."""
        c = self.cfg
        k = min(c.pre_nms_limit, self.anchors.size(0))
        top, order = ops.topk_desc(scores, k)                            # model.py:1345-1348
        # gather + boxes_scale + boxes_refine + boxes_clamp_ (:1341-1358) in one launch
        dets = ops.proposal_decode(self.anchors, deltas, order, top, c.rpn_bbox_std_dev, c.image_height,
                                   c.image_width)
        keep, kept = ops.nms_batched(dets, c.rpn_nms_threshold)          # :1364, score order == index order
        # keep[:proposal_count], gather, normalise (:1366-1374) in one launch
        rois, counts = ops.proposal_select(dets, keep, kept, min(c.proposal_count, k), c.image_height,
                                           c.image_width)
        return rois, counts, dets

    # ---------------------------------------------------------------- stage 2: detections
    def detections(self, rois, roi_counts, logits, bbox, windows):
        """mrn_refine (model.py:1389-1487), batched and sync-free.
        rois [B,P,4]; logits [B*P,C]; bbox [B*P,C,4]; windows [B,4] pixel (y1,x1,y2,x2).
        → class ids, scores, pixel boxes, normalised boxes for the mask head, counts."""
        c = self.cfg
        p = rois.size(1)
        # softmax/argmax (:791,1407-1415), delta gather, boxes_refine, scale, window clip, round (:1418-1432) and
        # the validity rule (:1437-1443) in one launch; excluded slots get a unique negative NMS class so they
        # neither suppress nor are suppressed in the class-aware pass that replaces the per-class loop (:1454-1475)
        dets, cls, class_ids = ops.detection_decode(logits, bbox, rois.contiguous(), roi_counts,
                                                    windows.to(torch.float32).contiguous(), c.rpn_bbox_std_dev,
                                                    c.image_height, c.image_width,
                                                    float(c.detection_min_confidence or 0.0))
        keep, kept = ops.nms_batched(dets, c.detection_nms_threshold, class_ids=cls)
        # keep ∩ foreground, top detection_max_instances by score, gathers (:1475-1487) in one launch
        return ops.detection_select(dets, cls, class_ids, keep, kept, min(c.detection_max_instances, p),
                                    c.image_height, c.image_width)

    # ---------------------------------------------------------------- whole step
    @torch.no_grad()
    def predict(self, images: torch.Tensor, windows: torch.Tensor, with_masks: bool = True,
                return_intermediates: bool = False, rois_override=None, host_counts=None, _whole: bool = False):
        """images [B,3,H,W] fp32 NCHW, already molded (resized/padded, mean-subtracted: model.py:1102-1110);
        windows [B,4] pixel (y1,x1,y2,x2) of the un-padded image area.
        rois_override = (rois [B,P,4] normalised, counts int32 [B]): measurement aid (SURVEY.md §8d "synthetic input —
        proposals"): the proposal stage still runs — its launches stay in the step — but the heads see these RoIs instead of
        its output, e.g. proposal_count VALID proposals per image whatever random weights make of the RPN.
        host_counts = (pinned int32 tensor [>= B], torch.cuda.Event): the detection counts are copied there (asynchronously)
        and the event recorded as soon as they exist, ahead of the mask head (detect() uses it)."""
        c = self.cfg
        assert images.is_cuda and images.dtype == torch.float32
        b = images.size(0)
        assert tuple(images.shape[1:]) == (3, c.image_height, c.image_width)
        if b > self.max_batch:
            # Which kernel a layer takes (F(4x4) / F(2x2) / fused conv3) must never depend on the batch — image i of a batch
            # equals image i alone bit for bit — but the kernels address tensors with 32-bit element offsets (< 2^30 elements),
            # so past max_batch a layer would fall to another kernel. Oversized batches therefore run as equal sub-batches.
            assert not return_intermediates, f"return_intermediates needs batch <= {self.max_batch} at this image size"
            assert host_counts is None, "host_counts: one launch group per call (batch <= max_batch)"
            n = -(-b // self.max_batch)
            step = -(-b // n)
            ro = rois_override
            parts = [self.predict(images[i:i + step], windows[i:i + step], with_masks,
                                  rois_override=None if ro is None else (ro[0][i:i + step], ro[1][i:i + step]))
                     for i in range(0, b, step)]
            cat = lambda f: torch.cat([getattr(p_, f) for p_ in parts], 0)
            return Detections(cat("class_ids"), cat("scores"), cat("boxes"), cat("counts"), cat("masks") if with_masks else None)
        k = self.sub_batches
        if (k > 1 and not _whole and b >= k and b % k == 0 and not return_intermediates and host_counts is None
                and ops.CONV_PROFILE is None          # per-launch event passes time whole-batch launches, one at a time
                and not torch.cuda.is_current_stream_capturing()):
            return self._predict_concurrent(images, windows, with_masks, rois_override, k)
        fms = self.backbone(images)                                        # [P2..P6] NHWC
        scores, deltas = self.rpn_heads(fms)
        rois, roi_counts, rpn_dets = self.proposals(scores, deltas)
        if rois_override is not None:
            o_rois, o_counts = rois_override
            assert tuple(o_rois.shape) == tuple(rois.shape) and o_rois.dtype == torch.float32 and o_rois.is_cuda
            assert tuple(o_counts.shape) == (b,) and o_counts.dtype == torch.int32 and o_counts.is_cuda
            rois, roi_counts = o_rois.contiguous(), o_counts.contiguous()
        p = rois.size(1)
        flat = rois.reshape(-1, 4).contiguous()
        # slots beyond an image's proposal count hold no RoI: RoIAlign and the head's GEMMs skip them (the reference's rois
        # tensor has only the surviving rows, model.py:1366-1374); their logits / bbox rows are never read (detections())
        # (only when the head's GEMMs honour the counts — the exact-fp32 kernel: the fp16-MFMA modes compute every row, so for
        # them RoIAlign fills every slot too and no uninitialised row ever enters a GEMM)
        skip = roi_counts if (modules.SKIP_EMPTY_ROI_TILES and self.classifier.honours_row_counts()) else None
        pooled = ops.roi_align_pyramid(fms[:4], flat, c.pool_size, self.image_area, rois_per_image=p,
                                       out_f16=self.classifier.wants_f16(), roi_counts=skip)
        logits, bbox = self.classifier(pooled, skip, p)
        ids, det_scores, boxes, mrois, counts = self.detections(rois, roi_counts, logits, bbox,
                                                                windows.to(self.device))
        if host_counts is not None:
            # detect(): the detection counts start their way to the host HERE, before the mask head is enqueued — by the time the
            # host has them (one event wait) the GPU still has the mask head to run, so building the per-image index lists and
            # enqueueing the paste / decode launches costs the GPU no idle time
            host_counts[0][:b].copy_(counts, non_blocking=True)
            host_counts[1].record()
        masks = None
        if with_masks:
            d = boxes.size(1)
            # the reference divides all four coordinates by h (model.py:1188), which is only right for
            # square inputs; here (y,x) are divided by (h,w)
            mp = ops.roi_align_pyramid(fms[:4], mrois.view(-1, 4), c.mask_pool_size, self.image_area,
                                       rois_per_image=d, out_kblocked=self.mask.wants_kblocked(c.mask_pool_size),
                                       out_f16=self.mask.wants_f16())
            m = self.mask(mp)
            masks = m.view(b, d, m.size(1), m.size(2), m.size(3))
        det = Detections(ids, det_scores, boxes, counts, masks)
        if return_intermediates:
            if skip is not None:   # the head rows of empty RoI slots were never computed: report them as zeros
                live = (torch.arange(p, device=self.device)[None, :] < roi_counts[:, None]).reshape(-1)
                logits = torch.where(live[:, None], logits, torch.zeros((), device=self.device))
                bbox = torch.where(live[:, None, None], bbox, torch.zeros((), device=self.device))
            return det, dict(feature_maps=fms, rpn_scores=scores, rpn_deltas=deltas, rois=rois,
                             roi_counts=roi_counts, rpn_dets=rpn_dets, logits=logits, bbox=bbox)
        return det

    def _predict_concurrent(self, images, windows, with_masks, rois_override, k):
        """The batch as k equal sub-batches, each on its own HIP stream (fork after the caller's stream, join before returning):
        short launches of one sub-batch run beside the other's instead of leaving CUs idle at their edges. Every tensor of the
        library is the caller's and every launch goes to the stream current at the call, so concurrent steps share nothing but the
        weights. Same result as one batch, bit for bit."""
        b = images.size(0)
        n = b // k
        cur = torch.cuda.current_stream(self.device)
        while len(self._side_streams) < k:
            self._side_streams.append(torch.cuda.Stream(self.device))
        windows = windows.to(self.device)
        for i in range(k):
            self._side_streams[i].wait_stream(cur)
            if self._keep_event is not None:
                self._side_streams[i].wait_event(self._keep_event)   # (the previous call may have come from another stream)
        # The previous call's sub-batch results were allocated on the side streams and read by the caller's stream (the cat below,
        # marked by an event). They are kept alive until here: their blocks can only be reused by allocations on a side stream, and
        # every side stream now waits for that cat. (Tensor.record_stream does the same bookkeeping per block with events of its
        # own; in bench.py's alt entry it cost a quarter of the throughput: 700 against 920 - 950 images/s.)
        self._keep = None
        parts = []
        try:
            for i in range(k):
                with torch.cuda.stream(self._side_streams[i]):
                    ro = None if rois_override is None else (rois_override[0][i * n:(i + 1) * n], rois_override[1][i * n:(i + 1) * n])
                    parts.append(self.predict(images[i * n:(i + 1) * n], windows[i * n:(i + 1) * n], with_masks, rois_override=ro,
                                              _whole=True))
        finally:
            # also when a sub-batch raised: the side streams may still be reading the caller's images / windows / RoIs, which the
            # caller (or the allocator, on the caller's stream) is free to reuse as soon as this call returns
            for i in range(k):
                cur.wait_stream(self._side_streams[i])
        cat = lambda f: torch.cat([getattr(p_, f) for p_ in parts], 0)
        out = Detections(cat("class_ids"), cat("scores"), cat("boxes"), cat("counts"), cat("masks") if with_masks else None)
        self._keep, self._keep_event = parts, torch.cuda.Event()
        self._keep_event.record(cur)
        return out

    def __del__(self):
        ev = getattr(self, "_keep_event", None)
        if ev is not None:          # the kept sub-batch results are freed with the object: not before their reader has run
            try:
                ev.synchronize()
            except Exception:
                pass

    # ---------------------------------------------------------------- images in, full-size masks out
    @torch.no_grad()
    def detect(self, images, timings: dict | None = None):
        """MaskRCNN.detect (model.py:1095-1138) for a list of RGB uint8 [h,w,3] images of any sizes: resize + pad +
        mean-subtract on the GPU (utils.resize_image, mold_image), predict, paste the masks at full size
        (datalib.full_masks, which the reference calls inside predict, model.py:1190) and map boxes and masks back
        to each original image (decode_boxes / decode_masks). ONE host synchronisation per batch (the detection counts, to
        size the per-image results); pasting and decoding run once per group of images that share a size — one launch per
        stage for a batch from one source — over the VALID detections of the group only. → per image (class_ids [n],
        scores [n], boxes [n,4], masks [n,h',w']) device tensors (views of the batch's tensors), or (None, None, None, None)
        when nothing was detected (model.py:1119-1120). The reference returns Python lists (:1132-1135); masks are bool when
        scale == 1 and uint8 grey levels otherwise, as in the reference.
        timings: filled with HIP-event milliseconds of the three stages (mold / predict / paste + decode) when given."""
        c = self.cfg
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timings is not None else None
        if ev:
            ev[0].record()
        molded, windows, metas = imagelib.mold_inputs(images, c, self.device)
        if ev:
            ev[1].record()
        nb = molded.size(0)
        if nb <= self.max_batch:
            if getattr(self, "_host_counts", None) is None or self._host_counts[0].numel() < nb:
                self._host_counts = (torch.empty(max(nb, 64), dtype=torch.int32).pin_memory(), torch.cuda.Event())
            det = self.predict(molded, windows, host_counts=self._host_counts)
            if ev:
                ev[2].record()
            self._host_counts[1].synchronize()                                  # the one host synchronisation (counts only)
            counts = self._host_counts[0][:nb].tolist()
        else:
            det = self.predict(molded, windows)
            if ev:
                ev[2].record()
            counts = det.counts.tolist()
        b, d = det.class_ids.shape
        windows_l = windows.tolist()
        results = [(None, None, None, None)] * b
        # the class channel of every detection's mask, [B*D, mh, mw]: what the paste reads (one gather for the batch)
        m = det.masks.view(b * d, det.masks.size(2), det.masks.size(3), det.masks.size(4))
        ids_flat = det.class_ids.reshape(-1)
        m_cls = torch.gather(m, 3, ids_flat.clamp(min=0).view(-1, 1, 1, 1).expand(-1, m.size(1), m.size(2), 1)).squeeze(3)
        groups = {}
        for i, n in enumerate(counts):
            if n > 0:
                groups.setdefault((metas[i][0], tuple(windows_l[i])), []).append(i)
        for (scale, window), idx in groups.items():
            rows = torch.tensor([i * d + j for i in idx for j in range(counts[i])], dtype=torch.int64).to(self.device, non_blocking=True)
            g_masks = m_cls.index_select(0, rows).unsqueeze(3)                  # [N,mh,mw,1]: class 0 of a one-class tensor
            g_boxes = det.boxes.reshape(-1, 4).index_select(0, rows)
            zeros = torch.zeros(rows.numel(), dtype=torch.int64, device=self.device)
            if scale == 1:
                pasted = ops.paste_masks(g_masks, zeros, g_boxes, c.image_height, c.image_width, channels_last=True)
                out_boxes = g_boxes
            else:
                l8 = ops.paste_masks(g_masks, zeros, g_boxes, c.image_height, c.image_width, channels_last=True, as_l8=True)
                pasted = imagelib.decode_masks(l8, scale, window)
                out_boxes = imagelib.decode_boxes(g_boxes, scale, window)
            off = 0
            for i in idx:
                n = counts[i]
                results[i] = (det.class_ids[i, :n], det.scores[i, :n], out_boxes[off:off + n], pasted[off:off + n])
                off += n
        if ev:
            ev[3].record()
            torch.cuda.synchronize()
            timings.update(mold_ms=ev[0].elapsed_time(ev[1]), predict_ms=ev[1].elapsed_time(ev[2]),
                           paste_decode_ms=ev[2].elapsed_time(ev[3]))
        return results
