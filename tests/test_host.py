"""CPU tests of the host logic: state-dict schema vs the reference's (golden), anchors, BN folding, SAME
padding, the sharding helper and the world_size-2 gloo all-gather of detections."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_schema_matches_reference():
    """A reference checkpoint (mask_rcnn_coco.pth layout) must load into our parameter tree unchanged."""
    from maskrcnn_amd import modules
    z = load_golden("schema")
    keys = [str(k) for k in z["keys"]]
    assert len(keys) == 800  # 800 keys for R101
    sd = modules.reference_schema("resnet101").state_dict()
    assert list(sd.keys()) == keys
    for k, shp in zip(keys, z["shapes"]):
        assert list(sd[k].shape) == [int(v) for v in shp if v >= 0], k
    r50 = modules.reference_schema("resnet50").state_dict()
    trunk = [k[len("fpn."):] for k in r50 if k.startswith("fpn.C")]
    assert trunk == [str(k) for k in z["r50_trunk_keys"]]


def test_synthetic_state_dict_is_deterministic():
    from maskrcnn_amd import modules
    a = modules.synthetic_state_dict("resnet50", 0, 1)
    b = modules.synthetic_state_dict("resnet50", 0, 1)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert float(a["fpn.C2.0.bn1.running_var"].min()) >= 0.5


def test_anchors_match_oracle_and_known_answers(oracle):
    from maskrcnn_amd.anchors import pyramid_anchors
    from maskrcnn_amd.config import InferenceConfig
    a = pyramid_anchors(InferenceConfig())
    assert tuple(a.shape) == (261888, 4)  # model.py:1019
    assert torch.equal(a, oracle.anchors_for(oracle.Cfg()))
    small = pyramid_anchors(InferenceConfig(image_height=256, image_width=384))
    assert torch.equal(small, oracle.anchors_for(oracle.Cfg(256, 384)))
    with pytest.raises(ValueError):
        InferenceConfig(image_height=1000, image_width=1024)  # model.py:978-983


def test_fold_bn_equals_batchnorm_eval():
    from maskrcnn_amd import modules
    g = torch.Generator().manual_seed(0)
    c = 16
    sd = {"c.weight": torch.randn(c, 8, 1, 1, generator=g), "c.bias": torch.randn(c, generator=g),
          "b.weight": torch.rand(c, generator=g) + 0.5, "b.bias": torch.randn(c, generator=g),
          "b.running_mean": torch.randn(c, generator=g), "b.running_var": torch.rand(c, generator=g) + 0.5}
    s, t = modules.fold_bn(sd, "c", "b", "cpu")
    x = torch.randn(2, 8, 5, 5, generator=g)
    want = F.batch_norm(F.conv2d(x, sd["c.weight"], sd["c.bias"]), sd["b.running_mean"], sd["b.running_var"],
                        sd["b.weight"], sd["b.bias"], False, 0.0, 1e-3)
    got = F.conv2d(x, sd["c.weight"]) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
    assert (got - want).abs().max().item() < 1e-5
    s2, t2 = modules.fold_bn(sd, "c", None, "cpu")
    assert s2 is None and torch.equal(t2, sd["c.bias"])
    w = modules.pack_weight(torch.arange(2 * 3 * 2 * 2.).view(2, 3, 2, 2), "cpu", cin_pad=4)
    assert tuple(w.shape) == (2, 2, 2, 4) and bool((w[..., 3] == 0).all())
    assert float(w[1, 0, 1, 2]) == float(torch.arange(24.).view(2, 3, 2, 2)[1, 2, 0, 1])


def test_same_pad_matches_golden():
    from maskrcnn_amd import ops
    z = load_golden("graph_small")
    for k, s, h, w, top, bottom, left, right in z["same_pad"]:
        assert ops.same_pad(int(h), int(w), int(k), int(s)) == (top, left, bottom, right)


def test_shard_range_partitions_the_batch():
    from maskrcnn_amd.dist import shard_range
    for gb, world in ((64, 8), (8, 1), (10, 4), (3, 8)):
        spans = [shard_range(gb, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, q, global_batch):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import importlib.util
    spec = importlib.util.spec_from_file_location("mdist", os.path.join(ROOT, "maskrcnn_amd", "dist.py"))
    mdist = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mdist)
    r, _, w = mdist.init_from_env(backend="gloo")
    d = 50
    lo, hi = mdist.shard_range(global_batch, r, w)
    b_local = hi - lo
    # detections whose content encodes the global image index
    packed = torch.zeros(b_local, d, 6)
    counts = torch.zeros(b_local, dtype=torch.int32)
    for i, gi in enumerate(range(lo, hi)):
        packed[i, :, 0] = gi
        packed[i, :, 1] = torch.arange(d) / d
        packed[i, :, 2:] = gi * 10.0 + torch.arange(4)
        counts[i] = gi + 1
    calls = []
    orig = torch.distributed.all_gather_into_tensor
    torch.distributed.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    gp, gc = mdist.all_gather_detections(packed, counts, global_batch=global_batch, max_detections=d)
    torch.distributed.all_gather_into_tensor = orig
    # two callers with different D in one process are independent (no process-global expectation)
    gp20, _ = mdist.all_gather_detections(packed[:, :20].contiguous(), counts, global_batch=global_batch, max_detections=20)
    assert tuple(gp20.shape) == (global_batch, 20, 6)
    # without max_detections a malformed block cannot be given the expected shape: it raises BEFORE the collective (every rank
    # does the same here, so nobody waits)
    try:
        mdist.all_gather_detections(torch.zeros(b_local, d), counts, global_batch=global_batch)
        raise AssertionError("malformed block without max_detections did not raise")
    except RuntimeError as e:
        assert "max_detections" in str(e)
    # a shard size that disagrees with shard_range is an error, not a hang
    bad = None
    try:
        mdist.all_gather_detections(torch.zeros(b_local + 1, d, 6), torch.zeros(b_local + 1, dtype=torch.int32),
                                    global_batch=global_batch, max_detections=d)
    except RuntimeError as e:
        bad = str(e)
    # ONE rank with a wrong shard (a layout not seen before): that rank raises after the collective, and the healthy rank
    # raises too, from the poisoned count rows — nobody is left waiting inside the collective
    one_sided = None
    extra = 1 if r == w - 1 else 0
    try:
        mdist.all_gather_detections(torch.zeros(b_local + extra, 49, 6), torch.zeros(b_local + extra, dtype=torch.int32),
                                    global_batch=global_batch, max_detections=d)
    except RuntimeError as e:
        one_sided = str(e)
    # steady state (the layout was validated by the first call): the LAST rank goes bad with a block that is not even 3-D.
    # It must still send the expected shape (no hang, no shape mismatch inside the collective) and raise afterwards; the
    # healthy rank returns without a host read, and check_gather_errors() reports the bad rank.
    late = None
    try:
        if r == w - 1:
            mdist.all_gather_detections(torch.zeros(b_local, d), counts, global_batch=global_batch, max_detections=d)
        else:
            mdist.all_gather_detections(packed, counts, global_batch=global_batch, max_detections=d)
            mdist.check_gather_errors()
    except RuntimeError as e:
        late = str(e)
    if w > 1:
        assert late is not None and (("rank(s) [%d]" % (w - 1)) in late or "holds a block" in late), late
    # a 0-dim block without global_batch (b used to become -1 and torch.zeros(-1, ...) raised BEFORE the collective,
    # leaving the peers inside it): the bad rank still takes part, then raises
    zero_dim = None
    try:
        if r == w - 1:
            mdist.all_gather_detections(torch.zeros(()), torch.zeros(0, dtype=torch.int32), max_detections=d)
        else:
            mdist.all_gather_detections(torch.zeros(0, d, 6), torch.zeros(0, dtype=torch.int32), max_detections=d)
    except RuntimeError as e:
        zero_dim = str(e)
    assert (zero_dim is not None) == (r == w - 1) or w == 1, zero_dim
    try:
        mdist.check_gather_errors()
    except RuntimeError:
        pass
    mdist.check_gather_errors()   # cleared by the read above: a second check is silent
    mdist.barrier()
    t = mdist.max_over_ranks(float(r + 1), "cpu")
    q.put((r, tuple(gp.shape), gp[:, 0, 0].tolist(), gp[:, 3, 5].tolist(), gc.tolist(), str(gc.dtype), t, len(calls),
           bad is not None and one_sided is not None and late is not None))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 7, 1])   # even shards; 4 + 3; 1 + 0 (a rank with no image)
def test_all_gather_detections_gloo_world2(global_batch):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q, global_batch)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, shape, ids, last, counts, dtype, t, ncalls, bad_raised in res:
        assert shape == (global_batch, 50, 6)
        assert ids == [float(i) for i in range(global_batch)]      # rank-major == global image order, padding gone
        assert last == [i * 10.0 + 3 for i in range(global_batch)]
        assert counts == [i + 1 for i in range(global_batch)] and dtype == "torch.int32"
        assert t == 2.0                                              # max over ranks
        assert ncalls == 1, "one collective per step"
        assert bad_raised


def test_winograd4_point_set_is_exact_and_dyadic():
    """The F(4x4,3x3) kernel's matrices (csrc/conv_wino4.hip: points 0, +-3/4, +-3/2, inf) are an exact Winograd algorithm
    — checked in float64 on random data — and B^T / A^T hold only dyadic rationals with short mantissas, i.e. the fp32
    kernel applies both transforms without rounding a coefficient (G g G^T is evaluated in double, once)."""
    import numpy as np
    a, b = 0.75, 1.5
    BT = np.array([[a * a * b * b, 0, -(a * a + b * b), 0, 1, 0],
                   [0, -a * b * b, -b * b, a, 1, 0], [0, a * b * b, -b * b, -a, 1, 0],
                   [0, -a * a * b, -a * a, b, 1, 0], [0, a * a * b, -a * a, -b, 1, 0],
                   [0, a * a * b * b, 0, -(a * a + b * b), 0, 1]])
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, a, -a, b, -b, 0], [0, a * a, a * a, b * b, b * b, 0],
                   [0, a ** 3, -a ** 3, b ** 3, -b ** 3, 1]])
    pts, scl = [0, a, -a, b, -b], [1, 1, 1, 1, 1]
    G = np.zeros((6, 3))
    for j, pj in enumerate(pts):
        nj = np.prod([pj - q for l, q in enumerate(pts) if l != j])
        G[j] = np.array([1, pj, pj * pj]) * scl[j] / nj
    G[5] = [0, 0, 1]
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
    y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
    ref = np.array([[np.sum(d[i:i + 3, j:j + 3] * g) for j in range(4)] for i in range(4)])
    assert np.abs(y - ref).max() < 1e-13
    for m in (BT, AT):
        assert np.array_equal(m.astype(np.float32).astype(np.float64), m)


def test_bench_traffic_measurement_degrades_to_a_reason_without_a_gpu(monkeypatch):
    """bench.py measures roofline.traffic itself through rocprofv3 child passes (measure_traffic_in_run). Anything that goes wrong
    there — no rocprofv3, already under a profiler, a pass that fails (here: no GPU in the build container) — must come back
    as (None, reason) and never cost the run its GPU number."""
    import argparse
    import bench
    args = argparse.Namespace(batch=1, arch="resnet50", proposals=10, precision="f32", traffic_timeout=120)
    monkeypatch.setenv("ROCPROFILER_FAKE", "1")
    got, why = bench.measure_traffic_in_run(args, 128, 128)
    assert got is None and "profiler" in why
    monkeypatch.delenv("ROCPROFILER_FAKE")
    if torch.cuda.is_available():
        return   # on a GPU box the passes would succeed: covered by the bench run itself
    import shutil
    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        got, why = bench.measure_traffic_in_run(args, 128, 128)
        assert got is None and "rocprofv3" in why
        return
    got, why = bench.measure_traffic_in_run(args, 128, 128)
    assert got is None and isinstance(why, str) and why


def test_dist_rehearsal_switch(monkeypatch):
    """MRCNN_DIST_REHEARSAL=1 (several ranks on one GPU box over gloo, for rehearsing bench.py's multi-rank control flow): every
    rank drives cuda:0; without it a rank drives its LOCAL_RANK."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mdist_r", os.path.join(ROOT, "maskrcnn_amd", "dist.py"))
    mdist = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mdist)
    monkeypatch.delenv("MRCNN_DIST_REHEARSAL", raising=False)
    assert not mdist.rehearsal() and [mdist.device_index(r) for r in range(4)] == [0, 1, 2, 3]
    monkeypatch.setenv("MRCNN_DIST_REHEARSAL", "1")
    assert mdist.rehearsal() and [mdist.device_index(r) for r in range(4)] == [0, 0, 0, 0]


def test_bench_self_launch_builds_the_torchrun_child_command():
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE) becomes the launcher of `python -m torch.distributed.run
    --nproc-per-node N bench.py --gpus N ...` — a child process, a free port, the caller's own arguments — before anything
    touches the GPU; --dry-run-launch prints that command instead of running it."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2",
                        "--dry-run-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = rec["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    i = cmd.index("--master-addr")
    assert cmd[i + 1] == "127.0.0.1" and cmd[i + 2] == "--master-port" and 1024 < int(cmd[i + 3]) < 65536
    j = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[j + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert rec["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under torchrun (WORLD_SIZE set) bench.py does NOT re-launch: with a world that disagrees with --gpus it refuses
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-launch"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                MASTER_PORT=str(_free_port())), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 8" in r.stderr


def test_bench_roofline_reports_co_dominant_families_together():
    """When the two largest kernel families are within 5 % of each other's summed time (direct implicit GEMM vs F(4x4) in the
    headline step), roofline.frac is the fraction of BOTH together and each is listed — the line cannot quote the better half."""
    import argparse
    import bench
    from types import SimpleNamespace as NS

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    def row(ms, flops, tag, t0=[0.0]):
        a = Ev(t0[0]); t0[0] += ms
        return (a, Ev(t0[0]), flops, (1, 1, 1), 1000, tag)
    peak = bench.F32_MFMA_PEAK_TFLOPS * 1e12
    # direct: 8.3 ms at 0.76 of peak; winograd4: 8.2 ms at 0.62 executed (x4 algorithmic); stem: 0.5 ms
    prof = [row(8.3, 0.76 * peak * 8.3e-3, "direct"), row(8.2, 4 * 0.62 * peak * 8.2e-3, "winograd4"), row(0.5, 1e9, "stem")]
    args = argparse.Namespace(roofline_steps=1, precision="f32", dump_conv=None, batch=8, arch="resnet50", proposals=1000)
    mods = NS(WINOGRAD=True, STEM_KERNEL=True, FUSED_BOTTLENECK=False, RPN_FUSED_HEADS=True, WINOGRAD4=True, WINOGRAD4_TRUNK=True)
    r = bench.conv_roofline(prof, args, 1024, 1024, mods, None, "test")
    assert [c["family"] for c in r["co_dominant"]] == ["direct", "winograd4"]
    assert abs(r["co_dominant"][0]["executed_frac"] - 0.76) < 1e-3 and abs(r["co_dominant"][1]["executed_frac"] - 0.62) < 1e-3
    assert abs(r["frac"] - (0.76 * 8.3 + 0.62 * 8.2) / 16.5) < 1e-3 and r["launches_per_step"] == 2
    assert "conv_igemm_f32" in r["kernel"] and "conv3x3_wino4_f32" in r["kernel"]
    # a clear winner stays alone
    prof = [row(8.3, 0.76 * peak * 8.3e-3, "direct"), row(6.0, 4 * 0.62 * peak * 6.0e-3, "winograd4")]
    r = bench.conv_roofline(prof, args, 1024, 1024, mods, None, "test")
    assert [c["family"] for c in r["co_dominant"]] == ["direct"] and abs(r["frac"] - 0.76) < 1e-3


def test_max_batch_per_launch_bound():
    """pipeline.max_batch_per_launch: the batch up to which every layer keeps the kernel it takes at batch 1 and no launch is
    refused (2^30-element limit of the 32-bit offsets): the RPN's shared activation on P2 bounds large images, the RoI heads'
    tensors bound small images with many proposals (256^2 x 1000 proposals is 85, not 511)."""
    from maskrcnn_amd.config import InferenceConfig
    from maskrcnn_amd.pipeline import max_batch_per_launch
    lim = 1 << 30
    for h, w, p, d in ((1024, 1024, 500, 50), (832, 1344, 1000, 50), (256, 256, 1000, 50), (128, 128, 4096, 4096),
                       (64, 64, 1000, 100)):
        cfg = InferenceConfig(image_height=h, image_width=w, pre_nms_limit=p, proposal_count=p, detection_max_instances=d)
        m = max_batch_per_launch(cfg)
        sizes = lambda b: [b * (h // 4) * (w // 4) * 512, b * p * 49 * 256, b * p * 1024, b * p * 81 * 5,
                           b * d * 28 * 28 * 256, b * d * 28 * 28 * 81, b * d * 14 * 14 * 256]
        assert max(sizes(m)) < lim <= max(sizes(m + 1)), (h, w, p, d, m)
    assert max_batch_per_launch(InferenceConfig(image_height=1024, image_width=1024)) == 31
    assert max_batch_per_launch(InferenceConfig(image_height=256, image_width=256, pre_nms_limit=1000, proposal_count=1000)) == 85


def test_rows_executed_matches_the_kernels_tile_rule():
    """ops.rows_executed (the profiling pass's host restatement of conv_common.hpp::tile_has_rows): an M tile runs iff it holds
    a valid row; valid rows are a prefix of every group of rows_per_group slots. Checked against a brute-force mask, including
    tiles that straddle groups, empty groups and a last partial tile."""
    import random
    from maskrcnn_amd import ops
    rnd = random.Random(0)
    for _ in range(200):
        rpg = rnd.choice([1, 7, 100, 128, 129, 1000])
        groups = rnd.randint(1, 9)
        bm = rnd.choice([32, 128, 256])
        counts = [rnd.choice([0, 0, rpg, rnd.randint(0, rpg)]) for _ in range(groups)]
        m = rpg * groups
        valid = [(i % rpg) < counts[i // rpg] for i in range(m)]
        want = sum(min(bm, m - t) for t in range(0, m, bm) if any(valid[t:t + bm]))
        got_rows, got_valid = ops.rows_executed(counts, rpg, m, bm)
        assert (got_rows, got_valid) == (want, sum(valid)), (counts, rpg, bm)
    # the headline's shape: 8 images x 1000 slots, 788 valid each, 128-row tiles -> 57 of 63 tiles run
    rows, valid = ops.rows_executed([788] * 8, 1000, 8000, 128)
    assert valid == 6304 and rows == 57 * 128   # 6 of the 63 tiles are skipped, one of them the 64-row tail
    assert ops.lib.mrcnn_conv_rows_tile_m(1024) == 128 and ops.lib.mrcnn_conv_rows_tile_m(64) == 256


def test_bench_roofline_books_executed_rows_and_is_recomputable():
    """A row-group GEMM's profile row carries the FLOPs of the tiles that RAN (index 6) and its row statistics (index 7):
    bench.conv_roofline must use them — not 2*M*K*N over every slot — and `frac` must be recomputable from `by_kernel`."""
    import argparse
    import bench
    from types import SimpleNamespace as NS

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    peak = bench.F32_MFMA_PEAK_TFLOPS * 1e12
    k, n = 12544, 1024
    slots, ran, valid = 8000, 7296, 6304
    t_fc1 = 2.0 * ran * k * n / (0.9 * peak) * 1e3       # ms at 0.90 of peak on the work that ran
    rows = [(Ev(0.0), Ev(t_fc1), 2.0 * valid * k * n, (valid, n, k), 1000, "direct", 2.0 * ran * k * n,
             {"rows_slots": slots, "rows_executed": ran, "rows_valid": valid}),
            (Ev(10.0), Ev(12.0), 0.5 * peak * 2e-3, (1, 1, 1), 1000, "direct")]
    args = argparse.Namespace(roofline_steps=1, precision="f32", dump_conv=None, batch=8, arch="resnet50", proposals=1000)
    mods = NS(WINOGRAD=True, STEM_KERNEL=True, FUSED_BOTTLENECK=False, RPN_FUSED_HEADS=True, WINOGRAD4=True, WINOGRAD4_TRUNK=True)
    r = bench.conv_roofline(rows, args, 1024, 1024, mods, None, "test")
    assert r["heads_rows"]["rows_executed"] == ran and r["heads_rows"]["rows_slots"] == slots
    want = (2.0 * ran * k * n + 0.5 * peak * 2e-3) / ((t_fc1 + 2.0) * 1e-3) / peak
    assert abs(r["frac"] - want) < 1e-3
    # every slot booked (round 4's defect) would have read 0.9 * 8000 / 7296 on the first row alone
    bk = r["by_kernel"]
    again = sum(v["executed_gflop_per_step"] for v in bk.values()) * 1e9 / (sum(v["ms_per_step"] for v in bk.values()) * 1e-3) / peak
    assert abs(again - r["conv_path"]["executed_frac"]) < 1e-3 and abs(again - r["frac"]) < 1e-3


def test_afrag_row_permutation_is_a_bijection_with_consecutive_lane_channels():
    """The weight layout contract of the one-launch fp16 kernels (include/maskrcnn_hip.h: mrcnn_pack_afrags_f16; csrc/bottleneck_f16.hip,
    csrc/mask_tail_f16.hip), restated: row rho of 16-channel block cb holds channel (cb>>1)*32 + (rho>>2)*8 + (cb&1)*4 + (rho&3).
    (a) it is a permutation of the channels for every even block count; (b) a lane of v_mfma_f32_16x16x32_f16's accumulator holds
    rows q*4 .. q*4+3 of a block (q = lane >> 4), so accumulators (2h, 2h+1) of a lane are channels h*32 + q*8 .. +7 in order —
    the lane's B-operand fragment (k = h*32 + q*8 + j) of the next 1x1 conv and 16 contiguous bytes of NHWC fp16 output."""
    ch = lambda cb, rho: (cb >> 1) * 32 + (rho >> 2) * 8 + (cb & 1) * 4 + (rho & 3)
    for nb in (4, 6, 16, 64):
        assert sorted(ch(cb, rho) for cb in range(nb) for rho in range(16)) == list(range(16 * nb))
        for h in range(nb // 2):
            for q in range(4):
                lane_channels = [ch(2 * h + half, q * 4 + r) for half in (0, 1) for r in range(4)]
                assert lane_channels == list(range(h * 32 + q * 8, h * 32 + q * 8 + 8))


def test_w4_forensics_decomposition_identifies_synthesised_causes(tmp_path):
    """tools/w4_forensics.py (the analysis behind DESIGN 5.1a'': what exactly is wrong in a wrong F(4x4) tile) on differences
    synthesised from three known causes — one component of one k tile multiplied with another k tile's U, one wave's nine
    components computed from another k tile's input, four positions of one component overwritten in the exchange buffer — and
    its restated transforms against a direct convolution. The tool must name each cause."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("w4_forensics", os.path.join(ROOT, "tools", "w4_forensics.py"))
    f = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(f)
    err, reps = f.selftest(str(tmp_path / "selftest.npz"))
    assert err < 1e-12                                     # B^T, G, A^T at the points 0, +-3/4, +-3/2, inf == a 3x3 convolution
    a, b, c = reps
    assert a["single_component_fit"]["component"] == [2, 4] and a["single_component_fit"]["residual"] < 1e-9
    ca = a["components"][0]
    assert ca["B_operand_fit"]["k_tile"] == 17 and ca["B_operand_fit"]["residual"] < 1e-6
    assert ca["U_used"]["best_other_U_[k_tile, xi, nu]"]["which"] == [15, 2, 4]
    assert b["quadrant_fit"]["wave"] == 2 and min(b["quadrant_fit"]["residuals"]) < 1e-9 and len(b["components"]) == 9
    assert all(cb["A_operand_fit"]["k_tile"] == 40 and cb["V_used"]["best_other_k_tile"]["k_tile"] == 38 for cb in b["components"])
    assert c["positions"] == [8, 9, 10, 11] and c["rounds"] == [1] and c["single_component_fit"]["component"] == [1, 1]
    assert "M_replaced" in c["components"][0]
