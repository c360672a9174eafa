"""RCCL under a driver-run test (BASELINE configs[3]: batch 64 over 8 ranks; the reference has no distributed code,
config.py:200-204 GPU_COUNT = 1). The one-GPU box can hold ONE rank on the nccl backend (RCCL refuses two ranks on a
device), so these run the production rank program at world size 1 with the collectives forced on — the same code path an
8-GPU node runs with 8 ranks — always as CHILD processes (torch.distributed.run), never in the pytest process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = dict(os.environ, MRCNN_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="VERSION")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(kw)
    return env


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_rccl_all_gather_world1_child(tmp_path):
    out = tmp_path / "rccl.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_child.py"), str(out)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads(out.read_text())
    assert rec["backend"] == "nccl" and rec["rccl_ranks"] == 1
    assert rec["collectives"] == 3, "one all_gather_into_tensor per step"
    assert rec["max_over_ranks"] == 1.0 and rec["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.gpu
def test_bench_under_torchrun_reports_rccl(tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), at the N
    this box has: the nccl group is initialised, every step ends in the RCCL all-gather, and the line says so."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--reps", "1", "--size", "256", "--batch", "2", "--proposals", "200", "--cpu-images", "0", "--roofline-steps", "0",
           "--alt-precision", "none", "--alt-config5", "0", "--measure-traffic", "0", "--in-flight", "1"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    # stdout is the JSON line and nothing else: RCCL's version banner (NCCL_DEBUG=VERSION on this pool) goes to stderr
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines, r.stdout[:600]
    line = json.loads(lines[0])
    assert line["dist_backend"] == "nccl" and line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 0 and line["config"]["global_batch"] == 2


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_rehearsal():
    """`python bench.py --gpus 2` with no WORLD_SIZE — how the driver may call it — launches its own torch.distributed.run child with
    two ranks. On the one-GPU box both ranks drive cuda:0 and gloo moves the detections block (MRCNN_DIST_REHEARSAL=1: RCCL refuses
    two ranks on one device; never a measurement): the CONTROL FLOW of N > 1 end to end — rendezvous on a free port, per-rank
    shards and seeds, barrier-bracketed timing, MAX over ranks, one all-gather per step, rank 0 alone printing, clean exit."""
    env = _env(MRCNN_DIST_REHEARSAL="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reps", "1",
           "--size", "256", "--batch", "2", "--proposals", "200", "--cpu-images", "0", "--roofline-steps", "0",
           "--alt-precision", "none", "--alt-config5", "0", "--measure-traffic", "0", "--in-flight", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), r.stdout[:600]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["dist_backend"] == "gloo" and line["rccl_ranks"] == 0
    assert "REHEARSAL" in line["config"]["parallelism"] and line["value"] > 0
