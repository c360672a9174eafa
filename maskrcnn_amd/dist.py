"""Multi-GPU data parallelism for inference (the reference has no distributed code).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU tests).
Images are independent (BatchNorm is frozen, model.py:1009-1016,1142), so the batch shards contiguously
across ranks with replicated weights and NO collective on the data path. The single exchange step is an
all-gather of the fixed-shape detections block [B_local, D, 6] fp32 + int32 counts (9.6 KB per rank at
B_local = 8, D = 50): latency-bound, one call per batch, never bucketed with anything else.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; returns (rank, local, world).
    A plain `python bench.py` run (no env) is world_size 1 with no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # under torchrun (RANK set) the group is created even at world size 1, so the RCCL path can be exercised
    # on a single GPU (MRCNN_FORCE_COLLECTIVE=1 makes the helpers below issue the collectives there too)
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_range(global_batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous image range [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_detections(packed: torch.Tensor, counts: torch.Tensor):
    """packed [B_local, D, 6] fp32, counts [B_local] int32 → ([world*B_local, D, 6], [world*B_local]),
    rank-major (== global image order under shard_range with equal shards). Identity at world 1."""
    if not _collectives_on():
        return packed, counts
    world = dist.get_world_size()
    packed = packed.contiguous()
    counts = counts.contiguous()
    out_p = packed.new_empty((world * packed.size(0),) + tuple(packed.shape[1:]))
    out_c = counts.new_empty(world * counts.size(0))
    dist.all_gather_into_tensor(out_p, packed)
    dist.all_gather_into_tensor(out_c, counts)
    return out_p, out_c


def _collectives_on() -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MRCNN_FORCE_COLLECTIVE") == "1"


def barrier():
    if _collectives_on():
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not _collectives_on():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
