"""Multi-GPU data parallelism for inference (the reference has no distributed code).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU tests).
Images are independent (BatchNorm is frozen, model.py:1009-1016,1142), so the batch shards contiguously
across ranks with replicated weights and NO collective on the data path. The single exchange step is an
all-gather of the fixed-shape detections block [B_local, D+1, 6] fp32 (detections + a row carrying the count;
9.8 KB per rank at B_local = 8, D = 50): latency-bound, ONE collective per batch, never bucketed with anything else.
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
        if "MASTER_PORT" not in os.environ:
            # every launcher (torchrun, bench.py's own self-launch) sets it; a fixed default would make two jobs on one node
            # collide. Alone (world 1) any free port will do; several ranks must be TOLD the same one.
            if world > 1:
                raise RuntimeError("maskrcnn_amd.dist: MASTER_PORT is not set (launch with torch.distributed.run, or "
                                   "`python bench.py --gpus N`, which picks a free port)")
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # MRCNN_DIST_REHEARSAL=1: several ranks on ONE GPU box (all on cuda:0, gloo moves the detections block): RCCL refuses
            # two ranks on one device, and the multi-rank control flow of bench.py can be exercised this way where only one GPU
            # is to be had. Never a measurement: the ranks share the card.
            backend = "gloo" if (rehearsal() or not torch.cuda.is_available()) else "nccl"
        if backend == "nccl":
            torch.cuda.set_device(local)
            # RCCL creates a group's communicator at init (device_id given: eager) or at its first collective, and with
            # NCCL_DEBUG=VERSION (exported on this pool's boxes) it prints its version banner there — on STDOUT, from C. A
            # benchmark's stdout is one JSON line: both happen here, with file descriptor 1 pointing at stderr meanwhile.
            with _stdout_to_stderr():
                # device_id binds the group to this rank's GPU (barrier() then needs no guess about "the current device")
                dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
                t = torch.zeros(1, device=torch.device("cuda", local))
                dist.all_reduce(t)
                torch.cuda.synchronize()
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class _stdout_to_stderr:
    """File descriptor 1 → 2 for the duration (C-level writes included); Python's own buffer is flushed first."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def rehearsal() -> bool:
    return os.environ.get("MRCNN_DIST_REHEARSAL") == "1"


def device_index(local_rank: int) -> int:
    """The GPU a rank drives: its LOCAL_RANK; in a one-GPU rehearsal every rank drives cuda:0."""
    return 0 if rehearsal() else local_rank


def shard_range(global_batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous image range [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_detections(packed: torch.Tensor, counts: torch.Tensor, global_batch: int | None = None,
                          max_detections: int | None = None):
    """packed [B_local, D, 6] fp32, counts [B_local] int32 → ([G, D, 6], [G] int32) in global image order
    (rank-major under shard_range). Identity at world 1.

    ONE collective per step: the counts ride in the same block as an extra row ([B, D+1, 6], row D = (count, 0...);
    exact in fp32 for counts < 2^24). `global_batch` (G) tells how the batch was sharded: ranks then own
    shard_range(G, r, world) images, which differ by one image when G % world != 0 — all_gather_into_tensor needs
    identical shapes, so every rank pads its block to ceil(G / world) images (count 0) and the padding is stripped
    after the gather. Without `global_batch` every rank must hold the same B_local (G = world * B_local).
    `max_detections` (D = cfg.detection_max_instances) is what every rank EXPECTS — pass it: it is what lets a rank whose
    block is malformed still send a (poisoned) block of the right shape instead of hanging its peers. Without it the block's
    own D is taken at face value, and a malformed block raises before the collective. Nothing is remembered between calls:
    two pipelines with different D in one process are two independent callers."""
    if not _collectives_on():
        return packed, counts
    world, rank = dist.get_world_size(), dist.get_rank()
    # Is this rank's shard what every other rank expects? Decided BEFORE any of its sizes is used to shape the block: a rank
    # whose shard disagrees with shard_range (or is not [b, D, 6] at all) must neither raise before the collective — the other
    # ranks would sit in all_gather_into_tensor until the backend's timeout — nor send a block of another shape. It sends a
    # poisoned block of the EXPECTED shape (count rows = -1) and EVERY rank raises after the gather.
    well_formed = packed.dim() == 3 and packed.size(2) == 6 and counts.dim() == 1
    d_exp = max_detections
    if d_exp is None:
        if not well_formed:
            raise RuntimeError("all_gather_detections: a malformed block and no max_detections: the expected shape is "
                               "unknown (pass max_detections=cfg.detection_max_instances)")
        d_exp = packed.size(1)
    b = packed.size(0) if packed.dim() >= 1 else 0   # never negative: it shapes the (possibly poisoned) block below
    if global_batch is None:
        # every rank claims its own b: nothing to check it against, b IS the expectation
        sizes = [b] * world
    else:
        sizes = [hi - lo for lo, hi in (shard_range(global_batch, r, world) for r in range(world))]
    b_max = max(sizes)
    ok = well_formed and sizes[rank] == b and counts.numel() == b and packed.size(1) == d_exp
    d = d_exp
    block = torch.zeros(b_max, d + 1, 6, dtype=torch.float32, device=counts.device if not well_formed else packed.device)
    if ok:
        block[:b, :d] = packed
        block[:b, d, 0] = counts.to(block.dtype)
    else:
        block[:, d, 0] = -1.0
    out = block.new_empty(world * b_max, d + 1, 6)
    dist.all_gather_into_tensor(out, block)
    if not ok:
        raise RuntimeError(f"all_gather_detections: rank {rank} holds a block of shape {tuple(packed.shape)} with "
                           f"{counts.numel()} counts; shard_range({global_batch}, {rank}, {world}) x max_detections says "
                           f"[{sizes[rank]}, {d}, 6]")
    # The healthy ranks learn of it from the gathered count rows. Looking at them is a device-to-host read, so it is done on
    # the FIRST step of every sharding layout (set-up-time validation) — and on every LATER step the same test is folded into
    # a device-side flag without a host read (`check_gather_errors()` reads it once, where the caller can afford a sync: after
    # its timed region, at the end of a serving window); the poisoned counts (-1) travel on to the caller in any case.
    layout = (tuple(sizes), d, str(out.device))
    bad_rows = out[:, d, 0] < 0
    if layout not in _VALIDATED:
        neg = bad_rows.nonzero().flatten().tolist()
        if neg:
            bad = sorted({int(i) // b_max for i in neg})
            raise RuntimeError(f"all_gather_detections: rank(s) {bad} hold a shard that disagrees with "
                               f"shard_range({global_batch}, r, {world})")
        _VALIDATED.add(layout)
        # the device-side flag of the later steps is allocated HERE — this step reads the device (so it is never inside a
        # hipGraph capture), and a flag created during a capture would live in the capture's private pool
        if str(out.device) not in _ERR_FLAG:
            _ERR_FLAG[str(out.device)] = torch.zeros(world, dtype=torch.int32, device=out.device)
    else:
        flag = _ERR_FLAG[str(out.device)]
        if flag.numel() != world:   # another process group since: a fresh flag (outside any capture: sizes only change at set-up)
            flag = _ERR_FLAG[str(out.device)] = torch.zeros(world, dtype=torch.int32, device=out.device)
        flag += bad_rows.view(world, b_max).any(dim=1).to(torch.int32)   # per rank; no host read
    if min(sizes) != b_max:  # strip the padding rows (index list cached per layout: no per-step host-to-device copy)
        out = out.index_select(0, _strip_index(tuple(sizes), b_max, out.device))
    return out[:, :d].contiguous(), out[:, d, 0].to(torch.int32)


def check_gather_errors() -> None:
    """Reads (one device-to-host copy) and clears the device-side flags all_gather_detections keeps after its first step:
    raises if any rank sent a poisoned block since the last check."""
    for key, flag in list(_ERR_FLAG.items()):
        hits = flag.tolist()
        flag.zero_()
        bad = [r for r, n in enumerate(hits) if n]
        if bad:
            raise RuntimeError(f"all_gather_detections: rank(s) {bad} sent a poisoned detections block "
                               f"({[hits[r] for r in bad]} step(s)) since the last check")


_ERR_FLAG: dict = {}
_VALIDATED: set = set()
_STRIP_CACHE: dict = {}


def _strip_index(sizes: tuple, b_max: int, device) -> torch.Tensor:
    key = (sizes, b_max, str(device))
    idx = _STRIP_CACHE.get(key)
    if idx is None:
        rows = [r * b_max + i for r in range(len(sizes)) for i in range(sizes[r])]
        idx = _STRIP_CACHE[key] = torch.tensor(rows, dtype=torch.int64, device=device)
    return idx


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
