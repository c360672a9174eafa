"""Child program of tests/test_gpu_dist.py: ONE rank of a torch.distributed job on the `nccl` backend (= RCCL on ROCm).
Started by `python -m torch.distributed.run --nproc-per-node 1 tests/rccl_child.py <out.json>` with MRCNN_FORCE_COLLECTIVE=1
so the collectives of maskrcnn_amd.dist are really issued at world size 1 (on a node with several GPUs the same program runs
unchanged with more ranks). Checks: init on nccl with the device set first, one all_gather_detections of DEVICE tensors equal
to its input (world 1) / in global image order (world > 1), barrier, max_over_ranks, check_gather_errors, clean shutdown.
Not a test module itself (no test_ prefix): pytest does not collect it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

from maskrcnn_amd import dist as mdist  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, local, world = mdist.init_from_env()
    assert td.is_initialized() and td.get_backend() == "nccl", td.get_backend()
    dev = torch.device("cuda", mdist.device_index(local))
    assert torch.cuda.current_device() == dev.index
    b, d = 8, 50
    g = torch.Generator().manual_seed(100 + rank)
    packed = torch.rand(b, d, 6, generator=g).to(dev)
    packed[:, :, 0] = float(rank)
    counts = torch.randint(0, d + 1, (b,), generator=g, dtype=torch.int32).to(dev)
    calls = []
    orig = td.all_gather_into_tensor
    td.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    for _ in range(3):   # first step validates the layout, the later ones run the sync-free steady state
        gp, gc = mdist.all_gather_detections(packed, counts, global_batch=world * b, max_detections=d)
    td.all_gather_into_tensor = orig
    torch.cuda.synchronize()
    assert gp.is_cuda and gc.is_cuda and gp.shape == (world * b, d, 6) and gc.dtype == torch.int32
    assert torch.equal(gp[rank * b:(rank + 1) * b], packed) and torch.equal(gc[rank * b:(rank + 1) * b], counts)
    assert gp[:, 0, 0].tolist() == [float(r) for r in range(world) for _ in range(b)]
    mdist.check_gather_errors()
    mdist.barrier()
    t = mdist.max_over_ranks(float(rank + 1), dev)
    assert t == float(world)
    if rank == 0:
        with open(out_path, "w") as fh:
            json.dump({"backend": td.get_backend(), "rccl_ranks": td.get_world_size(), "collectives": len(calls),
                       "max_over_ranks": t, "device": str(dev),
                       "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}, fh)
    td.destroy_process_group()


if __name__ == "__main__":
    main()
