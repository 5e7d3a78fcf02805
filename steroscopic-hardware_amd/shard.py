"""Multi-GPU sharding of the frame stream.

Independent stereo frames shard round-robin across ranks (frame i -> rank i mod G), one process
per GPU; there is NO data-path collective.  The only exchange is the two algorithm parameters
(pkg/despair/params.go: BlockSize, MaxDisparity), broadcast from rank 0 -- over RCCL/xGMI on
GPUs (backend "nccl"), over gloo in the CPU tests.
"""
import os


def frame_indices(rank, world, frames_per_rank):
    """Stream indices owned by ``rank``: rank, rank+world, ... (frames_per_rank of them)."""
    return [rank + world * k for k in range(frames_per_rank)]


def frame_seeds(rank, world, frames_per_rank, base=1000):
    """Synthetic-frame seeds of BASELINE config 4: frame i has seed 1000+i."""
    return [base + i for i in frame_indices(rank, world, frames_per_rank)]


def init(backend=None, device=None):
    """Joins the process group described by RANK/WORLD_SIZE/MASTER_* (torch.distributed.run).
    Returns (dist or None, rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return None, rank, world, local
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend=backend or "nccl", rank=rank, world_size=world, **kw)
    return dist, rank, world, local


def broadcast_params(dist, rank, block, maxdisp, device):
    """Rank 0's (BlockSize, MaxDisparity) reach every rank: one 2 x int32 broadcast."""
    import torch
    t = torch.tensor([block, maxdisp] if rank == 0 else [0, 0], dtype=torch.int32, device=device)
    if dist is not None:
        dist.broadcast(t, src=0)
    b, d = (int(v) for v in t.cpu())
    return b, d


def max_over_ranks(dist, seconds, device):
    import torch
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(dist):
    if dist is not None:
        dist.barrier()
