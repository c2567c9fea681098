"""Per-image sharding across the GPUs of one node.

The path has no data-path collective: rank r owns images [r*B, (r+1)*B) of the global
batch and a full weight replica.  The only exchange is one all-reduce(sum) of the
metric sums per step (RCCL over xGMI on GPUs; gloo in the CPU tests), matching the
reference's mean-of-per-image-values aggregation (modelseval.py:221-224).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device=None):
    """init_process_group from the torchrun environment (RANK/WORLD_SIZE/MASTER_* set by
    torch.distributed.run); no-op for a plain `python bench.py`."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    if backend != "nccl" or device is None:
        dist.init_process_group(backend, **kw)
        return True
    # RCCL prints a banner ("Hostname", "Librccl path") on STDOUT while the communicator is
    # created (eagerly with device_id, or at the first collective): do both with fd 1 pointed at
    # stderr so that a caller's stdout stays clean (bench.py must print exactly one JSON line).
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group(backend, **kw)
        t = torch.zeros(1, device=device)
        dist.all_reduce(t)
        torch.cuda.synchronize(device)
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    return True


def shard_first_index(rank: int, per_rank: int) -> int:
    """Global index of the first image of `rank` (weak scaling: per-rank batch fixed)."""
    return rank * per_rank


def reduce_metric_sums(t: torch.Tensor) -> torch.Tensor:
    """In-place all-reduce(sum) of the per-rank metric sums (fp64 vector)."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def max_over_ranks(value: float, device) -> float:
    if dist.is_available() and dist.is_initialized():
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
