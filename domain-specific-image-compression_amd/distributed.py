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
    """init_process_group from the torchrun environment; no-op for world size 1."""
    _, _, world = env_rank_world()
    if world == 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return True


def shard_first_index(rank: int, per_rank: int) -> int:
    """Global index of the first image of `rank` (weak scaling: per-rank batch fixed)."""
    return rank * per_rank


def reduce_metric_sums(t: torch.Tensor) -> torch.Tensor:
    """In-place all-reduce(sum) of the per-rank metric sums (fp64 vector)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def max_over_ranks(value: float, device) -> float:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
