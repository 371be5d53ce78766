"""Rank plumbing for the multi-GPU path: reads shard by contiguous ranges, the index is replicated,
there is NO collective on the data path.  torch.distributed is used only to
bracket the timed region and to combine (max time, sum bases); backend "nccl" (= RCCL) on GPUs,
"gloo" in the CPU tests."""
from __future__ import annotations

import os


def env_rank() -> tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(rank: int, world: int, n_items: int) -> tuple[int, int]:
    """Contiguous, input-order shard [first, last) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def weak_shard_first_read(rank: int, reads_per_rank: int) -> int:
    """Weak scaling: every rank owns reads_per_rank reads; rank r generates reads [r*n, (r+1)*n)."""
    return rank * reads_per_rank


def combine(elapsed_s: float, units: float, device: str = "cpu") -> tuple[float, float]:
    """(max over ranks of elapsed, sum over ranks of units).  Identity when not initialised."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed_s, units
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
