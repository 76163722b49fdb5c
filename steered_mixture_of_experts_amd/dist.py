"""Multi-GPU plumbing: independent blocks are sharded across ranks (one process per GPU);
the only exchange on the path is a 3-scalar all-reduce (RCCL over xGMI when the backend is
"nccl") for the global loss / MSE / kernel count at validation cadence (SURVEY 8(e):
smoe.py:1758-1761 accumulate exactly these three numbers over blocks on the host).
Works unchanged on the "gloo" backend (CPU tests)."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch


def world() -> Tuple[int, int]:
    """(rank, world_size) of the default process group, (0, 1) when not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(num_blocks: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous range [lo, hi) of the row-major block index owned by ``rank``:
    ceil(B/R) blocks per rank, the last ranks may own fewer (or none)."""
    per = -(-num_blocks // world_size)
    lo = min(num_blocks, rank * per)
    hi = min(num_blocks, lo + per)
    return lo, hi


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of a small tensor (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allgather_blocks(local: np.ndarray, num_blocks: int) -> np.ndarray:
    """Untimed end-of-run assembly: concatenate every rank's per-block array (leading axis =
    local blocks, shard_range order) into the full (num_blocks, ...) array on every rank."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return local
    parts: List[object] = [None] * ws
    dist.all_gather_object(parts, local)
    out = np.concatenate([p for p in parts if p is not None and len(p) > 0], axis=0)
    assert out.shape[0] == num_blocks, (out.shape, num_blocks)
    return out
