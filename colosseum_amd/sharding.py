"""Sharding of independent instances over the GPUs of one node.

Instances (environment parameterisation x seed, enumerated seed-major like the reference's
colosseum/experiment/folder_structuring.py:76-104) are split into contiguous blocks, one per rank; there is
no data-path collective.  The only communication is the final gather of per-instance results
(`gather_instances`), an all-gather over RCCL/xGMI on GPUs ("nccl" backend) or gloo in the CPU tests."""
from typing import Tuple

import numpy as np


def shard_range(n_instances: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: instance i goes to rank i * world // n (first ranks get the remainder)."""
    base, rem = divmod(n_instances, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_instances(local: np.ndarray, n_instances: int, dist=None, device=None) -> np.ndarray:
    """All ranks obtain the [n_instances, ...] result array from their contiguous shards (ragged shards are
    padded to the largest one for the collective)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert len(local) == n_instances
        return np.asarray(local)
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_instances, rank, world)
    assert len(local) == hi - lo, f"rank {rank} holds {len(local)} results for shard [{lo},{hi})"
    longest = -(-n_instances // world)
    pad = np.zeros((longest,) + tuple(local.shape[1:]), local.dtype)
    pad[: len(local)] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    parts = []
    for r in range(world):
        a, b = shard_range(n_instances, r, world)
        parts.append(out[r][: b - a].cpu().numpy())
    return np.concatenate(parts)
