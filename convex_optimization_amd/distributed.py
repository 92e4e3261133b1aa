"""Data-parallel partitioning of independent images over the GPUs of one node.

The reference solves one 2-D image per call (multiplying_matrix.py:31 unpacks exactly two
dimensions); a batch, or the channels of an RGB image, are independent problems that share the blur
eigenvalues.  They shard evenly over ranks with no collective on the data path; the only exchange is
an all-reduce of the per-image loss / residual records (a few doubles per image and iteration),
used for a global stop decision and for reporting.  One process per GPU; backend "nccl" (RCCL over
xGMI) on the GPU node, "gloo" in CPU tests.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the items owned by `rank`: contiguous blocks, sizes differing by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_loss_records(ring: torch.Tensor, n_items_global: int, lo: int) -> torch.Tensor:
    """ring: [iters][local_images][8] loss records of this rank's images.  Returns the
    [iters][n_items_global][8] records of every image on every rank (sum all-reduce of a tensor that
    is zero outside the rank's own slice)."""
    iters, nloc, w = ring.shape
    full = torch.zeros((iters, n_items_global, w), dtype=ring.dtype, device=ring.device)
    full[:, lo:lo + nloc] = ring
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full


def global_stop(local_converged: torch.Tensor) -> bool:
    """True when every image on every rank has converged (MIN all-reduce of a 0/1 flag)."""
    flag = local_converged.to(torch.int32).min().reshape(1).clone() if local_converged.numel() else torch.ones(1, dtype=torch.int32, device=local_converged.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())
