"""Multi-GPU layer of the hot path: one process per GPU, weights and text features replicated and
resident, the image batch sharded on dim 0, ONE all-gather of the per-GPU L2-normalised image embeddings per step
(``torch.distributed`` backend "nccl" = RCCL over xGMI) before the shared ``img @ txt^T`` kernel.  Gather, not reduce:
results are bitwise independent of the number of ranks.  ECE accumulators (3*(n_bins+1) float64) merge with one
all-reduce at the end of an evaluation, not per step.

The reference has no counterpart (single process + nn.DataParallel re-broadcasting the weights every call,
coop.py:268-272, tempscaling.py:117-120)."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) of a global batch of n that rank owns: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_embeddings(local: torch.Tensor, group=None) -> torch.Tensor:
    """[b, E] per rank -> [world*b, E] on every rank, rank-major (equal b on every rank)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    try:
        dist.all_gather_into_tensor(out, local, group=group)
    except (RuntimeError, NotImplementedError):   # backends without the flat form (older gloo)
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, local, group=group)
    return out


def all_gather_ragged(local: torch.Tensor, n_global: int, group=None) -> torch.Tensor:
    """Ragged variant for a global batch that does not divide evenly: pads each shard to the largest shard, gathers,
    and strips the padding so that row i of the result is global row i."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_global, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    gathered = all_gather_embeddings(pad, group)
    return torch.cat([gathered[r * width: r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def merge_ece_bins(bins: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank (count, sum_conf, sum_correct) accumulators; exact for the counts, fp64 sums otherwise."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(bins, op=dist.ReduceOp.SUM, group=group)
    return bins


def all_gather_varlen(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate per-rank tensors whose dim-0 lengths differ (rank-major): lengths are exchanged first, shards padded to
    the longest, one all-gather, padding stripped."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(x.item()) for x in lens]
    width = max(lens)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    gathered = all_gather_embeddings(pad, group)
    return torch.cat([gathered[r * width: r * width + lens[r]] for r in range(world)], dim=0)


def gather_samples(evaluator, proximity=None, group=None):
    """End-of-evaluation merge for a sharded test split: sums the bin accumulators and replaces the evaluator's kept
    (conf, pred, gt) vectors by the all-rank concatenation, so that ``evaluate`` returns identical numbers on every rank.
    Returns the gathered proximity vector (or None)."""
    merge_ece_bins(evaluator.bins, group)
    if evaluator.keep_samples and evaluator._conf:
        evaluator._conf = [all_gather_varlen(torch.cat(evaluator._conf), group)]
        evaluator._pred = [all_gather_varlen(torch.cat(evaluator._pred), group)]
        evaluator._gt = [all_gather_varlen(torch.cat(evaluator._gt), group)]
    return None if proximity is None else all_gather_varlen(proximity, group)
