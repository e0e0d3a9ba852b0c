"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

Inference shards CLIPS: every rank transcribes a contiguous slice with a full weight replica and
there is no collective on the data path -- only the final gather of the token ids.
The fine-tune step adds exactly two exchanges: the 2-float (sum of masked CE, number of valid
tokens) all-reduce that makes the loss normalisation batch-global (train_whisper_ipa.py:260-261) and
the bucketed all-reduce of the decoder gradients, after which the per-tensor clip runs
(train_whisper_ipa.py:287-303) so single-process semantics are reproduced.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch


def world() -> Tuple[int, int]:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split, ceil(n/world) per rank (the last ranks may get fewer or none)."""
    per = (n_items + world_size - 1) // world_size
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def shard_indices(indices: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Rank r takes slice r of ONE shared draw (every rank must pass the same ``indices``), which keeps a
    DP step equivalent to the single-process ``np.random.choice`` batch (train_whisper_ipa.py:548)."""
    lo, hi = shard_bounds(len(indices), world_size, rank)
    return list(indices[lo:hi])


def gather_token_rows(local_rows: List[List[int]], group=None) -> List[List[int]]:
    """All ranks' token rows in rank order (rows are ragged: EOT-trimmed id lists)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return list(local_rows)
    out: List = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, local_rows, group=group)
    return [row for part in out for row in part]


def allreduce_loss_stats(sum_ce: torch.Tensor, n_valid: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(sum of masked CE, valid-token count) summed over ranks: loss = sum / max(count, 1)."""
    import torch.distributed as dist

    both = torch.stack([sum_ce.reshape(()).float(), n_valid.reshape(()).float()])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
    return both[0], both[1]


def bucketed(named: Dict[str, torch.Tensor], bucket_bytes: int = 64 << 20) -> Iterable[List[str]]:
    """Group gradient tensors (insertion order = decoder block order) into buckets of about
    ``bucket_bytes``: one RCCL call per bucket keeps each xGMI ring transfer large (7 links x ~153 GB/s,
    per-link bound) while letting a bucket start as soon as its block's backward is done."""
    cur, size = [], 0
    for name, t in named.items():
        nbytes = t.numel() * t.element_size()
        if cur and size + nbytes > bucket_bytes:
            yield cur
            cur, size = [], 0
        cur.append(name)
        size += nbytes
    if cur:
        yield cur


def allreduce_grads(grads: Dict[str, torch.Tensor], bucket_bytes: int = 64 << 20, group=None) -> None:
    """In-place SUM all-reduce of every gradient, one flat buffer per bucket."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for names in bucketed(grads, bucket_bytes):
        flat = torch.cat([grads[n].reshape(-1) for n in names])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for n in names:
            k = grads[n].numel()
            grads[n].copy_(flat[off:off + k].view_as(grads[n]))
            off += k
