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


def usable_host_cores() -> int:
    """cores this process may really use: affinity mask and cgroup quota (os.cpu_count() reports the whole host)"""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def host_threads_per_rank(cores: int = 0, cap: int = 16) -> int:
    """torch CPU threads for THIS rank: the usable cores divided by the ranks that share the host (LOCAL_WORLD_SIZE as set by
    torch.distributed.run, else WORLD_SIZE -- single-node jobs), at most ``cap``.  One process per GPU means eight processes
    per node: each asking for every core (8 x 16 threads) would starve the Python threads that enqueue 63 graph launches per
    pass.  At least 1."""
    import os

    cores = cores or usable_host_cores()
    local = 1
    for key in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        v = os.environ.get(key)
        if v and v.isdigit() and int(v) >= 1:
            local = int(v)
            break
    return max(1, min(cap, cores // local))


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


def require_even_shards(n_items: int, world_size: int) -> int:
    """Items per rank of a data-parallel training batch.  The batch must split evenly: with ceil(n/world) shards the last
    ranks of e.g. 12 clips on 8 GPUs would get nothing, fail to build a batch and leave the step's collectives mismatched."""
    if world_size < 1 or n_items < world_size or n_items % world_size != 0:
        lo, hi = max(world_size, n_items // world_size * world_size), (n_items // world_size + 1) * world_size
        raise ValueError(f"batch size {n_items} must be a positive multiple of the {world_size} data-parallel ranks (e.g. {lo} or {hi})")
    return n_items // world_size


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


def _collective_device(group=None) -> torch.device:
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def agree_on_step(token_width: int, failed: bool = False, group=None) -> int:
    """One MAX all-reduce per training step that carries (a) whether ANY rank failed to build its batch and (b) the widest
    token matrix.  Returns the global width, or -1 when some rank failed -- every rank gets the same answer, so all of
    them leave the step loop together instead of one rank abandoning the others inside the gradient all-reduce."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return -1 if failed else int(token_width)
    t = torch.tensor([1 if failed else 0, int(token_width)], dtype=torch.int64, device=_collective_device(group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    flag, width = (int(v) for v in t.tolist())
    return -1 if flag else width


class SegmentReducer:
    """Asynchronous SUM all-reduce of finished segments of ONE flat gradient buffer.

    The backward pass of the decoder finishes its gradient segments from the back ([final ln], block L-1, ..., block 0,
    [embeddings]); each is handed to ``reduce(lo, hi)`` as soon as it is complete and travels over RCCL while the earlier
    blocks are still computing.  Segments are large and contiguous (one decoder block = 7.1 M .. 28 M floats), which is
    what a ring over point-to-point xGMI links wants.  ``wait()`` joins them in issue order.  Single process: no-ops."""

    def __init__(self, flat: torch.Tensor, group=None):
        self.flat, self.group, self.pending = flat, group, []
        self.active = world()[1] > 1

    def reduce(self, lo: int, hi: int) -> None:
        if self.active and hi > lo:
            import torch.distributed as dist

            self.pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self) -> int:
        n = len(self.pending)
        for work in self.pending:
            work.wait()
        self.pending = []
        return n
