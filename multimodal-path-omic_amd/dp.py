"""Process-per-GPU data parallelism for slide windows.

Slides are independent units, so the only exchange step is ONE all-reduce of the flat fp32 gradient
buffer per optimiser step (RCCL over xGMI via torch.distributed backend 'nccl'; 'gloo' on CPU for
tests).  Parameter .grad tensors are views into one contiguous buffer, so backward accumulates
straight into the bucket and the collective needs no packing.  The reference's nn.DataParallel
(models/mcat/main.py:267-268) is not reproduced: with batch_size=1 it never used more than one GPU.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


class FlatGradBucket:
    """One contiguous fp32 gradient buffer; every parameter's .grad is a view into it."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        """Sum over ranks, divide by world size: with per-rank 1/grad_acc_step loss scaling the update
        equals the reference's accumulation over world_size * grad_acc_step slides."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))


def assign_slides(lengths: Sequence[int], world_size: int) -> "List[List[int]]":
    """Length-aware split of one accumulation window across ranks: greedy longest-first bin packing
    on the patch count (step time = slowest rank; SURVEY.md section 7, hard part 7).  Deterministic;
    every rank computes the same assignment.  Returns slide indices per rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    loads = [0] * world_size
    out: "List[List[int]]" = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += lengths[i]
    for r in range(world_size):
        out[r].sort()
    return out
