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


SLICE_ALIGN = 64        # elements: every parameter's slice starts on a 256-byte boundary of the flat buffers


def _aligned_offsets(params):
    """Start offset of every parameter in the flat buffers and their total length.  Slices are 256-byte aligned: the
    GEMM kernels read weights with 16-byte loads, and one 1-element bias (attention_c.bias) packed tight would leave
    every parameter behind it 4-byte aligned, i.e. on the general (guarded) GEMM body."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + SLICE_ALIGN - 1) // SLICE_ALIGN * SLICE_ALIGN
    return offs, off


class FlatGradBucket:
    """One contiguous fp32 gradient buffer; every parameter owns a slice (`p._mpo_grad_view`, 256-byte aligned,
    `offsets[i]`; the padding between slices stays zero).

    Per window:  begin() -> forward/backward -> finish().  begin() unsets every `p.grad`, so the HIP
    backward entries write their parameter gradients straight into the slices (ops.grad_out) and autograd
    adopts those views as `.grad` without an accumulate kernel; finish() copies in the few gradients that
    came from stock torch ops and zero-fills slices of unused parameters.  A second backward before the
    optimiser step accumulates in place into the same slices."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, n = _aligned_offsets(self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            p._mpo_grad_view = self.flat[off:off + p.numel()].view_as(p)
            p.grad = p._mpo_grad_view.view(p.shape)

    def begin(self):
        for p in self.params:
            p.grad = None
            p._mpo_slice_taken = False

    def finish(self):
        for p in self.params:
            view = p._mpo_grad_view
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            else:
                continue
            p.grad = view.view(p.shape)

    def head_numel(self, param) -> int:
        """Number of leading bucket elements owned by `param`, which must be the FIRST parameter of the bucket (the
        split exchange reduces flat[head:] while the patch layer's weight gradient, flat[:head], is still being
        computed).  Raises if the layout assumption does not hold."""
        if param._mpo_grad_view.data_ptr() != self.flat.data_ptr():
            raise RuntimeError("split exchange: the patch layer's weight (H.0.weight) must lead the gradient bucket")
        return param.numel()

    def zero(self):
        self.flat.zero_()
        for p in self.params:
            p.grad = p._mpo_grad_view.view(p.shape)

    def all_reduce_mean(self, group=None):
        """Sum over ranks, divide by world size: with per-rank 1/grad_acc_step loss scaling the update
        equals the reference's accumulation over world_size * grad_acc_step slides."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == "nccl":            # RCCL averages inside the collective: no extra 16 MB pass
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                self.flat.div_(dist.get_world_size(group))


    def all_reduce_mean_async(self, lo: int = 0, hi: int = None, group=None):
        """Start the mean all-reduce of flat[lo:hi] and return a handle whose wait() orders the current stream behind it
        (None when there is nothing to reduce): lets the caller keep computing into OTHER slices of the bucket meanwhile."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return None
        view = self.flat[lo:hi]
        if dist.get_backend(group) == "nccl":
            return dist.all_reduce(view, op=dist.ReduceOp.AVG, group=group, async_op=True)
        work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group, async_op=True)
        world = dist.get_world_size(group)

        class _Then:
            def wait(self_inner):
                work.wait()
                view.div_(world)
        return _Then()


class FlatAdam:
    """torch.optim.Adam(lr, betas, eps, weight_decay) (the reference's default optimiser,
    models/mcat/main.py:284-300) as ONE HIP kernel over flat buffers: parameters are re-pointed at slices of
    a flat fp32 buffer laid out like the gradient bucket; exp_avg / exp_avg_sq are flat too."""

    def __init__(self, bucket: FlatGradBucket, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.bucket, self.lr, self.betas, self.eps, self.wd = bucket, lr, betas, eps, weight_decay
        self.flat_p = torch.zeros_like(bucket.flat)
        for p, off in zip(bucket.params, bucket.offsets):
            sl = self.flat_p[off:off + p.numel()].view_as(p)
            sl.copy_(p.data)
            p.data = sl
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=self.flat_p.device)   # step count, device-resident

    def step(self, bump: bool = True):
        """One update.  The step count is incremented and read on the device, so the call can sit inside a
        captured HIP graph and still apply the right bias correction on every replay.  bump=False: the caller has
        advanced t_dev already (ops.bump_step_counters, one launch for it and the dropout epoch)."""
        from . import _lib as L
        if bump:
            self.t_dev += 1
        b = self.bucket
        L.check(L.lib().mpo_adam_step_flat(L.ptr(self.flat_p), L.ptr(b.flat), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                           self.flat_p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                           0, L.ptr(self.t_dev), L.stream_of(self.flat_p)), "mpo_adam_step_flat")


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
