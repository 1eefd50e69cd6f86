"""Caller-side semantics of the reference's train()/validate() loops (row H9 of SURVEY.md section
8(a); models/mcat/main.py:19-155), restated for window-batched execution: `ces` loss, risk score,
gradient accumulation over `grad_acc_step` slides, Harrell's C-index.  No per-slide host sync
(the reference's loss.item() at main.py:49 is exactly what this harness must not do)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .ops import BagBatch


def ces_loss(hazards, survs, label, censorship, alpha: float = 0.75, eps: float = 1e-7, reduction: str = "mean"):
    """CrossEntropySurvivalLoss (models/loss.py:5-28) for B slides at once: hazards/survs (B,C),
    label (B,) int64, censorship (B,) float.  reduction 'mean' | 'sum' | 'none' over slides."""
    y = label.view(-1, 1).long()
    c = censorship.view(-1, 1).float()
    s_pad = torch.cat([torch.ones_like(c), survs], 1)
    reg = -(1 - c) * (torch.log(torch.gather(s_pad, 1, y).clamp(min=eps))
                      + torch.log(torch.gather(hazards, 1, y).clamp(min=eps)))
    s_y = torch.gather(survs, 1, y).clamp(min=eps)
    ce = -(c * torch.log(s_y) + (1 - c) * torch.log(1 - s_y))
    loss = ((1 - alpha) * ce + alpha * reg).view(-1)
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def risk_score(survs):
    """risk = -sum_j survs_j (models/mcat/main.py:56)."""
    return -survs.sum(dim=1)


def concordance_index_censored(event, time, risk, tied_tol: float = 1e-8) -> float:
    """Harrell's C with scikit-survival's conventions (the reference calls
    sksurv.metrics.concordance_index_censored, models/mcat/main.py:81): comparable pairs need an
    event at the earlier time, or equal times with the other subject censored; risk ties within
    tied_tol count one half.  Vectorised O(n^2) numpy."""
    event = np.asarray(event, dtype=bool)
    time = np.asarray(time, dtype=np.float64)
    risk = np.asarray(risk, dtype=np.float64)
    later = (time[None, :] > time[:, None]) | ((time[None, :] == time[:, None]) & ~event[None, :])
    comparable = later & event[:, None]
    np.fill_diagonal(comparable, False)
    den = comparable.sum()
    if den == 0:
        raise ValueError("no comparable pairs")
    diff = risk[:, None] - risk[None, :]
    ties = np.abs(diff) <= tied_tol
    num = ((diff > 0) & ~ties & comparable).sum() + 0.5 * (ties & comparable).sum()
    return float(num) / float(den)


def make_window(slides: Sequence[dict], device, bag_dtype=torch.float32):
    """List of slide dicts (synthetic.make_cohort layout) -> (BagBatch, omics per group (B,d_i), labels, censorship)."""
    bags = BagBatch.from_list([s["wsi"].to(device=device, dtype=bag_dtype, non_blocking=True) for s in slides])
    n_groups = len(slides[0]["omics"])
    omics = [torch.stack([s["omics"][i] for s in slides]).to(device, non_blocking=True) for i in range(n_groups)]
    labels = torch.tensor([s["survival_class"] for s in slides], dtype=torch.int64).to(device, non_blocking=True)
    cens = torch.tensor([float(s["censorship"]) for s in slides]).to(device, non_blocking=True)
    return bags, omics, labels, cens


def train_window(model, bags: BagBatch, omics, labels, cens, grad_acc_step: int, loss: str = "ces", lambda_reg: float = 0.01):
    """Forward + backward of one window; gradients ACCUMULATE into .grad with the reference's
    1/grad_acc_step scaling per slide (models/mcat/main.py:69-70).  Returns (per-slide loss, risk) tensors
    on the device -- no host sync.  loss: 'ces' (models/loss.py:5-28) or 'cesar' (:88-101: ces + lambda_reg * ||A_b||_2 of
    the slide's co-attention map, models/nacagat/main.py:49-50)."""
    from . import ops
    if loss == "cesar":
        hazards, survs, _, att = model.forward_window(bags, omics, inference=True)    # the map is an output here
        per_slide, risk = ops.ces_loss(hazards, survs, labels, cens)
        per_slide = per_slide + lambda_reg * ops.map_block_norm(att["coattn"])
    elif loss == "ces":
        if getattr(model, "fusion", None) == "concat":
            # head, loss and the backward of both in one launch: the loss gradient is known before the forward
            w = _slide_weights(bags.n_slides, grad_acc_step, labels.device)
            _, _, _, att = model.forward_window(bags, omics, ces_targets=(labels, cens, w))
            per_slide, risk = att["loss"], att["risk"]
            per_slide.backward(w)
            return per_slide.detach(), risk
        hazards, survs, _, _ = model.forward_window(bags, omics)
        per_slide, risk = ops.ces_loss(hazards, survs, labels, cens)              # one HIP launch each way
    else:
        raise ValueError(f"loss '{loss}' is not built (ces | cesar)")
    # d(sum(loss) / grad_acc_step) / d(loss_b) = 1 / grad_acc_step: hand it over as a cached constant instead of
    # building the sum / div graph (five tiny launches per window)
    per_slide.backward(_slide_weights(per_slide.numel(), grad_acc_step, per_slide.device))
    return per_slide.detach(), risk


_slide_weight_cache = {}


def _slide_weights(n: int, grad_acc_step: int, device) -> torch.Tensor:
    key = (n, grad_acc_step, str(device))
    w = _slide_weight_cache.get(key)
    if w is None:
        w = _slide_weight_cache[key] = torch.full((n,), 1.0 / grad_acc_step, device=device, dtype=torch.float32)
    return w


class GraphedWindowStep:
    """One window step (forward, `ces` loss, backward, gradients into the flat bucket, optionally the Adam
    update) captured ONCE into a HIP graph and replayed: ~500 launches per window cost one graph launch on
    the host instead of ~4 ms of Python / launch overhead.

    Static inputs: the window's tensors are resident and fixed (one captured graph per resident window).
    Frozen-at-capture values that must change per step live on the device: the dropout epoch (ops.set_rng_epoch,
    bumped inside the graph) and Adam's step count (dp.FlatAdam.t_dev).  With world_size > 1 leave the optimiser
    out (`opt=None`): replay, then all-reduce the bucket and step eagerly.
    """

    def __init__(self, model, bucket, window, grad_acc_step: int, opt=None, warmup: int = 2, pool=None,
                 split_patch_grad: bool = False, prime: bool = True):
        """split_patch_grad (data-parallel steps, opt=None): the patch layer's weight gradient -- a 0.3 ms GEMM nobody
        downstream waits for -- is captured into a SECOND graph, `replay_tail()`.  The caller replays the main graph,
        starts the all-reduce of every other gradient (bucket.all_reduce_mean_async(lo=head)), replays the tail while
        that collective runs, then reduces the head slice: the exchange step hides behind compute."""
        from . import ops
        self.model, self.bucket, self.opt = model, bucket, opt
        self.window, self.acc = window, grad_acc_step
        self.split = bool(split_patch_grad)
        if self.split and opt is not None:
            raise ValueError("split_patch_grad is for steps whose optimiser runs after an all-reduce (opt=None)")
        self.tail_graph = None
        window[0].plan()                      # the work plan's H2D copy must not happen inside the capture
        dev = bucket.flat.device
        if ops._rng_epoch_tensor is None:
            ops.set_rng_epoch(torch.zeros(1, dtype=torch.int64, device=dev))
        self.epoch = ops._rng_epoch_tensor
        # warm-up runs execute real steps (allocator, lazy initialisation): construction must not train the model, so the
        # optimiser state they touch is put back afterwards (the capture itself executes nothing)
        keep = None if opt is None else [t.clone() for t in (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.t_dev)]
        keep_epoch = self.epoch.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body()
            if keep is not None:
                for t, k in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.t_dev), keep):
                    t.copy_(k)
            self.epoch.copy_(keep_epoch)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads (RCCL's watchdog under torch.distributed) may issue HIP calls meanwhile
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local"):
            self.loss, self.risk = self._body(flush=not self.split)
        if self.split:
            self._held = list(ops._deferred_patch)          # keep the queued operands alive between the two graphs
            self.tail_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.tail_graph, pool=self.graph.pool(), capture_error_mode="thread_local"):
                ops.flush_patch_weight_grads()
        # The FIRST replay of a captured graph pays for its upload (2-4 ms against a 1.2 ms step, measured): pay it here,
        # with the state it touches put back, so that a caller's first step is a step like any other.
        if prime:
            grads = bucket.flat.clone()
            for _ in range(2):
                self.graph.replay()
                if self.tail_graph is not None:
                    self.tail_graph.replay()
            if keep is not None:
                for t, k in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.t_dev), keep):
                    t.copy_(k)
            self.epoch.copy_(keep_epoch)
            bucket.flat.copy_(grads)
            torch.cuda.current_stream(dev).synchronize()

    def _body(self, flush: bool = True):
        from . import ops
        ops.bump_step_counters(self.epoch, self.opt.t_dev if self.opt is not None else None)   # one launch for both
        self.bucket.begin()
        bags, omics, labels, cens = self.window
        ops.defer_patch_weight_grad = self.split
        try:
            out = train_window(self.model, bags, omics, labels, cens, self.acc)
        finally:
            ops.defer_patch_weight_grad = False
        self.bucket.finish()
        if self.split and flush:
            ops.flush_patch_weight_grads()
        if self.opt is not None:
            self.opt.step(bump=False)
        return out

    def head_numel(self) -> int:
        """Number of leading bucket elements that only replay_tail() writes (the patch layer's weight)."""
        return self.bucket.head_numel(self.model.H[0].weight)

    def replay_tail(self):
        if self.tail_graph is not None:
            self.tail_graph.replay()

    def pool(self):
        return self.graph.pool()

    def __call__(self):
        self.graph.replay()
        return self.loss, self.risk
