"""torch.autograd.Function wrappers around the C-ABI entries of libmpo_hip.so.

One Function per kernel family of SURVEY.md section 2 (K1..K6).  Each forward/backward is ONE call
across the ABI; the sequence of kernel launches lives in csrc/capi.hip.  Buffers (outputs, saved
tensors, workspaces) are torch allocations; the library keeps nothing.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib as L


# Upper bound on the workgroups of a window's work plan (None: one per CU).  A measurement knob (tools/gpu_probe_overlap.py):
# persistent bag kernels on fewer CUs leave the rest to whatever another stream launches.
plan_workgroups = None
# Workgroups of the patch layer's weight-gradient kernel (None / 0: one per CU).  A data-parallel step sets 224: the kernel is
# persistent and owns its CUs outright, so the gradient all-reduce that runs beside it on another stream needs CUs of its own
# (measured with tools/gpu_probe_dp_overlap.py).
wgrad_workgroups = None


@dataclass
class BagBatch:
    """A window of slides' bags concatenated along rows ("ragged").

    data  (total_rows, E) fp32 or bf16;  cu  device int32 (n_slides+1) row offsets;
    lengths  host list of M_b (grid sizing needs max and total on the host, without a sync).
    """
    data: torch.Tensor
    cu: torch.Tensor
    lengths: List[int]
    _plan: object = None

    def __post_init__(self):
        # an empty bag has no softmax (the reference's nn.MultiheadAttention returns NaN for it): refuse it here, before a
        # workgroup-less slide reaches the kernels
        if len(self.lengths) == 0 or any(int(m) <= 0 for m in self.lengths):
            raise ValueError(f"every slide of a window needs at least one patch row (lengths {list(self.lengths)[:8]}...)")
        if int(self.data.shape[0]) != sum(int(m) for m in self.lengths):
            raise ValueError(f"{int(self.data.shape[0])} rows for lengths summing to {sum(self.lengths)}")

    def plan(self):
        """Work plan of the bag passes (mpo_bag_plan): rows per workgroup uniform over the whole window, so every
        slide gets row ranges in proportion to its length.  Built once per window (one small H2D copy), kept alive
        with the batch; returns a ctypes pointer for the C ABI."""
        if self._plan is None:
            target = L.lib().mpo_coattn_target_workgroups()
            if plan_workgroups is not None:             # (fewer row ranges than CUs: leaves CUs to kernels of another stream)
                target = max(len(self.lengths), min(target, int(plan_workgroups)))
            rpw = -(-self.total_rows // target)
            rpw = max(32, -(-rpw // 32) * 32)
            # every slide rounds its workgroup count up: grow the range until the WHOLE window fits one workgroup per CU
            # (a window with more slides than CUs cannot).  A second, nearly empty round of workgroups costs a 50-us bag
            # pass little, but doubles the 0.4-ms persistent patch-layer kernel (measured on the 2k-30k windows, r02).
            while len(self.lengths) <= target and sum(-(-m // rpw) for m in self.lengths) > target:
                rpw += 32
            starts = [0]
            for m in self.lengths:
                starts.append(starts[-1] + -(-m // rpw))
            wg = torch.tensor(starts, dtype=torch.int32).to(self.data.device, non_blocking=True)
            c = L.BagPlanC(L.ptr(wg), starts[-1], rpw)
            self._plan = (wg, c)
        return ctypes.byref(self._plan[1])

    @property
    def n_slides(self):
        return len(self.lengths)

    @property
    def total_rows(self):
        return int(self.data.shape[0])

    @property
    def max_rows(self):
        return max(self.lengths)

    @staticmethod
    def from_list(bags: "List[torch.Tensor]") -> "BagBatch":
        lengths = [int(b.shape[0]) for b in bags]
        data = bags[0] if len(bags) == 1 else torch.cat(bags, 0)
        return BagBatch(data.contiguous(), make_cu(lengths, data.device), lengths)

    @staticmethod
    def from_lengths(data: torch.Tensor, lengths: "List[int]") -> "BagBatch":
        """Rows already concatenated (e.g. a window slab shipped by ingest.WindowFeeder)."""
        lengths = [int(m) for m in lengths]
        if int(data.shape[0]) != sum(lengths):
            raise ValueError(f"{int(data.shape[0])} rows for lengths summing to {sum(lengths)}")
        return BagBatch(data, make_cu(lengths, data.device), lengths)

    def with_data(self, data: torch.Tensor) -> "BagBatch":
        assert data.shape[0] == self.total_rows
        return BagBatch(data, self.cu, self.lengths, self._plan)

    def split_map(self, flat_map: torch.Tensor, n_q: int) -> "List[torch.Tensor]":
        """Ragged attention map -> list of (n_q, M_b) views (slide b starts at n_q * cu[b]).  The list also carries the
        flat tensor (`.flat`, `.batch`, `.n_q`) for window-level consumers such as the attention-regularised loss."""
        out, off = RaggedMaps(), 0
        out.flat, out.batch, out.n_q = flat_map, self, n_q
        for m in self.lengths:
            out.append(flat_map[n_q * off:n_q * (off + m)].view(n_q, m))
            off += m
        return out


class RaggedMaps(list):
    """list of per-slide (n_q, M_b) views + the flat ragged tensor they alias."""
    flat: Optional[torch.Tensor] = None
    batch: Optional["BagBatch"] = None
    n_q: int = 0


def make_cu(lengths, device):
    cu = [0]
    for m in lengths:
        if m < 1:
            raise ValueError("every slide needs at least one patch")
        cu.append(cu[-1] + int(m))
    return torch.tensor(cu, dtype=torch.int32).to(device, non_blocking=True)


def grad_out(p):
    """Where a Function's backward writes the gradient of parameter `p`.

    With dp.FlatGradBucket every parameter owns a slice of ONE flat fp32 gradient buffer.  While `p.grad`
    is still unset in this window the kernels write straight into that slice and the returned view is
    adopted by autograd as `p.grad` (no accumulate kernel); otherwise a fresh tensor is returned and
    autograd adds it."""
    view = getattr(p, "_mpo_grad_view", None)
    if view is not None and p.grad is None and not getattr(p, "_mpo_slice_taken", False):
        # the slice is handed out ONCE per window (FlatGradBucket.begin() clears the flag): the kernels overwrite their
        # gradient outputs, so a second producer of the same parameter (two forwards before one backward, tied
        # weights) gets its own tensor and autograd adds the two
        p._mpo_slice_taken = True
        return view.view(p.shape)          # a FRESH alias: autograd only steals a gradient nobody else references
    return torch.empty_like(p)


_rng_epoch_tensor = None


def set_rng_epoch(t):
    """Install (or clear with None) the device uint64 scalar the kernels add (x 2^40) to their dropout offsets;
    harness.GraphedWindowStep bumps it inside the captured graph so replays draw fresh masks."""
    global _rng_epoch_tensor
    _rng_epoch_tensor = t


def _epoch():
    return L.ptr(_rng_epoch_tensor) if _rng_epoch_tensor is not None else None


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------ linear
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the fp32 MFMA GEMM (stands in for torch.nn.functional.linear on the small-row tail)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act: str = "none"):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = torch.empty(x2.shape[0], weight.shape[0], device=x.device, dtype=torch.float32)
        L.check(L.lib().mpo_linear_forward(L.ptr(x2), L.ptr(weight), L.ptr(bias), L.ptr(y), x2.shape[0],
                                           weight.shape[1], weight.shape[0], 1.0, L.ACT[act], L.stream_of(x)),
                "mpo_linear_forward")
        ctx.save_for_backward(x2, weight, y)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, y = ctx.saved_tensors
        dy = dy.reshape(-1, dy.shape[-1]).contiguous()
        if ctx.act == "relu":
            dy = dy * (y > 0)
        elif ctx.act == "tanh":
            dy = dy * (1 - y * y)
        elif ctx.act == "sigmoid":
            dy = dy * (y * (1 - y))
        elif ctx.act == "elu":
            dy = dy * torch.where(y > 0, torch.ones_like(y), y + 1)
        dy = dy.contiguous()
        R, I, O = x2.shape[0], weight.shape[1], weight.shape[0]
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None      # (raw patch features take no gradient)
        dw = torch.empty_like(weight)
        db = torch.empty(O, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
        s = L.stream_of(dy)
        if dx is not None:
            L.check(L.lib().mpo_linear_backward_input(L.ptr(dy), L.ptr(weight), L.ptr(dx), R, I, O, 1.0, 0, s),
                    "mpo_linear_backward_input")
        L.check(L.lib().mpo_linear_backward_weight(L.ptr(dy), L.ptr(x2), L.ptr(dw), L.ptr(db), R, I, O, 1.0, s),
                "mpo_linear_backward_weight")
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db, None


def linear(x, weight, bias=None, act="none"):
    return LinearFn.apply(x, weight, bias, act)


# ------------------------------------------------------------------------------------ K1
class CoAttnMCATFn(torch.autograd.Function):
    """MCAT co-attention over a ragged window (models/mcat/mcat.py:97)."""

    @staticmethod
    def forward(ctx, query, bag_data, in_w, in_b, out_w, out_b, batch: BagBatch, need_weights: bool,
                bag_relu_gate: float = 0.0):
        lib = L.lib()
        ctx.set_materialize_grads(False)
        ctx.bag_relu_gate = float(bag_relu_gate)
        ctx.bag_bias = getattr(bag_data, "_mpo_bias_param", None)
        n_slides = batch.n_slides
        R, E = query.shape
        n_q = R // n_slides
        dev = query.device
        query = query.contiguous()
        out = torch.empty(R, E, device=dev, dtype=torch.float32)
        amap = torch.empty(n_q * batch.total_rows, device=dev, dtype=torch.float32) if need_weights else None
        saved = torch.empty(lib.mpo_coattn_saved_floats(n_slides, n_q, E), device=dev, dtype=torch.float32)
        ws = _workspace(lib.mpo_coattn_workspace_bytes(n_slides, n_q, E, batch.max_rows), dev)
        L.check(lib.mpo_coattn_mcat_forward(
            L.ptr(bag_data), L.bag_dtype_code(bag_data), L.ptr(batch.cu), n_slides, batch.total_rows, batch.max_rows,
            L.ptr(query), n_q, E, L.ptr(in_w), L.ptr(in_b), L.ptr(out_w), L.ptr(out_b),
            L.ptr(out), L.ptr(amap), L.ptr(saved), batch.plan(), L.ptr(ws), ws.numel(), L.stream_of(query)),
            "mpo_coattn_mcat_forward")
        ctx.save_for_backward(query, bag_data, in_w, out_w, saved, amap)
        ctx.param_refs = (in_w, in_b, out_w, out_b)
        ctx.batch = batch
        ctx.n_q = n_q
        return out, amap          # the map (if any) is differentiable: backward accepts its gradient

    @staticmethod
    def backward(ctx, d_out, d_map):
        lib = L.lib()
        query, bag_data, in_w, out_w, saved, amap = ctx.saved_tensors
        batch, n_q = ctx.batch, ctx.n_q
        R, E = query.shape
        dev = query.device
        d_out = d_out.contiguous() if d_out is not None else torch.zeros(R, E, device=dev)
        if d_map is not None:
            d_map = d_map.contiguous()
        d_query = torch.empty_like(query)
        d_bag = torch.empty_like(bag_data)
        # with the fused gate d_bag IS the patch layer's pre-activation gradient: its column sums (that layer's bias
        # gradient) fall out of the kernel's copy-out loop and travel on the tensor to PatchFcFn.backward
        colsum = _bias_grad_slot(ctx.bag_bias, E, dev) if ctx.bag_relu_gate != 0.0 else None
        d_in_w, d_in_b, d_out_w, d_out_b = (grad_out(p) for p in ctx.param_refs)
        ws = _workspace(lib.mpo_coattn_workspace_bytes(batch.n_slides, n_q, E, batch.max_rows), dev)
        L.check(lib.mpo_coattn_mcat_backward(
            L.ptr(bag_data), L.bag_dtype_code(bag_data), L.ptr(batch.cu), batch.n_slides, batch.total_rows,
            batch.max_rows, L.ptr(query), n_q, E, L.ptr(in_w), L.ptr(out_w), L.ptr(saved), L.ptr(amap),
            L.ptr(d_out), L.ptr(d_map), L.ptr(d_query), 0, L.ptr(d_bag), L.ptr(colsum), L.ptr(d_in_w), L.ptr(d_in_b),
            L.ptr(d_out_w), L.ptr(d_out_b), ctx.bag_relu_gate, batch.plan(), L.ptr(ws), ws.numel(), L.stream_of(query)),
            "mpo_coattn_mcat_backward")
        if colsum is not None:
            d_bag._mpo_colsum = colsum
        return d_query, d_bag, d_in_w, d_in_b, d_out_w, d_out_b, None, None, None


def coattn_mcat(query, batch: BagBatch, in_w, in_b, out_w, out_b, need_weights: bool, bag_relu_gate: float = 0.0):
    """query (n_slides*n_q, E) -> (out (n_slides*n_q, E), ragged map or None).
    bag_relu_gate = 1/(1-p) when the bag comes from patch_fc(..., pre_gated_grad=True): d_bag then already
    carries the ReLU/dropout derivative (see include/mpo_hip.h)."""
    return CoAttnMCATFn.apply(query, batch.data, in_w, in_b, out_w, out_b, batch, need_weights, bag_relu_gate)


stats = {"colsum_handoffs": 0}           # counters the tests read to make sure a fused path really ran


def _bias_grad_slot(bag_param, E, dev):
    """Where a co-attention backward writes the column sums of its (pre-gated) d_bag = the bias gradient of the patch
    layer that produced the bag: straight into that bias's slice of the flat gradient bucket when patch_fc() tagged the
    bag with its bias and the slice is still unset (PatchFcFn.backward then finds the data in place and skips its copy),
    else a fresh tensor that travels on d_bag."""
    if bag_param is not None and getattr(bag_param, "_mpo_grad_view", None) is not None and bag_param.grad is None \
            and bag_param.numel() == E and not getattr(bag_param, "_mpo_slice_taken", False):
        return bag_param._mpo_grad_view.view(E)         # (PatchFcFn.backward's grad_out(bias) then claims this slice)
    return torch.empty(E, device=dev, dtype=torch.float32)

# Data-parallel steps split the backward in two: everything except the patch layer's weight gradient (dW_H = g^T X, a
# 0.3 ms library GEMM that nothing downstream waits for) runs first, then the all-reduce of all other gradients is
# started and dW_H is computed WHILE that collective runs (harness.GraphedWindowStep(split_patch_grad=True)).
defer_patch_weight_grad = False
_deferred_patch = []

def flush_patch_weight_grads():
    """Compute the patch-layer weight gradients PatchFcFn.backward queued (into the bucket slices it already returned)."""
    for g, x, dw in _deferred_patch:
        patch_weight_grad(g, x, dw)
    _deferred_patch.clear()


# ------------------------------------------------------------------------------------ patch layer (row H2)
class PatchFcFn(torch.autograd.Function):
    """H_bag = dropout_p(relu(X W_H^T + b)) for a bf16-stored window (models/mcat/mcat.py:24-29,87).

    Forward: one pass of csrc/patch_fc_fwd.hip (embed 128 / 256 / 512: the same kernel, see its header), the dropout mask kept
    in H_bag as zeros.  Backward: the ReLU / dropout derivative (in the consumer's kernel when it can, else one element-wise
    pass) and dW_H = g^T X on csrc/patch_wgrad.hip.  X never needs a gradient (it is data).  Other geometries raise."""

    @staticmethod
    def forward(ctx, x, weight, bias, drop_p: float, pre_gated_grad: bool, batch=None):
        lib = L.lib()
        if not patch_fc_kernel_supported(x, weight):
            raise ValueError(f"patch layer: built for a contiguous bf16 window through Linear(1024, 128 | 256 | 512) "
                             f"(got {x.dtype} {tuple(x.shape)} -> {weight.shape[0]})")
        if batch is None:                   # a bare patch matrix: one slide
            batch = BagBatch(x, make_cu([x.shape[0]], x.device), [x.shape[0]])
        h = torch.empty(x.shape[0], weight.shape[0], device=x.device, dtype=torch.bfloat16)
        seed, off = _reserve(h.numel() // 16 + 2) if drop_p > 0 else (0, 0)
        ws = _workspace(lib.mpo_patch_fc_workspace_bytes(weight.shape[0], weight.shape[1]), x.device)
        L.check(lib.mpo_patch_fc_forward(L.ptr(x), L.ptr(batch.cu), batch.n_slides, batch.total_rows, batch.max_rows,
                                         x.shape[1], L.ptr(weight), L.ptr(bias), weight.shape[0], float(drop_p), seed, off,
                                         _epoch(), L.ptr(h), batch.plan(), L.ptr(ws), ws.numel(), L.stream_of(x)),
                "mpo_patch_fc_forward")
        drop_p = _realised_drop(drop_p) if drop_p > 0 else 0.0
        ctx.save_for_backward(x, h)
        ctx.param_refs = (weight, bias)
        ctx.drop_p, ctx.pre_gated = float(drop_p), bool(pre_gated_grad)
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = L.lib()
        x, h = ctx.saved_tensors
        dh = dh.contiguous()
        if ctx.pre_gated:
            g = dh
        else:
            g = None
        dw, db = (grad_out(p) for p in ctx.param_refs)
        if g is None:                       # ReLU/dropout derivative + the bias gradient (column sums) in one pass
            g = torch.empty_like(dh)
            ws = _workspace(lib.mpo_patch_epilogue_backward_workspace_bytes(g.numel(), g.shape[1]), g.device)
            L.check(lib.mpo_patch_epilogue_backward(L.ptr(h), L.ptr(dh), L.ptr(g), g.numel(), g.shape[1], ctx.drop_p,
                                                    L.ptr(db), L.ptr(ws), ws.numel(), L.stream_of(g)),
                    "mpo_patch_epilogue_backward")
        else:
            ready = getattr(dh, "_mpo_colsum", None)
            if ready is not None and ready.shape == db.shape:
                if ready.data_ptr() != db.data_ptr():       # (already in the bucket slice when patch_fc tagged the bag)
                    db.copy_(ready)         # produced by the co-attention backward kernel while it wrote dh
                stats["colsum_handoffs"] += 1
            else:
                _colsum_two_stage(g, db)
        if defer_patch_weight_grad and getattr(ctx.param_refs[0], "_mpo_grad_view", None) is not None \
                and dw.data_ptr() == ctx.param_refs[0]._mpo_grad_view.data_ptr():
            _deferred_patch.append((g, x, dw))      # dw aliases the bucket slice: filled by flush_patch_weight_grads()
        else:
            patch_weight_grad(g, x, dw)
        return None, dw, db, None, None, None


def _colsum_two_stage(g: torch.Tensor, out: torch.Tensor, block: int = 256) -> torch.Tensor:
    """Column sums of a tall bf16 matrix in fp32: (rows/256, 256, d) -> sum(1) -> sum(0).  One flat torch.sum over
    480k rows runs at 1.7-2.4 TB/s (101-142 us); the blocked form takes 56 us (measured r01)."""
    rows, d = g.shape
    main = rows // block * block
    if main:
        torch.sum(g[:main].view(rows // block, block, d).sum(1, dtype=torch.float32), 0, out=out)
    else:
        out.zero_()
    if main < rows:
        out += g[main:].sum(0, dtype=torch.float32)
    return out


def patch_weight_grad(g: torch.Tensor, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """dW_H = g^T x into `out` (embed, patch_dim) fp32 on csrc/patch_wgrad.hip: bf16 operands, embed 128 / 256 / 512,
    patch_dim 128, 256, 512, 1024 or 2048, any number of rows (4 GiB of patches and more go in row segments).  Anything else raises."""
    e, k = out.shape
    if not (g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and e in (128, 256, 512) and k in (128, 256, 512, 1024, 2048)
            and g.shape == (x.shape[0], e) and x.shape[1] == k
            and g.is_contiguous() and x.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32):
        raise ValueError(f"patch weight gradient: g {g.dtype} {tuple(g.shape)} x {x.dtype} {tuple(x.shape)} -> {tuple(out.shape)} is not a "
                         f"geometry the kernel is built for (contiguous bf16, embed 128/256/512, patch_dim 128/256/512/1024/2048)")
    lib = L.lib()
    ws = _workspace(lib.mpo_patch_weight_grad_workspace_bytes(e, k), g.device)
    wgs = int(wgrad_workgroups or 0)
    if wgs and k != 1024:
        wgs = 0                                        # (the setting is for the patch layer's own gradient, patch_dim 1024)
    L.check(lib.mpo_patch_weight_grad(L.ptr(g), L.ptr(x), g.shape[0], e, k, L.ptr(out), wgs, L.ptr(ws), ws.numel(),
                                      L.stream_of(g)), "mpo_patch_weight_grad")
    return out


class PatchFcF32Fn(torch.autograd.Function):
    """H_bag = dropout_p(relu(X W_H^T + b)) for an fp32-stored window through Linear(1024, 256) (models/mcat/mcat.py:24-29,87),
    hand-written both ways (csrc/patch_fc_f32.hip): products as three bf16 MFMA terms of hi / lo operand splits (fp32
    accumulation, ~4e-6 absolute on H_bag), dropout mask kept in H_bag as zeros; backward = ONE pass that applies the
    ReLU / dropout derivative read off H_bag to the incoming gradient and forms dW_H = g^T X and db_H = colsum(g)."""

    @staticmethod
    def forward(ctx, x, weight, bias, drop_p: float):
        lib = L.lib()
        h = torch.empty(x.shape[0], weight.shape[0], device=x.device, dtype=torch.float32)
        seed, off = _reserve(h.numel() // 16 + 2) if drop_p > 0 else (0, 0)
        ws = _workspace(lib.mpo_patch_fc_f32_workspace_bytes(0), x.device)
        L.check(lib.mpo_patch_fc_f32_forward(L.ptr(x), x.shape[0], x.shape[1], L.ptr(weight), L.ptr(bias), weight.shape[0],
                                             float(drop_p), seed, off, _epoch(), feature_scale(x), L.ptr(h), L.ptr(ws), ws.numel(),
                                             L.stream_of(x)),
                "mpo_patch_fc_f32_forward")
        ctx.save_for_backward(x, h)
        ctx.param_refs = (weight, bias)
        ctx.gate = 1.0 / (1.0 - _realised_drop(drop_p)) if drop_p > 0 else 1.0
        return h

    @staticmethod
    def backward(ctx, dh):
        lib = L.lib()
        x, h = ctx.saved_tensors
        dh = dh.contiguous()
        dw, db = (grad_out(p) for p in ctx.param_refs)
        ws = _workspace(lib.mpo_patch_fc_f32_workspace_bytes(1), x.device)
        L.check(lib.mpo_patch_fc_f32_backward(L.ptr(dh), L.ptr(h), L.ptr(x), x.shape[0], h.shape[1], x.shape[1], ctx.gate, L.ptr(dw),
                                              L.ptr(db), L.ptr(ws), ws.numel(), L.stream_of(x)), "mpo_patch_fc_f32_backward")
        return None, dw, db, None


def feature_scale(x) -> float:
    """The power of two that puts max |x| of an fp32 patch matrix into [2^14, 2^15): the fp16 operand splits of
    mpo_patch_fc_f32_forward then use fp16's range whatever the features' own scale (1e-4 or 1e4).  One reduction over the
    window and one host read, cached on the tensor (a resident window is scanned once; call it -- or one eager forward --
    before capturing a graph)."""
    s = getattr(x, "_mpo_feature_scale", None)
    if s is None:
        if x.numel() == 0:
            return 1.0
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("feature_scale(x) needs one host read: call ops.feature_scale(window.data) before capturing the graph")
        m = float(x.detach().abs().amax())
        s = math.ldexp(1.0, max(-100, min(100, 15 - math.frexp(m)[1]))) if (m > 0.0 and math.isfinite(m)) else 1.0
        try:
            x._mpo_feature_scale = s
        except AttributeError:
            pass
    return s


def patch_fc_f32_supported(x, weight) -> bool:
    """mpo_patch_fc_f32_*: an fp32 window through Linear(1024, 256)."""
    return x.dtype == torch.float32 and x.is_contiguous() and tuple(weight.shape) == (256, 1024)


def patch_fc_f32(x, weight, bias, drop_p: float):
    """H_bag = dropout(relu(x W^T + b)) in fp32 storage on the hand-written kernels (no library GEMM)."""
    return PatchFcF32Fn.apply(x, weight, bias, float(drop_p))


# ------------------------------------------------------------------------------------ row f1: patch layer + K1 in one pass
class PatchCoAttnMCATFn(torch.autograd.Function):
    """H_bag = dropout_p(relu(X W_H^T + b_H)) AND MCAT's co-attention over it (models/mcat/mcat.py:24-29,87,97) with ONE
    pass over the raw bf16 patch matrix: mpo_patch_coattn_mcat_forward.  H_bag is written once (bf16) and kept for the
    backward, which is K1's backward pass (d_bag arrives already multiplied by the ReLU / dropout derivative, with its
    column sums = the patch layer's bias gradient) followed by the patch layer's weight gradient g^T X."""

    @staticmethod
    def forward(ctx, x, patch_w, patch_b, query, in_w, in_b, out_w, out_b, batch: BagBatch, need_weights: bool, drop_p: float,
                tokens: "TokenPair | None" = None):
        lib = L.lib()
        ctx.set_materialize_grads(False)
        n_slides = batch.n_slides
        R, E = query.shape
        n_q = R // n_slides
        dev, T = query.device, batch.total_rows
        query = query.contiguous()
        h_bag = torch.empty(T, E, device=dev, dtype=torch.bfloat16)
        out = tokens.slot(0, (R, E)) if tokens is not None else torch.empty(R, E, device=dev, dtype=torch.float32)
        amap = torch.empty(n_q * T, device=dev, dtype=torch.float32) if need_weights else None
        saved = torch.empty(lib.mpo_coattn_saved_floats(n_slides, n_q, E), device=dev, dtype=torch.float32)
        ws = _workspace(lib.mpo_patch_coattn_workspace_bytes(n_slides, n_q, E, x.shape[1]), dev)
        seed, off = _reserve(T * E // 16 + 2) if drop_p > 0 else (0, 0)
        L.check(lib.mpo_patch_coattn_mcat_forward(
            L.ptr(x), L.ptr(batch.cu), n_slides, T, batch.max_rows, x.shape[1], L.ptr(patch_w), L.ptr(patch_b), float(drop_p),
            seed, off, _epoch(), L.ptr(query), n_q, E, L.ptr(in_w), L.ptr(in_b), L.ptr(out_w), L.ptr(out_b),
            L.ptr(h_bag), L.ptr(out), L.ptr(amap), L.ptr(saved), batch.plan(), L.ptr(ws), ws.numel(), L.stream_of(query)),
            "mpo_patch_coattn_mcat_forward")
        ctx.save_for_backward(x, h_bag, query, in_w, out_w, saved, amap)
        ctx.param_refs = (patch_w, patch_b, in_w, in_b, out_w, out_b)
        ctx.batch, ctx.n_q = batch, n_q
        ctx.gate = 1.0 / (1.0 - _realised_drop(drop_p)) if drop_p > 0 else 1.0
        ctx.mark_non_differentiable(h_bag)
        # the query handed on to its second consumer (the omic branch's tokens): its gradient then arrives HERE and is
        # folded into the last GEMM of the backward (d_query += ...) instead of costing autograd an add launch
        return out, amap, h_bag, query.view_as(query)

    @staticmethod
    def backward(ctx, d_out, d_map, _d_h, d_qpass):
        lib = L.lib()
        x, h_bag, query, in_w, out_w, saved, amap = ctx.saved_tensors
        batch, n_q = ctx.batch, ctx.n_q
        R, E = query.shape
        dev = query.device
        d_out = d_out.contiguous() if d_out is not None else torch.zeros(R, E, device=dev)
        d_map = d_map.contiguous() if d_map is not None else None
        accumulate = d_qpass is not None
        d_query = d_qpass.contiguous() if accumulate else torch.empty_like(query)    # (in place on the incoming gradient)
        g = torch.empty_like(h_bag)               # d(pre-activation of the patch layer): ReLU/dropout derivative applied in-kernel
        patch_w, patch_b, p_in_w, p_in_b, p_out_w, p_out_b = ctx.param_refs
        d_pw, d_pb = grad_out(patch_w), grad_out(patch_b)
        d_in_w, d_in_b, d_out_w, d_out_b = (grad_out(p) for p in (p_in_w, p_in_b, p_out_w, p_out_b))
        ws = _workspace(lib.mpo_coattn_workspace_bytes(batch.n_slides, n_q, E, batch.max_rows), dev)
        L.check(lib.mpo_coattn_mcat_backward(
            L.ptr(h_bag), L.MPO_BF16, L.ptr(batch.cu), batch.n_slides, batch.total_rows, batch.max_rows, L.ptr(query), n_q, E,
            L.ptr(in_w), L.ptr(out_w), L.ptr(saved), L.ptr(amap), L.ptr(d_out), L.ptr(d_map), L.ptr(d_query), int(accumulate),
            L.ptr(g), L.ptr(d_pb), L.ptr(d_in_w), L.ptr(d_in_b), L.ptr(d_out_w), L.ptr(d_out_b), ctx.gate, batch.plan(),
            L.ptr(ws), ws.numel(), L.stream_of(query)), "mpo_coattn_mcat_backward")
        stats["colsum_handoffs"] += 1
        if defer_patch_weight_grad and getattr(patch_w, "_mpo_grad_view", None) is not None \
                and d_pw.data_ptr() == patch_w._mpo_grad_view.data_ptr():
            _deferred_patch.append((g, x, d_pw))   # filled by flush_patch_weight_grads() (data-parallel split exchange)
        else:
            patch_weight_grad(g, x, d_pw)
        return None, d_pw, d_pb, d_query, d_in_w, d_in_b, d_out_w, d_out_b, None, None, None, None


class TokenPair:
    """(2, rows, d) fp32 buffer for the two token sets the branch-batched tail consumes (co-attention output | omic
    tokens).  Producers write their half in place (`slot`), `stack` hands the whole buffer to the encoder: the
    torch.stack copy and its backward disappear from the step."""

    def __init__(self, rows: int, d: int, device):
        self.buf = torch.empty(2, rows, d, device=device, dtype=torch.float32)

    def slot(self, i: int, shape):
        return self.buf[i].view(*shape)

    def stack(self, first, second):
        for i, t in enumerate((first, second)):
            if t.data_ptr() != self.buf[i].data_ptr() or t.numel() != self.buf[i].numel() or not t.is_contiguous():
                raise ValueError("TokenPair.stack: the halves must be the tensors produced into slot(0) and slot(1)")
        return _TokenPairFn.apply(first, second, self)


class _TokenPairFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second, pair):
        ctx.shapes = (first.shape, second.shape)
        return pair.buf.view_as(pair.buf)

    @staticmethod
    def backward(ctx, d):
        return d[0].view(ctx.shapes[0]), d[1].view(ctx.shapes[1]), None


def _realised_drop(p: float) -> float:
    """The fused kernel draws 8 random bits per element: its drop probability is round(256 p) / 256."""
    return float(int(p * 256.0 + 0.5)) / 256.0


def patch_coattn_mcat(x_bf16, batch: BagBatch, patch_w, patch_b, drop_p: float, query, in_w, in_b, out_w, out_b,
                      need_weights: bool, tokens: "TokenPair | None" = None):
    """-> (out (n_slides*n_q, E), ragged map | None, H_bag (rows, E) bf16, not differentiable, query handed on).
    tokens: `out` is produced into tokens.slot(0).  Use the returned query (not the argument) for the query's other
    consumer: its gradient is then folded into this op's backward."""
    return PatchCoAttnMCATFn.apply(x_bf16, patch_w, patch_b, query, in_w, in_b, out_w, out_b, batch, need_weights,
                                   float(drop_p), tokens)


def fused_patch_coattn_supported(x, embed: int, n_q: int) -> bool:
    """mpo_patch_coattn_mcat_forward is built for a bf16 window, 1024 -> 256, at most 8 omic queries."""
    return x.dtype == torch.bfloat16 and x.shape[1] == 1024 and embed == 256 and n_q <= 8


def patch_fc_kernel_supported(x, weight) -> bool:
    """mpo_patch_fc_forward: a bf16 window through Linear(1024, 128 | 256 | 512)."""
    return x.dtype == torch.bfloat16 and x.is_contiguous() and weight.shape[0] in (128, 256, 512) and weight.shape[1] == 1024


def patch_fc(x_bf16, weight, bias, drop_p: float, pre_gated_grad: bool = False, batch: "BagBatch | None" = None):
    """H_bag = dropout(relu(x W^T + b)), bf16.  batch (the window's row offsets / work plan): lets Linear(1024, 256) run as
    one hand-written pass; the tensor then carries `_mpo_keep_scale` = 1 / (1 - realised dropout rate) for consumers that
    apply the ReLU / dropout derivative themselves."""
    h = PatchFcFn.apply(x_bf16, weight, bias, drop_p, pre_gated_grad, batch)
    if pre_gated_grad:
        h._mpo_bias_param = bias          # lets the consumer's backward write this layer's bias gradient in place
    if drop_p > 0:
        h._mpo_keep_scale = 1.0 / (1.0 - _realised_drop(drop_p))
    return h


# ------------------------------------------------------------------------------------ tail (6 x d tokens per slide)
import torch.nn.functional as F  # noqa: E402


def _reserve(span: int):
    """Reserve `span` counters of the dropout generator for one C-ABI call's streams."""
    global _rng_calls
    seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    off = _rng_calls
    _rng_calls += int(span) + 1
    return seed, off


def gated_scores(x, wa, ba, wb, bb, wc, bc, drop_p: float):
    """AttentionNetGated scores alone (models/blocks.py:42-47), for the module's stand-alone forward():
    HIP GEMMs (fused tanh / sigmoid) + the element-wise product.  x (..., L, D) -> (..., L, n_classes).
    The model path uses gated_pool(), which fuses scorer, pooling and rho in one C-ABI call."""
    a = linear(x, wa, ba, "tanh")
    b = linear(x, wb, bb, "sigmoid")
    if drop_p > 0.0:
        a = F.dropout(a, drop_p, True)
        b = F.dropout(b, drop_p, True)
    return linear(a * b, wc, bc)


class CagFn(torch.autograd.Function):
    """K3: ContextualAttentionGate (models/blocks.py:247-253), one C-ABI call each way.
    residual / dest: the caller's  residual + C  (models/blocks.py:110) is produced into `dest` by the forward's last launch (no
    element-wise pass).  The second output hands q on to its other consumers: their gradient then arrives HERE and the
    backward's d_q product accumulates onto it (no gradient-add launch)."""

    @staticmethod
    def forward(ctx, q, q_hat, residual, dest, *params):
        lib = L.lib()
        ctx.set_materialize_grads(False)
        q, q_hat = q.contiguous(), q_hat.contiguous()
        rows, dim = q.shape
        hidden = params[0].shape[0]
        c = torch.empty(rows, hidden, device=q.device, dtype=torch.float32)
        saved = torch.empty(lib.mpo_cag_saved_floats(rows, hidden), device=q.device, dtype=torch.float32)
        pa = L.ptr_array(params)
        total = None
        if residual is not None:
            residual = residual.contiguous()
            # dest = (TokenPair, slot): a non-tensor argument -- the slot's view is created here, so autograd sees a fresh output
            total = dest[0].slot(dest[1], (rows, hidden)) if dest is not None else torch.empty_like(c)
        L.check(lib.mpo_cag_forward(L.ptr(q), L.ptr(q_hat), rows, dim, hidden, pa, L.ptr(c), L.ptr(saved), L.ptr(residual),
                                    L.ptr(total), L.stream_of(q)), "mpo_cag_forward")
        ctx.save_for_backward(q, q_hat, c, saved, *params)
        ctx.param_refs = params
        ctx.has_residual = residual is not None
        return (total if total is not None else c), q.view_as(q)

    @staticmethod
    def backward(ctx, dc, d_qpass):
        lib = L.lib()
        q, q_hat, c, saved, *params = ctx.saved_tensors
        rows, dim = q.shape
        hidden = params[0].shape[0]
        if dc is None:
            dc = torch.zeros_like(c)
        dc = dc.contiguous()
        accumulate = d_qpass is not None
        dq = d_qpass.contiguous() if accumulate else torch.empty_like(q)      # (in place on the incoming gradient)
        dqh = torch.empty_like(q_hat)
        grads = [grad_out(p) for p in ctx.param_refs]
        ws = _workspace(lib.mpo_cag_workspace_bytes(rows, hidden), q.device)
        pa, ga = L.ptr_array(params), L.ptr_array(grads)
        L.check(lib.mpo_cag_backward(L.ptr(q), L.ptr(q_hat), rows, dim, hidden, pa, L.ptr(saved), L.ptr(c),
                                     L.ptr(dc), L.ptr(dq), int(accumulate), L.ptr(dqh), ga, L.ptr(ws), ws.numel(),
                                     L.stream_of(q)), "mpo_cag_backward")
        return (dq, dqh, dc if ctx.has_residual else None, None, *grads)


def contextual_gate(q, q_hat, cag, residual=None, dest=None, hand_on: bool = False):
    """C = CAG(q, q_hat); with `residual`: residual + C (into slot dest[1] of the TokenPair dest[0] when given).  hand_on: also returns q for its other
    consumers (use THAT tensor there: their gradient is then folded into this op's backward)."""
    out, q_pass = CagFn.apply(q, q_hat, residual, dest, cag.fc1[0].weight, cag.fc1[0].bias, cag.fc2[0].weight, cag.fc2[0].bias,
                              cag.fc3[0].weight, cag.fc3[0].bias, cag.G[1].weight, cag.G[1].bias,
                              cag.E[1].weight, cag.E[1].bias, cag.fc_c[0].weight, cag.fc_c[0].bias)
    return (out, q_pass) if hand_on else out


class EncoderFn(torch.autograd.Function):
    """K4: the whole post-norm nn.TransformerEncoder over (n_slides, T, d) tokens; `branches` encoders of identical
    geometry (x stacked branch-major, params branch-major) go through ONE launch sequence."""

    @staticmethod
    def forward(ctx, x, geom, drop_p, *params):
        lib = L.lib()
        branches, n_slides, T, d, ff, heads, layers = geom
        x = x.contiguous()
        y = torch.empty_like(x)
        saved = torch.empty(lib.mpo_encoder_saved_floats(branches * n_slides, T, d, ff, heads, layers), device=x.device,
                            dtype=torch.float32)
        seed, off = _reserve(lib.mpo_encoder_rng_span(branches * n_slides, T, d, ff, layers)) if drop_p > 0 else (0, 0)
        pa = L.ptr_array(params)
        L.check(lib.mpo_encoder_forward(L.ptr(x), branches, n_slides, T, d, ff, heads, layers, pa, float(drop_p), seed, off,
                                        _epoch(), L.ptr(y), L.ptr(saved), L.stream_of(x)), "mpo_encoder_forward")
        ctx.save_for_backward(x, saved, *params)
        ctx.param_refs = params
        ctx.geom, ctx.drop = geom, (float(drop_p), seed, off)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        x, saved, *params = ctx.saved_tensors
        branches, n_slides, T, d, ff, heads, layers = ctx.geom
        drop_p, seed, off = ctx.drop
        dx = torch.empty_like(x)
        grads = [grad_out(p) for p in ctx.param_refs]
        ws = _workspace(lib.mpo_encoder_workspace_bytes(branches * n_slides, T, d, ff), x.device)
        pa, ga = L.ptr_array(params), L.ptr_array(grads)
        dy = dy.contiguous()
        L.check(lib.mpo_encoder_backward(
            L.ptr(x), branches, n_slides, T, d, ff, heads, layers, pa, drop_p, seed, off, _epoch(), L.ptr(saved), L.ptr(dy),
            L.ptr(dx), ga, L.ptr(ws), ws.numel(), L.stream_of(x)), "mpo_encoder_backward")
        return (dx, None, None, *grads)


def _encoder_params(layers):
    params = []
    for ly in layers:
        params += [ly.self_attn.in_proj_weight, ly.self_attn.in_proj_bias, ly.self_attn.out_proj.weight,
                   ly.self_attn.out_proj.bias, ly.linear1.weight, ly.linear1.bias, ly.linear2.weight, ly.linear2.bias,
                   ly.norm1.weight, ly.norm1.bias, ly.norm2.weight, ly.norm2.bias]
    return params


def _encoder_geom(layers):
    l0 = layers[0]
    return (l0.linear1.in_features, l0.linear1.out_features, l0.self_attn.num_heads, len(layers), l0.dropout.p)


def encoder(x, layers, training: bool):
    """x (B, T, d) through a stack of nn.TransformerEncoderLayer parameter holders."""
    return encoder_branches([x], [layers], training)[0]


def encoder_branches(xs, layer_stacks, training: bool):
    """Several encoders of identical geometry (MCAT's path_transformer and omic_transformer) on inputs of identical
    shape (B, T, d), batched into one launch sequence.  Returns one (B, T, d) tensor per branch."""
    if len(xs) == 1:
        return [encoder_stacked(xs[0].unsqueeze(0), layer_stacks, training)[0]]
    if any(x.shape != xs[0].shape for x in xs):
        raise ValueError("branch-batched encoders need identical input shapes")
    return list(encoder_stacked(torch.stack(list(xs)), layer_stacks, training).unbind(0))


def encoder_stacked(x, layer_stacks, training: bool):
    """x (branches, B, T, d) -> (branches, B, T, d): branch i goes through layer_stacks[i]."""
    nb, b, t, d = x.shape
    g0 = _encoder_geom(layer_stacks[0])
    if len(layer_stacks) != nb or any(_encoder_geom(ls) != g0 for ls in layer_stacks) or g0[0] != d:
        raise ValueError("branch-batched encoders need one layer stack per branch, all of identical geometry")
    params = [p for ls in layer_stacks for p in _encoder_params(ls)]
    geom = (nb, b, t, d, g0[1], g0[2], g0[3])
    y = EncoderFn.apply(x.reshape(nb * b * t, d), geom, g0[4] if training else 0.0, *params)
    return y.view(nb, b, t, d)


class BagSelfAttentionFn(torch.autograd.Function):
    """f3: the attention core of nn.MultiheadAttention with query = key = value = the M rows of a bag
    (models/ge_nacagat/ge_nacagat.py:27,49), one C-ABI call each way; no M x M state is kept between them."""

    @staticmethod
    def forward(ctx, qkv, heads: int, drop_p: float, need_map: bool):
        lib = L.lib()
        n_bags, m, d3 = qkv.shape
        d = d3 // 3
        qkv = qkv.contiguous()
        out = torch.empty((n_bags, m, d), device=qkv.device, dtype=torch.float32)
        saved = torch.empty(lib.mpo_bag_self_attention_saved_floats(n_bags, m, d, heads), device=qkv.device, dtype=torch.float32)
        amap = torch.empty((n_bags, m, m), device=qkv.device, dtype=torch.float32) if need_map else None
        seed, off = _reserve(1) if drop_p > 0 else (0, 0)
        L.check(lib.mpo_bag_self_attention_forward(L.ptr(qkv), n_bags, m, d, heads, float(drop_p), seed, off, _epoch(), L.ptr(out),
                                                   L.ptr(saved), L.ptr(amap), L.stream_of(qkv)), "mpo_bag_self_attention_forward")
        ctx.save_for_backward(qkv, out, saved)
        ctx.geom, ctx.drop = (n_bags, m, d, heads), (float(drop_p), seed, off)
        if amap is not None:
            ctx.mark_non_differentiable(amap)       # the reference returns the map and never differentiates it
        return out, amap

    @staticmethod
    def backward(ctx, d_out, _d_map):
        lib = L.lib()
        qkv, out, saved = ctx.saved_tensors
        n_bags, m, d, heads = ctx.geom
        drop_p, seed, off = ctx.drop
        d_qkv = torch.empty_like(qkv)
        ws = _workspace(lib.mpo_bag_self_attention_workspace_bytes(n_bags, m, d, heads), qkv.device)
        d_out = d_out.contiguous()
        L.check(lib.mpo_bag_self_attention_backward(L.ptr(qkv), L.ptr(out), L.ptr(saved), L.ptr(d_out), n_bags, m, d, heads, drop_p,
                                                    seed, off, _epoch(), L.ptr(d_qkv), L.ptr(ws), ws.numel(), L.stream_of(qkv)),
                "mpo_bag_self_attention_backward")
        return d_qkv, None, None, None


def bag_self_attention(x, mha, training: bool, need_weights: bool = True):
    """x (M, d) or (n_bags, M, d) through the parameters of an nn.MultiheadAttention `mha` as self-attention
    (query = key = value = x) -> (output, map averaged over heads | None) like mha(x, x, x).  The in / out projections are
    plain GEMMs over M rows; the attention itself is bag_selfattn.hip."""
    if mha.in_proj_weight is None or mha.bias_k is not None or mha.add_zero_attn:
        raise NotImplementedError("bag self-attention: packed in_proj, no bias_k / zero-attn (the reference's constructor)")
    if need_weights and mha.num_heads != 1:
        raise NotImplementedError("bag self-attention: the M x M map is returned for one head (models/ge_nacagat/ge_nacagat.py:27)")
    xb = x if x.dim() == 3 else x.unsqueeze(0)
    qkv = linear(xb.float(), mha.in_proj_weight, mha.in_proj_bias)          # the many-row fp32 MFMA GEMM (gemm_f32_rows.hip)
    out, amap = BagSelfAttentionFn.apply(qkv, mha.num_heads, mha.dropout if training else 0.0, bool(need_weights))
    out = linear(out, mha.out_proj.weight, mha.out_proj.bias)
    if x.dim() == 2:
        out, amap = out[0], (amap[0] if amap is not None else None)
    return out, amap


class GatedPoolFn(torch.autograd.Function):
    """K5: gated attention-MIL scorer + softmax pooling + rho, one C-ABI call each way."""

    @staticmethod
    def forward(ctx, x, geom, head_p, rho_p, interleave, *params):
        lib = L.lib()
        ctx.set_materialize_grads(False)
        branches, n_slides, Lr, d = geom
        bt = branches * n_slides
        x = x.contiguous()
        scores = torch.empty(bt * Lr, device=x.device, dtype=torch.float32)
        h = torch.empty((n_slides, branches * d) if interleave else (bt, d), device=x.device, dtype=torch.float32)
        saved = torch.empty(lib.mpo_gated_pool_saved_floats(bt, Lr, d), device=x.device, dtype=torch.float32)
        seed, off = _reserve(lib.mpo_gated_pool_rng_span(bt, Lr, d)) if (head_p > 0 or rho_p > 0) else (0, 0)
        pa = L.ptr_array(params)
        L.check(lib.mpo_gated_pool_forward(L.ptr(x), branches, n_slides, Lr, d, pa, float(head_p), float(rho_p), seed, off,
                                           _epoch(), L.ptr(scores), L.ptr(h), int(interleave), L.ptr(saved), L.stream_of(x)),
                "mpo_gated_pool_forward")
        ctx.save_for_backward(x, saved, h, *params)
        ctx.param_refs = params
        ctx.geom, ctx.drop, ctx.interleave = geom, (float(head_p), float(rho_p)), bool(interleave)
        return scores, h

    @staticmethod
    def backward(ctx, d_scores, dh):
        lib = L.lib()
        x, saved, h, *params = ctx.saved_tensors
        branches, n_slides, Lr, d = ctx.geom
        head_p, rho_p = ctx.drop
        dx = torch.empty_like(x)
        grads = [grad_out(p) for p in ctx.param_refs]
        if dh is None:
            dh = torch.zeros_like(h)
        ws = _workspace(lib.mpo_gated_pool_workspace_bytes(branches * n_slides, Lr, d), x.device)
        pa, ga = L.ptr_array(params), L.ptr_array(grads)
        dh = dh.contiguous()
        d_sc = d_scores.contiguous() if d_scores is not None else None
        L.check(lib.mpo_gated_pool_backward(
            L.ptr(x), branches, n_slides, Lr, d, pa, head_p, rho_p, L.ptr(saved), L.ptr(h), L.ptr(dh), int(ctx.interleave),
            L.ptr(d_sc), L.ptr(dx), ga, L.ptr(ws), ws.numel(), L.stream_of(x)), "mpo_gated_pool_backward")
        return (dx, None, None, None, None, *grads)


def gated_pool(tokens, head, rho, training: bool):
    """tokens (B, L, d) -> raw scores (B, 1, L), pooled embedding (B, d)   (models/mcat/mcat.py:105-109)."""
    return gated_pool_branches([tokens], [head], [rho], training)[0]


def gated_pool_branches(tokens, heads, rhos, training: bool):
    """Several pooling heads of identical geometry on token sets of identical shape (B, L, d) -- the model's
    path / omic attention heads + rho -- in one launch sequence.  Returns [(scores (B,1,L), h (B,d))] per branch."""
    if len(tokens) == 1:
        sc, h = gated_pool_stacked(tokens[0].unsqueeze(0), heads, rhos, training)
        return [(sc[0], h[0])]
    if any(t.shape != tokens[0].shape for t in tokens):
        raise ValueError("branch-batched pooling needs identical token shapes")
    sc, h = gated_pool_stacked(torch.stack(list(tokens)), heads, rhos, training)
    return list(zip(sc.unbind(0), h.unbind(0)))


def gated_pool_stacked(tokens, heads, rhos, training: bool, interleave: bool = False):
    """tokens (branches, B, L, d) -> raw scores (branches, B, 1, L), pooled embeddings (branches, B, d) -- or, with
    interleave, (B, branches * d): row b = [h_branch0 | h_branch1 | ...], the concatenation ConcatFusion reads."""
    nb, b, l, d = tokens.shape
    params = []
    for head, rho in zip(heads, rhos):
        if head.attention_c.weight.shape[0] != 1 or head.attention_a[0].weight.shape != (d, d):
            raise NotImplementedError("gated pooling kernel: n_classes=1 and hidden_dim == input_dim only")
        params += [head.attention_a[0].weight, head.attention_a[0].bias, head.attention_b[0].weight, head.attention_b[0].bias,
                   head.attention_c.weight, head.attention_c.bias, rho[0].weight, rho[0].bias]
    if len(heads) != nb or len(rhos) != nb or len({(h.drop_p, r[2].p) for h, r in zip(heads, rhos)}) != 1:
        raise ValueError("branch-batched pooling needs one head + rho per branch with identical dropout rates")
    scores, h = GatedPoolFn.apply(tokens.reshape(nb * b * l, d), (nb, b, l, d), heads[0].drop_p if training else 0.0,
                                  rhos[0][2].p if training else 0.0, bool(interleave), *params)
    return scores.view(nb, b, 1, l), (h if interleave else h.view(nb, b, d))


class OmicSnnFn(torch.autograd.Function):
    """self.G: all omic SNNs of a window in grouped launches (models/mcat/mcat.py:32-45,90-92)."""

    @staticmethod
    def forward(ctx, drop_p, n_groups, tokens, *args):
        lib = L.lib()
        xs, params = [a.contiguous() for a in args[:n_groups]], args[n_groups:]
        n_slides, d = xs[0].shape[0], params[0].shape[0]
        dev = xs[0].device
        widths = (ctypes.c_int * n_groups)(*[int(x.shape[1]) for x in xs])
        g_bag = tokens.slot(1, (n_slides, n_groups, d)) if tokens is not None else \
            torch.empty(n_slides, n_groups, d, device=dev, dtype=torch.float32)
        saved = torch.empty(lib.mpo_omic_snn_saved_floats(n_slides, n_groups, d), device=dev, dtype=torch.float32)
        seed, off = _reserve(lib.mpo_omic_snn_rng_span(n_slides, n_groups, d)) if drop_p > 0 else (0, 0)
        xa, pa = L.ptr_array(xs), L.ptr_array(params)
        L.check(lib.mpo_omic_snn_forward(xa, widths, n_groups, n_slides, d, pa, float(drop_p), seed, off, _epoch(),
                                         L.ptr(g_bag), L.ptr(saved), L.stream_of(g_bag)), "mpo_omic_snn_forward")
        ctx.save_for_backward(g_bag, saved, *xs, *params)
        ctx.param_refs, ctx.n_groups, ctx.drop = params, n_groups, (float(drop_p), seed, off)
        return g_bag

    @staticmethod
    def backward(ctx, d_g):
        lib = L.lib()
        g_bag, saved, *rest = ctx.saved_tensors
        n = ctx.n_groups
        xs, params = rest[:n], rest[n:]
        n_slides, d = xs[0].shape[0], params[0].shape[0]
        drop_p, seed, off = ctx.drop
        widths = (ctypes.c_int * n)(*[int(x.shape[1]) for x in xs])
        grads = [grad_out(p) for p in ctx.param_refs]
        ws = _workspace(lib.mpo_omic_snn_workspace_bytes(n_slides, n, d), g_bag.device)
        xa, pa, ga = L.ptr_array(xs), L.ptr_array(params), L.ptr_array(grads)
        L.check(lib.mpo_omic_snn_backward(xa, widths, n, n_slides, d, pa, drop_p, seed, off, _epoch(), L.ptr(g_bag),
                                          L.ptr(saved), L.ptr(d_g.contiguous()), ga, L.ptr(ws), ws.numel(),
                                          L.stream_of(g_bag)), "mpo_omic_snn_backward")
        return (None, None, None, *([None] * n), *grads)


def omic_snn(omics, g_modules, training: bool, tokens: "TokenPair | None" = None):
    """omics: per group (B, d_i) -> G_bag (B, N, d).  g_modules: the nn.ModuleList self.G (parameter holders).
    tokens: G_bag is produced into tokens.slot(1)."""
    params = []
    for g in g_modules:
        params += [g[0][0].weight, g[0][0].bias, g[1][0].weight, g[1][0].bias]
    p = g_modules[0][0][2].p if training else 0.0
    return OmicSnnFn.apply(p, len(omics), tokens, *[o.float() for o in omics], *params)


class MapBlockNormFn(torch.autograd.Function):
    """Frobenius norm of every slide's (n_q, M_b) block of a ragged attention map -> (n_slides,)."""

    @staticmethod
    def forward(ctx, flat_map, batch: BagBatch, n_q: int):
        lib = L.lib()
        flat_map = flat_map.contiguous()
        sq = torch.empty(batch.n_slides * n_q, device=flat_map.device, dtype=torch.float32)
        L.check(lib.mpo_map_block_dot(L.ptr(flat_map), L.ptr(flat_map), L.ptr(batch.cu), batch.n_slides, n_q, L.ptr(sq),
                                      L.stream_of(flat_map)), "mpo_map_block_dot")
        norm = sq.view(batch.n_slides, n_q).sum(1).sqrt()
        ctx.save_for_backward(flat_map, norm)
        ctx.batch, ctx.n_q = batch, n_q
        return norm

    @staticmethod
    def backward(ctx, d_norm):
        lib = L.lib()
        flat_map, norm = ctx.saved_tensors
        scale = (d_norm / norm.clamp_min(1e-30)).contiguous()
        d_map = torch.empty_like(flat_map)
        L.check(lib.mpo_map_block_scale(L.ptr(flat_map), L.ptr(scale), L.ptr(ctx.batch.cu), ctx.batch.n_slides, ctx.n_q,
                                        L.ptr(d_map), L.stream_of(flat_map)), "mpo_map_block_scale")
        return d_map, None, None


def map_block_norm(maps) -> torch.Tensor:
    """maps: the RaggedMaps a window forward returns -> per-slide ||A_b||_2, differentiable into the flat map."""
    return MapBlockNormFn.apply(maps.flat, maps.batch, maps.n_q)


class SurvivalHeadFn(torch.autograd.Function):
    """logits (B, C) -> hazards, survs, Y (models/mcat/mcat.py:130-138) on the HIP head kernels."""

    @staticmethod
    def forward(ctx, logits):
        lib = L.lib()
        ctx.set_materialize_grads(False)
        logits = logits.contiguous()
        b, c = logits.shape
        hz, sv, y = (torch.empty_like(logits) for _ in range(3))
        L.check(lib.mpo_survival_head_forward(L.ptr(logits), b, c, L.ptr(hz), L.ptr(sv), L.ptr(y), L.stream_of(logits)),
                "mpo_survival_head_forward")
        ctx.save_for_backward(hz, sv, y)
        return hz, sv, y

    @staticmethod
    def backward(ctx, dhz, dsv, dy):
        lib = L.lib()
        hz, sv, y = ctx.saved_tensors
        b, c = hz.shape
        dhz, dsv, dy = (t.contiguous() if t is not None else None for t in (dhz, dsv, dy))
        dl = torch.empty_like(hz)
        L.check(lib.mpo_survival_head_backward(L.ptr(hz), L.ptr(sv), L.ptr(y), L.ptr(dhz), L.ptr(dsv), L.ptr(dy), b, c,
                                               L.ptr(dl), L.stream_of(hz)), "mpo_survival_head_backward")
        return dl


def survival_head(logits):
    return SurvivalHeadFn.apply(logits)


class CesLossFn(torch.autograd.Function):
    """'ces' loss (models/loss.py:5-28) for a whole window in one launch each way: per-slide losses + risks.
    The torch formulation costs ~60 tiny launches per window (cat/gather/clamp/log and their backward)."""

    @staticmethod
    def forward(ctx, hazards, survs, label, censorship, alpha, eps):
        lib = L.lib()
        ctx.set_materialize_grads(False)        # an unused output must not cost a zero-fill launch in backward
        hazards, survs = hazards.contiguous(), survs.contiguous()
        label = label.view(-1).to(torch.int64).contiguous()
        censorship = censorship.view(-1).to(torch.float32).contiguous()
        b, c = hazards.shape
        loss = torch.empty(b, device=hazards.device, dtype=torch.float32)
        risk = torch.empty(b, device=hazards.device, dtype=torch.float32)
        L.check(lib.mpo_ces_loss_forward(L.ptr(hazards), L.ptr(survs), L.ptr(label), L.ptr(censorship), b, c, float(alpha),
                                         float(eps), L.ptr(loss), L.ptr(risk), L.stream_of(hazards)), "mpo_ces_loss_forward")
        ctx.save_for_backward(hazards, survs, label, censorship)
        ctx.cfg = (float(alpha), float(eps))
        ctx.mark_non_differentiable(risk)
        return loss, risk

    @staticmethod
    def backward(ctx, d_loss, _d_risk):
        lib = L.lib()
        if d_loss is None:
            return None, None, None, None, None, None
        hazards, survs, label, censorship = ctx.saved_tensors
        alpha, eps = ctx.cfg
        b, c = hazards.shape
        # loss.sum().backward() hands an expanded (stride-0) gradient: pass its one element, no materialisation
        scalar = d_loss.stride(0) == 0 and b > 1
        d_loss = d_loss.as_strided((1,), (1,)) if scalar else d_loss.contiguous()
        d_hz, d_sv = torch.empty_like(hazards), torch.empty_like(survs)
        L.check(lib.mpo_ces_loss_backward(L.ptr(hazards), L.ptr(survs), L.ptr(label), L.ptr(censorship), b, c, alpha, eps,
                                          L.ptr(d_loss), int(scalar), L.ptr(d_hz), L.ptr(d_sv), L.stream_of(hazards)),
                "mpo_ces_loss_backward")
        return d_hz, d_sv, None, None, None, None


def ces_loss(hazards, survs, label, censorship, alpha: float = 0.75, eps: float = 1e-7):
    """-> (per-slide 'ces' loss (B,), risk (B,)); reduce with .sum()/.mean() as the caller needs."""
    return CesLossFn.apply(hazards, survs, label, censorship, alpha, eps)


class FusionHeadFn(torch.autograd.Function):
    """K6: concat-fusion MLP + classifier + survival head."""

    @staticmethod
    def forward(ctx, hcat, *params):
        lib = L.lib()
        ctx.set_materialize_grads(False)
        hcat = hcat.contiguous()
        b, din = hcat.shape
        hidden, dout, c = params[0].shape[0], params[2].shape[0], params[4].shape[0]
        hz = torch.empty(b, c, device=hcat.device, dtype=torch.float32)
        sv, y = torch.empty_like(hz), torch.empty_like(hz)
        saved = torch.empty(lib.mpo_fusion_head_saved_floats(b, hidden, dout, c), device=hcat.device, dtype=torch.float32)
        pa = L.ptr_array(params)
        L.check(lib.mpo_fusion_head_forward(L.ptr(hcat), b, din, hidden, dout, c, pa, L.ptr(hz), L.ptr(sv), L.ptr(y),
                                            L.ptr(saved), L.stream_of(hcat)), "mpo_fusion_head_forward")
        ctx.save_for_backward(hcat, saved, hz, sv, y, *params)
        ctx.param_refs = params
        return hz, sv, y

    @staticmethod
    def backward(ctx, dhz, dsv, dy):
        lib = L.lib()
        hcat, saved, hz, sv, y, *params = ctx.saved_tensors
        b, din = hcat.shape
        hidden, dout, c = params[0].shape[0], params[2].shape[0], params[4].shape[0]
        d_hcat = torch.empty_like(hcat)
        grads = [grad_out(p) for p in ctx.param_refs]
        ws = _workspace(lib.mpo_fusion_head_workspace_bytes(b, hidden, dout, c), hcat.device)
        pa, ga = L.ptr_array(params), L.ptr_array(grads)
        dhz, dsv, dy = (t.contiguous() if t is not None else None for t in (dhz, dsv, dy))
        L.check(lib.mpo_fusion_head_backward(
            L.ptr(hcat), b, din, hidden, dout, c, pa, L.ptr(saved), L.ptr(hz), L.ptr(sv), L.ptr(y), L.ptr(dhz), L.ptr(dsv),
            L.ptr(dy), L.ptr(d_hcat), ga, L.ptr(ws), ws.numel(), L.stream_of(hcat)), "mpo_fusion_head_backward")
        return (d_hcat, *grads)


def fusion_head(h_path, h_omic, fusion_layer, classifier):
    """(B,d),(B,d) -> hazards, survs, Y (B, C)   (models/fusion.py:17-19 + models/mcat/mcat.py:126-138)."""
    return fusion_head_cat(torch.cat([h_path, h_omic], dim=-1), fusion_layer, classifier)


def fusion_head_cat(hcat, fusion_layer, classifier):
    """hcat (B, 2d) = [h_path | h_omic] already concatenated."""
    seq = fusion_layer.fusion_layer
    return FusionHeadFn.apply(hcat, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, classifier.weight,
                              classifier.bias)


class FusionHeadLossFn(torch.autograd.Function):
    """K6 + `ces` loss for a training step: ConcatFusion MLP + classifier GEMMs, then head, loss and the backward of both
    in ONE launch.  The gradient the caller sends into the per-slide loss must be known up front: `slide_weight`
    (B device floats; 1 / grad_acc_step in the reference's loop, models/mcat/main.py:69-70) -- backward() refuses any other
    gradient tensor.  Returns (loss (B,), risk (B,), hazards, survs, Y); only `loss` carries gradient."""

    @staticmethod
    def forward(ctx, hcat, label, censorship, slide_weight, alpha, eps, *params):
        lib = L.lib()
        ctx.set_materialize_grads(False)
        hcat = hcat.contiguous()
        b, din = hcat.shape
        hidden, dout, c = params[0].shape[0], params[2].shape[0], params[4].shape[0]
        label = label.view(-1).to(torch.int64).contiguous()
        censorship = censorship.view(-1).to(torch.float32).contiguous()
        if slide_weight.shape != (b,) or slide_weight.dtype != torch.float32 or not slide_weight.is_contiguous():
            raise ValueError("fusion_head_loss: slide_weight must be a contiguous fp32 tensor of one value per slide")
        dev = hcat.device
        hz = torch.empty(b, c, device=dev, dtype=torch.float32)
        sv, y = torch.empty_like(hz), torch.empty_like(hz)
        loss = torch.empty(b, device=dev, dtype=torch.float32)
        risk = torch.empty(b, device=dev, dtype=torch.float32)
        saved = torch.empty(lib.mpo_fusion_head_loss_saved_floats(b, hidden, dout, c), device=dev, dtype=torch.float32)
        L.check(lib.mpo_fusion_head_loss_forward(
            L.ptr(hcat), b, din, hidden, dout, c, L.ptr_array(params), L.ptr(label), L.ptr(censorship), L.ptr(slide_weight),
            float(alpha), float(eps), L.ptr(hz), L.ptr(sv), L.ptr(y), L.ptr(loss), L.ptr(risk), L.ptr(saved),
            L.stream_of(hcat)), "mpo_fusion_head_loss_forward")
        ctx.save_for_backward(hcat, saved, slide_weight, *params)
        ctx.param_refs = params
        ctx.mark_non_differentiable(risk, hz, sv, y)
        return loss, risk, hz, sv, y

    @staticmethod
    def backward(ctx, d_loss, *_unused):
        lib = L.lib()
        hcat, saved, slide_weight, *params = ctx.saved_tensors
        if d_loss is None:
            return (None,) * (6 + len(params))
        if d_loss.data_ptr() != slide_weight.data_ptr() or d_loss.shape != slide_weight.shape:
            raise RuntimeError("fusion_head_loss: backward() must be driven with the slide_weight tensor given to forward "
                               "(the loss gradient is folded into the forward launch)")
        b, din = hcat.shape
        hidden, dout, c = params[0].shape[0], params[2].shape[0], params[4].shape[0]
        d_hcat = torch.empty_like(hcat)
        grads = [grad_out(p) for p in ctx.param_refs]
        ws = _workspace(lib.mpo_fusion_head_workspace_bytes(b, hidden, dout, c), hcat.device)
        L.check(lib.mpo_fusion_head_loss_backward(
            L.ptr(hcat), b, din, hidden, dout, c, L.ptr_array(params), L.ptr(saved), L.ptr(d_hcat), L.ptr_array(grads),
            L.ptr(ws), ws.numel(), L.stream_of(hcat)), "mpo_fusion_head_loss_backward")
        return (d_hcat, None, None, None, None, None, *grads)


def fusion_head_loss_cat(hcat, fusion_layer, classifier, label, censorship, slide_weight, alpha: float = 0.75, eps: float = 1e-7):
    """Training-step K6: -> (per-slide `ces` loss, risk, hazards, survs, Y); drive backward with `slide_weight` itself."""
    seq = fusion_layer.fusion_layer
    return FusionHeadLossFn.apply(hcat, label, censorship, slide_weight, alpha, eps, seq[0].weight, seq[0].bias,
                                  seq[2].weight, seq[2].bias, classifier.weight, classifier.bias)


def bump_step_counters(rng_epoch=None, adam_step=None):
    """rng_epoch (int64[1]) += 1 and adam_step (int32[1]) += 1 in one launch (either may be None)."""
    t = rng_epoch if rng_epoch is not None else adam_step
    L.check(L.lib().mpo_step_counters_bump(L.ptr(rng_epoch), L.ptr(adam_step), L.stream_of(t)), "mpo_step_counters_bump")


# ------------------------------------------------------------------------------------ K2
_rng_calls = 0


def next_dropout_stream(n_elements: int):
    """(seed, offset) for one dropout mask of n_elements: the generator's counter space is carved sequentially per
    process, the seed follows torch.initial_seed() (so torch.manual_seed(rank-dependent) de-correlates ranks)."""
    global _rng_calls
    seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    offset = _rng_calls
    _rng_calls += (n_elements + 3) // 4 + 1
    return seed, offset


# False: the patch-side gradient of K2 as library GEMM + mpo_nacagat_patch_grad (the r02 path; kept for the small model and as
# the cross-check of csrc/k2_patchgrad.hip in tools/gpu_diag_nacagat.py)
k2_fused_patch_grad = True


class CoAttnNaCAGaTFn(torch.autograd.Function):
    """NaCAGaT narrow-gated attention core over a ragged window (models/blocks.py:114-206).
    Returns (q_proj, attn_out, post-dropout map).  K = H W_k^T + b_k is a plain GEMM done here with
    torch (rocBLAS/hipBLASLt), like the model's patch layer; everything else is HIP.  K is always fp32,
    also for a bf16-stored bag: the gate multiplies k's rounding error (hard part 4)."""

    @staticmethod
    def forward(ctx, query, bag_data, in_w, in_b, out_w, out_b, batch: BagBatch, drop_p: float, bag_relu_gate: float = 0.0):
        lib = L.lib()
        n_slides = batch.n_slides
        R, E = query.shape
        n_q = R // n_slides
        dev, T = query.device, batch.total_rows
        query = query.contiguous()
        ctx.set_materialize_grads(False)
        ctx.bag_relu_gate = float(bag_relu_gate)
        ctx.bag_bias = getattr(bag_data, "_mpo_bias_param", None)
        if ctx.bag_relu_gate != 0.0 and bag_data.dtype != torch.bfloat16:
            raise ValueError("bag_relu_gate (fused ReLU/dropout derivative of the patch layer) needs a bf16-stored bag")
        big = E == 512
        if big and ctx.bag_relu_gate != 0.0:
            raise ValueError("embed_dim 512: the fused ReLU/dropout gate of the patch layer is built for embed_dim <= 256")
        hb = bag_data
        if bag_data.dtype == torch.bfloat16 and E == 256:
            # HIP key projection: bf16 bag (exact) x fp32 weights split into three bf16 terms, fp32 accumulate and output
            kbag = torch.empty(T, E, device=dev, dtype=torch.float32)
            w_k, b_k = in_w[E:2 * E], in_b[E:2 * E]
            L.check(lib.mpo_key_projection(L.ptr(bag_data), T, E, L.ptr(w_k), L.ptr(b_k), L.ptr(kbag), L.stream_of(query)),
                    "mpo_key_projection")
        elif big:
            # 'big' (models/nacagat/nacagat.py:17-18): both bags in the split-halves layout [2][T][256] of include/mpo_hip.h --
            # K half h straight out of its own GEMM (rows 256 h .. of W_k), the bag as one strided copy
            xf = bag_data.float().contiguous()
            kbag = torch.empty(2, T, 256, device=dev, dtype=torch.float32)
            for h in range(2):
                L.check(lib.mpo_linear_forward(L.ptr(xf), L.ptr(in_w[E + 256 * h:E + 256 * (h + 1)]), L.ptr(in_b[E + 256 * h:E + 256 * (h + 1)]),
                                               L.ptr(kbag[h]), T, E, 256, 1.0, L.ACT["none"], L.stream_of(query)), "mpo_linear_forward")
            hb = bag_data.view(T, 2, 256).permute(1, 0, 2).contiguous()
        else:
            # fp32 bag (or the small model's bf16 bag): the exact-fp32 MFMA GEMM of the token tail in its many-row form
            kbag = torch.empty(T, E, device=dev, dtype=torch.float32)
            L.check(lib.mpo_linear_forward(L.ptr(bag_data.float().contiguous()), L.ptr(in_w[E:2 * E]), L.ptr(in_b[E:2 * E]), L.ptr(kbag),
                                           T, E, E, 1.0, L.ACT["none"], L.stream_of(query)), "mpo_linear_forward")
        q_proj = torch.empty(R, E, device=dev, dtype=torch.float32)
        out = torch.empty(R, E, device=dev, dtype=torch.float32)
        amap = torch.empty(n_q * T, device=dev, dtype=torch.float32)
        score_maps = torch.empty(2 * n_q * T, device=dev, dtype=torch.float32)
        saved = torch.empty(lib.mpo_nacagat_saved_floats(n_slides, n_q, E), device=dev, dtype=torch.float32)
        ws = _workspace(lib.mpo_nacagat_workspace_bytes(n_slides, n_q, E, batch.max_rows, T), dev)
        seed, offset = next_dropout_stream(n_q * T) if drop_p > 0 else (0, 0)
        L.check(lib.mpo_coattn_nacagat_forward(
            L.ptr(kbag), L.MPO_F32, L.ptr(hb), L.bag_dtype_code(bag_data), L.ptr(batch.cu), n_slides, T, batch.max_rows,
            L.ptr(query), n_q, E, L.ptr(in_w), L.ptr(in_b), L.ptr(out_w), L.ptr(out_b), float(drop_p), seed, offset,
            _epoch(), L.ptr(q_proj), L.ptr(out), L.ptr(amap), L.ptr(score_maps), L.ptr(saved),
            batch.plan(), L.ptr(ws), ws.numel(), L.stream_of(query)), "mpo_coattn_nacagat_forward")
        ctx.save_for_backward(query, bag_data, kbag, in_w, in_b, out_w, saved, score_maps, amap)
        ctx.hb = hb if big else None                  # (the split-halves copy of the bag: kept for the backward)
        ctx.param_refs = (in_w, in_b, out_w, out_b)
        ctx.batch, ctx.n_q, ctx.drop = batch, n_q, (float(drop_p), seed, offset)
        # (4th output: the query handed on to its other consumers -- NaCAGaT's CAG and the omic branch's tokens; their gradient
        #  arrives here and the backward's d_query product accumulates onto it)
        return q_proj, out, amap, query.view_as(query)

    @staticmethod
    def backward(ctx, d_qproj, d_out, d_map, d_qpass=None):
        lib = L.lib()
        query, bag_data, kbag, in_w, in_b, out_w, saved, score_maps, amap = ctx.saved_tensors
        batch, n_q = ctx.batch, ctx.n_q
        drop_p, seed, offset = ctx.drop
        R, E = query.shape
        dev, T = query.device, batch.total_rows
        d_out = d_out.contiguous() if d_out is not None else torch.zeros(R, E, device=dev)
        d_qproj = d_qproj.contiguous() if d_qproj is not None else None
        d_map = d_map.contiguous() if d_map is not None else None
        accumulate = d_qpass is not None
        d_query = d_qpass.contiguous() if accumulate else torch.empty_like(query)      # (in place on the incoming gradient)
        d_k = torch.empty_like(kbag, dtype=bag_data.dtype)       # a bf16 bag takes its key gradient in bf16 (see below)
        # bf16 bag: the patch-side gradient is finished by ONE pass after the dK W_k GEMM (mpo_nacagat_patch_grad) instead
        # of outer-product kernel -> addmm_ read-modify-write -> element-wise derivative pass
        big = E == 512
        hb = ctx.hb if big else bag_data
        fused_patch = bag_data.dtype == torch.bfloat16 and not big
        d_h = None if fused_patch else torch.empty_like(hb)
        d_ctx = torch.empty(R, E, device=dev, dtype=torch.float32) if fused_patch else None
        d_in_w, d_in_b, d_out_w, d_out_b = (grad_out(p) for p in ctx.param_refs)
        ws = _workspace(lib.mpo_nacagat_workspace_bytes(batch.n_slides, n_q, E, batch.max_rows, T), dev)
        L.check(lib.mpo_coattn_nacagat_backward(
            L.ptr(kbag), L.MPO_F32, L.ptr(hb), L.bag_dtype_code(bag_data), L.ptr(batch.cu), batch.n_slides, T,
            batch.max_rows, L.ptr(query), n_q, E, L.ptr(in_w), L.ptr(in_b), L.ptr(out_w), drop_p, seed, offset,
            _epoch(), L.ptr(saved), L.ptr(score_maps), L.ptr(amap), L.ptr(d_out), L.ptr(d_map), L.ptr(d_qproj),
            L.ptr(d_query), int(accumulate), L.ptr(d_k), L.bag_dtype_code(d_k), L.ptr(d_in_b[E:2 * E]), L.ptr(d_h), L.ptr(d_ctx), L.ptr(d_in_w), L.ptr(d_in_b), L.ptr(d_out_w),
            L.ptr(d_out_b), batch.plan(), L.ptr(ws), ws.numel(), L.stream_of(query)), "mpo_coattn_nacagat_backward")
        # back through the caller-side GEMM  K = H W_k^T + b_k.  The forward K stays fp32 (the gate amplifies its
        # rounding); its GRADIENT goes through bf16 operands with fp32 accumulation for a bf16 bag: dW_k as a
        # batched split-K product (one 480 000-deep fp32 contraction took 1.19 ms in rocBLAS), dH += dK W_k as a
        # bf16 GEMM (0.59 ms in fp32).
        w_k = in_w[E:2 * E]
        if fused_patch:
            gate = ctx.bag_relu_gate
            colsum = _bias_grad_slot(ctx.bag_bias, E, dev) if gate != 0.0 else None
            if E == 256 and k2_fused_patch_grad:
                # dH = (dK W_k + A_drop^T dctx) (.) gate in ONE hand-written pass (no library GEMM): csrc/k2_patchgrad.hip
                d_h = torch.empty_like(d_k)
                L.check(lib.mpo_nacagat_patch_grad_fused(L.ptr(batch.cu), batch.n_slides, T, batch.max_rows, n_q, E, L.ptr(amap),
                                                         L.ptr(d_ctx), L.ptr(d_k), L.ptr(w_k), L.ptr(bag_data), L.ptr(d_h), gate,
                                                         L.ptr(colsum), batch.plan(), L.ptr(ws), ws.numel(), L.stream_of(query)),
                        "mpo_nacagat_patch_grad_fused")
            else:                                  # the small model (E = 128): dK W_k on the fp32 MFMA GEMM (many-row form), then the one-pass epilogue
                dhf = torch.empty(T, E, device=dev, dtype=torch.float32)
                L.check(lib.mpo_linear_backward_input(L.ptr(d_k.float()), L.ptr(w_k), L.ptr(dhf), T, E, E, 1.0, 0, L.stream_of(query)),
                        "mpo_linear_backward_input")
                d_h = dhf.to(torch.bfloat16)
                L.check(lib.mpo_nacagat_patch_grad(L.ptr(batch.cu), batch.n_slides, T, batch.max_rows, n_q, E, L.ptr(amap),
                                                   L.ptr(d_ctx), L.ptr(d_h), L.ptr(bag_data), L.ptr(d_h), gate, L.ptr(colsum),
                                                   batch.plan(), L.ptr(ws), ws.numel(), L.stream_of(query)), "mpo_nacagat_patch_grad")
            if colsum is not None:
                d_h._mpo_colsum = colsum          # the producing layer's bias gradient (PatchFcFn.backward picks it up)
            patch_weight_grad(d_k, bag_data, d_in_w[E:2 * E])     # dW_k = d_k^T H_bag (hand-written for 256 x 256, bf16)
        elif big:
            # split-halves results back to [T, 512], then the two products through the key projection on the fp32 MFMA GEMMs
            # (a bf16 bag: in fp32, rounded once on the way out)
            s_ = L.stream_of(query)
            dkf = d_k.permute(1, 0, 2).reshape(T, E).float().contiguous()
            dhf = d_h.permute(1, 0, 2).reshape(T, E).float().contiguous()
            L.check(lib.mpo_linear_backward_input(L.ptr(dkf), L.ptr(w_k), L.ptr(dhf), T, E, E, 1.0, 1, s_), "mpo_linear_backward_input")
            L.check(lib.mpo_linear_backward_weight(L.ptr(dkf), L.ptr(bag_data.float().contiguous()), L.ptr(d_in_w[E:2 * E]), None, T, E, E,
                                                   1.0, s_), "mpo_linear_backward_weight")
            d_h = dhf.to(bag_data.dtype)
        else:
            # fp32 bag: d_h += d_k W_k and dW_k = d_k^T H on the fp32 MFMA GEMMs (many-row / long-K forms)
            s_ = L.stream_of(query)
            L.check(lib.mpo_linear_backward_input(L.ptr(d_k), L.ptr(w_k), L.ptr(d_h), T, E, E, 1.0, 1, s_), "mpo_linear_backward_input")
            L.check(lib.mpo_linear_backward_weight(L.ptr(d_k), L.ptr(bag_data), L.ptr(d_in_w[E:2 * E]), None, T, E, E, 1.0, s_),
                    "mpo_linear_backward_weight")
        # (d_in_b[E:2E], the key bias gradient = column sums of d_k, came out of the kernel that wrote d_k)
        return d_query, d_h, d_in_w, d_in_b, d_out_w, d_out_b, None, None, None


def coattn_nacagat(query, batch: BagBatch, in_w, in_b, out_w, out_b, drop_p: float, bag_relu_gate: float = 0.0,
                   hand_on: bool = False):
    """query (n_slides*n_q, E) -> (q_proj, attn_out (n_slides*n_q, E), ragged post-dropout map[, query handed on]).
    bag_relu_gate = 1/(1-p) when the bag comes from patch_fc(..., pre_gated_grad=True): d_bag then already carries
    that layer's ReLU/dropout derivative (bf16 bags only).  hand_on: a 4th result, the query for its other consumers."""
    res = CoAttnNaCAGaTFn.apply(query, batch.data, in_w, in_b, out_w, out_b, batch, drop_p, bag_relu_gate)
    return res if hand_on else res[:3]
