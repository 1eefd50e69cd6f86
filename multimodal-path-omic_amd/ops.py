"""torch.autograd.Function wrappers around the C-ABI entries of libmpo_hip.so.

One Function per kernel family of SURVEY.md section 2 (K1..K6).  Each forward/backward is ONE call
across the ABI; the sequence of kernel launches lives in csrc/capi.hip.  Buffers (outputs, saved
tensors, workspaces) are torch allocations; the library keeps nothing.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib as L


@dataclass
class BagBatch:
    """A window of slides' bags concatenated along rows ("ragged").

    data  (total_rows, E) fp32 or bf16;  cu  device int32 (n_slides+1) row offsets;
    lengths  host list of M_b (grid sizing needs max and total on the host, without a sync).
    """
    data: torch.Tensor
    cu: torch.Tensor
    lengths: List[int]

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

    def with_data(self, data: torch.Tensor) -> "BagBatch":
        assert data.shape[0] == self.total_rows
        return BagBatch(data, self.cu, self.lengths)

    def split_map(self, flat_map: torch.Tensor, n_q: int) -> "List[torch.Tensor]":
        """Ragged attention map -> list of (n_q, M_b) views (slide b starts at n_q * cu[b])."""
        out, off = [], 0
        for m in self.lengths:
            out.append(flat_map[n_q * off:n_q * (off + m)].view(n_q, m))
            off += m
        return out


def make_cu(lengths, device):
    cu = [0]
    for m in lengths:
        if m < 1:
            raise ValueError("every slide needs at least one patch")
        cu.append(cu[-1] + int(m))
    return torch.tensor(cu, dtype=torch.int32).to(device, non_blocking=True)


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------ linear
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the fp32 MFMA GEMM (stand-in for F.linear on the small-row tail)."""

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
        dx = torch.empty_like(x2)
        dw = torch.empty_like(weight)
        db = torch.empty(O, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
        s = L.stream_of(dy)
        L.check(L.lib().mpo_linear_backward_input(L.ptr(dy), L.ptr(weight), L.ptr(dx), R, I, O, 1.0, 0, s),
                "mpo_linear_backward_input")
        L.check(L.lib().mpo_linear_backward_weight(L.ptr(dy), L.ptr(x2), L.ptr(dw), L.ptr(db), R, I, O, 1.0, s),
                "mpo_linear_backward_weight")
        return dx.view(ctx.xshape), dw, db, None


def linear(x, weight, bias=None, act="none"):
    return LinearFn.apply(x, weight, bias, act)


# ------------------------------------------------------------------------------------ K1
class CoAttnMCATFn(torch.autograd.Function):
    """MCAT co-attention over a ragged window (models/mcat/mcat.py:97)."""

    @staticmethod
    def forward(ctx, query, bag_data, in_w, in_b, out_w, out_b, batch: BagBatch, need_weights: bool):
        lib = L.lib()
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
            L.ptr(out), L.ptr(amap), L.ptr(saved), L.ptr(ws), ws.numel(), L.stream_of(query)),
            "mpo_coattn_mcat_forward")
        ctx.save_for_backward(query, bag_data, in_w, out_w, saved, amap)
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
        d_out = d_out.contiguous()
        if d_map is not None:
            d_map = d_map.contiguous()
        d_query = torch.empty_like(query)
        d_bag = torch.empty_like(bag_data)
        d_in_w = torch.empty_like(in_w)
        d_in_b = torch.empty(3 * E, device=dev, dtype=torch.float32)
        d_out_w = torch.empty_like(out_w)
        d_out_b = torch.empty(E, device=dev, dtype=torch.float32)
        ws = _workspace(lib.mpo_coattn_workspace_bytes(batch.n_slides, n_q, E, batch.max_rows), dev)
        L.check(lib.mpo_coattn_mcat_backward(
            L.ptr(bag_data), L.bag_dtype_code(bag_data), L.ptr(batch.cu), batch.n_slides, batch.total_rows,
            batch.max_rows, L.ptr(query), n_q, E, L.ptr(in_w), L.ptr(out_w), L.ptr(saved), L.ptr(amap),
            L.ptr(d_out), L.ptr(d_map), L.ptr(d_query), L.ptr(d_bag), L.ptr(d_in_w), L.ptr(d_in_b),
            L.ptr(d_out_w), L.ptr(d_out_b), L.ptr(ws), ws.numel(), L.stream_of(query)),
            "mpo_coattn_mcat_backward")
        return d_query, d_bag, d_in_w, d_in_b, d_out_w, d_out_b, None, None


def coattn_mcat(query, batch: BagBatch, in_w, in_b, out_w, out_b, need_weights: bool):
    """query (n_slides*n_q, E) -> (out (n_slides*n_q, E), ragged map or None)."""
    return CoAttnMCATFn.apply(query, batch.data, in_w, in_b, out_w, out_b, batch, need_weights)


# ------------------------------------------------------------------------------------ tail (6 x d tokens per slide)
# The linears below run on the HIP fp32-MFMA GEMM; the element-wise glue between them is being
# moved into fused HIP kernels family by family (K3..K6).
import torch.nn.functional as F  # noqa: E402


def gated_scores(x, wa, ba, wb, bb, wc, bc, drop_p: float):
    """AttentionNetGated scores (models/blocks.py:42-47): x (..., L, D) -> (..., L, n_classes)."""
    a = linear(x, wa, ba, "tanh")
    b = linear(x, wb, bb, "sigmoid")
    if drop_p > 0.0:
        a = F.dropout(a, drop_p, True)
        b = F.dropout(b, drop_p, True)
    return linear(a * b, wc, bc)


def contextual_gate(q, q_hat, cag):
    """ContextualAttentionGate.forward (models/blocks.py:247-253) on (R, D) rows."""
    g = F.elu(linear(q, cag.fc1[0].weight, cag.fc1[0].bias, "elu") + linear(q_hat, cag.fc2[0].weight, cag.fc2[0].bias, "elu"))
    g = F.layer_norm(g, g.shape[-1:], cag.G[1].weight, cag.G[1].bias, cag.G[1].eps)
    e = F.elu(linear(q_hat, cag.fc3[0].weight, cag.fc3[0].bias, "elu"))
    e = F.layer_norm(e, e.shape[-1:], cag.E[1].weight, cag.E[1].bias, cag.E[1].eps)
    return linear(g * e, cag.fc_c[0].weight, cag.fc_c[0].bias, "elu")


def encoder_layer(x, layer, training: bool):
    """One post-norm TransformerEncoderLayer on x (B, T, d) (torch/nn/modules/transformer.py:661)."""
    sa = layer.self_attn
    b, t, d = x.shape
    h = sa.num_heads
    p = layer.dropout.p if training else 0.0
    qkv = linear(x, sa.in_proj_weight, sa.in_proj_bias).view(b, t, 3, h, d // h)
    q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
    s = torch.softmax(q @ k.transpose(-1, -2) / float(d // h) ** 0.5, dim=-1)
    if p > 0.0:
        s = F.dropout(s, p, True)
    o = (s @ v).transpose(1, 2).reshape(b, t, d)
    o = linear(o, sa.out_proj.weight, sa.out_proj.bias)
    if p > 0.0:
        o = F.dropout(o, p, True)
    x = F.layer_norm(x + o, (d,), layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
    f = linear(x, layer.linear1.weight, layer.linear1.bias, "relu")
    if p > 0.0:
        f = F.dropout(f, p, True)
    f = linear(f, layer.linear2.weight, layer.linear2.bias)
    if p > 0.0:
        f = F.dropout(f, p, True)
    return F.layer_norm(x + f, (d,), layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)


# ------------------------------------------------------------------------------------ K2
_rng_calls = 0


def next_dropout_stream(n_elements: int):
    """(seed, offset) for one dropout mask of n_elements: Philox counter space is carved sequentially per
    process, the seed follows torch.initial_seed() (so torch.manual_seed(rank-dependent) de-correlates ranks)."""
    global _rng_calls
    seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    offset = _rng_calls
    _rng_calls += (n_elements + 3) // 4 + 1
    return seed, offset


class CoAttnNaCAGaTFn(torch.autograd.Function):
    """NaCAGaT narrow-gated attention core over a ragged window (models/blocks.py:114-206).
    Returns (q_proj, attn_out, post-dropout map).  K = H W_k^T + b_k is a plain GEMM done here with
    torch (rocBLAS/hipBLASLt), like the model's patch layer; everything else is HIP.  K is always fp32,
    also for a bf16-stored bag: the gate multiplies k's rounding error (hard part 4)."""

    @staticmethod
    def forward(ctx, query, bag_data, in_w, in_b, out_w, out_b, batch: BagBatch, drop_p: float):
        lib = L.lib()
        n_slides = batch.n_slides
        R, E = query.shape
        n_q = R // n_slides
        dev, T = query.device, batch.total_rows
        query = query.contiguous()
        kbag = F.linear(bag_data.float(), in_w[E:2 * E], in_b[E:2 * E])
        tkbag = torch.empty_like(kbag)
        q_proj = torch.empty(R, E, device=dev, dtype=torch.float32)
        out = torch.empty(R, E, device=dev, dtype=torch.float32)
        amap = torch.empty(n_q * T, device=dev, dtype=torch.float32)
        score_maps = torch.empty(2 * n_q * T, device=dev, dtype=torch.float32)
        saved = torch.empty(lib.mpo_nacagat_saved_floats(n_slides, n_q, E), device=dev, dtype=torch.float32)
        ws = _workspace(lib.mpo_nacagat_workspace_bytes(n_slides, n_q, E, batch.max_rows, T), dev)
        seed, offset = next_dropout_stream(n_q * T) if drop_p > 0 else (0, 0)
        L.check(lib.mpo_coattn_nacagat_forward(
            L.ptr(kbag), L.MPO_F32, L.ptr(bag_data), L.bag_dtype_code(bag_data), L.ptr(batch.cu), n_slides, T, batch.max_rows,
            L.ptr(query), n_q, E, L.ptr(in_w), L.ptr(in_b), L.ptr(out_w), L.ptr(out_b), float(drop_p), seed, offset,
            L.ptr(tkbag), L.ptr(q_proj), L.ptr(out), L.ptr(amap), L.ptr(score_maps), L.ptr(saved),
            L.ptr(ws), ws.numel(), L.stream_of(query)), "mpo_coattn_nacagat_forward")
        ctx.save_for_backward(query, bag_data, kbag, tkbag, in_w, in_b, out_w, saved, score_maps, amap)
        ctx.batch, ctx.n_q, ctx.drop = batch, n_q, (float(drop_p), seed, offset)
        return q_proj, out, amap

    @staticmethod
    def backward(ctx, d_qproj, d_out, d_map):
        lib = L.lib()
        query, bag_data, kbag, tkbag, in_w, in_b, out_w, saved, score_maps, amap = ctx.saved_tensors
        batch, n_q = ctx.batch, ctx.n_q
        drop_p, seed, offset = ctx.drop
        R, E = query.shape
        dev, T = query.device, batch.total_rows
        d_out = d_out.contiguous() if d_out is not None else torch.zeros(R, E, device=dev)
        d_qproj = d_qproj.contiguous() if d_qproj is not None else None
        d_map = d_map.contiguous() if d_map is not None else None
        d_query = torch.empty_like(query)
        d_k = torch.empty_like(kbag)
        d_tk = torch.empty_like(kbag)
        d_h = torch.empty_like(bag_data)
        d_in_w = torch.empty_like(in_w)
        d_in_b = torch.empty(3 * E, device=dev, dtype=torch.float32)
        d_out_w = torch.empty_like(out_w)
        d_out_b = torch.empty(E, device=dev, dtype=torch.float32)
        ws = _workspace(lib.mpo_nacagat_workspace_bytes(batch.n_slides, n_q, E, batch.max_rows, T), dev)
        L.check(lib.mpo_coattn_nacagat_backward(
            L.ptr(kbag), L.ptr(tkbag), L.MPO_F32, L.ptr(bag_data), L.bag_dtype_code(bag_data), L.ptr(batch.cu), batch.n_slides, T,
            batch.max_rows, L.ptr(query), n_q, E, L.ptr(in_w), L.ptr(in_b), L.ptr(out_w), drop_p, seed, offset,
            L.ptr(saved), L.ptr(score_maps), L.ptr(amap), L.ptr(d_out), L.ptr(d_map), L.ptr(d_qproj),
            L.ptr(d_query), L.ptr(d_k), L.ptr(d_tk), L.ptr(d_h), L.ptr(d_in_w), L.ptr(d_in_b), L.ptr(d_out_w),
            L.ptr(d_out_b), L.ptr(ws), ws.numel(), L.stream_of(query)), "mpo_coattn_nacagat_backward")
        # back through the caller-side GEMM  K = H W_k^T + b_k
        d_h = torch.addmm(d_h.float(), d_k, in_w[E:2 * E]).to(bag_data.dtype)
        d_in_w[E:2 * E] = torch.mm(d_k.t(), bag_data.float())
        d_in_b[E:2 * E] = d_k.sum(0)
        return d_query, d_h, d_in_w, d_in_b, d_out_w, d_out_b, None, None


def coattn_nacagat(query, batch: BagBatch, in_w, in_b, out_w, out_b, drop_p: float):
    """query (n_slides*n_q, E) -> (q_proj, attn_out (n_slides*n_q, E), ragged post-dropout map)."""
    return CoAttnNaCAGaTFn.apply(query, batch.data, in_w, in_b, out_w, out_b, batch, drop_p)
