"""Drop-in nn.Modules for the attention slots of the reference models.

Each class keeps the constructor arguments, parameter names/shapes (state_dict keys) and forward()
signature of the module it replaces, so a reference checkpoint loads with
load_state_dict and a reference model works after `model.co_attention = CoAttention(...)` etc.
Every forward runs hand-written HIP kernels through ops.py; there is no eager fallback.

Besides the reference's per-slide signature each module has `forward_window(...)`, which takes a
whole gradient-accumulation window of slides as one ragged batch (ops.BagBatch): the arithmetic per
slide is identical, the launches are shared.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .ops import BagBatch


class _Proj(nn.Module):
    """Parameter holder with nn.Linear's names (weight, bias); used where the reference has a
    (NonDynamicallyQuantizable)Linear sub-module whose arithmetic is fused into a kernel."""

    def __init__(self, in_features, out_features, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.zeros_(self.bias)

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


def _as_bag(key, value) -> torch.Tensor:
    if key is not value and not (key.data_ptr() == value.data_ptr() and key.shape == value.shape):
        raise NotImplementedError("co-attention kernels require key and value to be the same bag tensor "
                                  "(the reference always passes H_bag twice: models/mcat/mcat.py:97)")
    if key.dim() != 2:
        raise ValueError(f"bag must be (M, E), got {tuple(key.shape)}")
    return key


class CoAttention(nn.Module):
    """MCAT's genomic-guided co-attention: replaces `nn.MultiheadAttention(embed_dim, num_heads=1)`
    at models/mcat/mcat.py:48.  Parameters: in_proj_weight (3E,E), in_proj_bias (3E),
    out_proj.weight (E,E), out_proj.bias (E) -- nn.MultiheadAttention's own names and init.

    forward(query=(N,E), key=bag, value=bag, need_weights=bool) -> (out (N,E), A (N,M) | None),
    the keyword-only call of models/mcat/mcat.py:97.
    """

    def __init__(self, embed_dim: int, num_heads: int = 1, dropout: float = 0.0, device=None, dtype=None):
        super().__init__()
        if num_heads != 1:
            raise NotImplementedError("the reference only ever uses num_heads=1 (models/mcat/mcat.py:48)")
        if dropout != 0.0:
            raise NotImplementedError("MCAT's co-attention has no attention dropout (module default 0.0)")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim, device=device, dtype=dtype))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim, device=device, dtype=dtype))
        self.out_proj = _Proj(embed_dim, embed_dim, device=device, dtype=dtype)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.in_proj_bias)

    def forward_window(self, query: torch.Tensor, bags: BagBatch, need_weights: bool = False, bag_relu_gate: float = 0.0):
        """query (n_slides, N, E) -> out (n_slides, N, E), list of (N, M_b) maps or None."""
        n_slides, n_q, e = query.shape
        out, amap = ops.coattn_mcat(query.reshape(n_slides * n_q, e), bags, self.in_proj_weight, self.in_proj_bias,
                                    self.out_proj.weight, self.out_proj.bias, need_weights, bag_relu_gate)
        return out.view(n_slides, n_q, e), (bags.split_map(amap, n_q) if need_weights else None)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, need_weights: bool = True,
                **unused) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        bag = _as_bag(key, value)
        out, maps = self.forward_window(query.unsqueeze(0), BagBatch.from_list([bag]), need_weights)
        return out[0], (maps[0] if need_weights else None)


class AttentionNetGated(nn.Module):
    """Gated attention-MIL scorer; replaces models/blocks.py:13-48 (same constructor, same
    state_dict: attention_a.0.*, attention_b.0.*, attention_c.*).  forward(x (L,D)) -> (A (L,n_classes), x).

    A = W_c [ Drop(tanh(W_a x)) * Drop(sigmoid(W_b x)) ];  both dropouts are p = 0.25 (hard-wired at :34-36).
    """

    def __init__(self, input_dim: int = 256, hidden_dim: int = 256, dropout_p: bool = True, n_classes: int = 1):
        super().__init__()
        a = [nn.Linear(input_dim, hidden_dim), nn.Tanh()]
        b = [nn.Linear(input_dim, hidden_dim), nn.Sigmoid()]
        if dropout_p:
            a.append(nn.Dropout(0.25))
            b.append(nn.Dropout(0.25))
        self.attention_a = nn.Sequential(*a)          # parameter holders: arithmetic runs in ops.*
        self.attention_b = nn.Sequential(*b)
        self.attention_c = nn.Linear(hidden_dim, n_classes)
        self.drop_p = 0.25 if dropout_p else 0.0

    def scores(self, x: torch.Tensor) -> torch.Tensor:
        """x (..., L, D) -> raw scores (..., L, n_classes)."""
        return ops.gated_scores(x, self.attention_a[0].weight, self.attention_a[0].bias,
                                self.attention_b[0].weight, self.attention_b[0].bias,
                                self.attention_c.weight, self.attention_c.bias,
                                self.drop_p if self.training else 0.0)

    def forward(self, x: torch.Tensor):
        return self.scores(x), x


class ContextualAttentionGate(nn.Module):
    """NaCAGaT's Contextual Attention Gate; replaces models/blocks.py:232-253.
    G = LN(ELU(ELU(fc1 Q) + ELU(fc2 Qh))),  E = LN(ELU(ELU(fc3 Qh))),  C = ELU(fc_c(G * E)).
    state_dict: fc1.0.*, fc2.0.*, fc3.0.*, G.1.*, E.1.*, fc_c.0.*"""

    def __init__(self, dim: int = 256, hidden_dim: int = 128):
        super().__init__()
        self.fc1 = nn.Sequential(nn.Linear(dim, hidden_dim), nn.ELU())
        self.fc2 = nn.Sequential(nn.Linear(dim, hidden_dim), nn.ELU())
        self.fc3 = nn.Sequential(nn.Linear(dim, hidden_dim), nn.ELU())
        self.G = nn.Sequential(nn.ELU(), nn.LayerNorm(hidden_dim))
        self.E = nn.Sequential(nn.ELU(), nn.LayerNorm(hidden_dim))
        self.fc_c = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.ELU())

    def forward(self, Q: torch.Tensor, Q_hat: torch.Tensor) -> torch.Tensor:
        return ops.contextual_gate(Q, Q_hat, self)


class PreGatingContextualAttention(nn.Module):
    """NaCAGaT's narrow-gated co-attention + CAG; replaces models/blocks.py:51-111.
    Same constructor (embed_dim, num_heads, device, dtype, dropout_p=0.25), same parameters
    (in_proj_weight/bias, out_proj.*, CAG.*), Xavier-uniform in-projection and zero biases (:81-90).
    forward(query, key, value) -> (attn_out + CAG(query, q_proj), A (N,M)); A is post-dropout in
    training (:189-190, :206)."""

    def __init__(self, embed_dim, num_heads, device=None, dtype=None, dropout_p: float = 0.25):
        super().__init__()
        if num_heads != 1:
            raise NotImplementedError("the reference module only works with num_heads=1 (SURVEY 3.3)")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout_p
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim, device=device, dtype=dtype))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim, device=device, dtype=dtype))
        self.out_proj = _Proj(embed_dim, embed_dim, device=device, dtype=dtype)
        self.CAG = ContextualAttentionGate(dim=embed_dim, hidden_dim=embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.in_proj_bias)
        nn.init.zeros_(self.out_proj.bias)

    def forward_window(self, query: torch.Tensor, bags: BagBatch, bag_relu_gate: float = 0.0, pair=None):
        """pair (ops.TokenPair): attn_out + CAG is produced into pair.slot(0) by CAG's last launch and the query is handed
        through both ops, so the window step has no element-wise add, no stack copy and no gradient adds for the three
        uses of the omic tokens; a third result is then the query to give to the omic branch."""
        n_slides, n_q, e = query.shape
        q2 = query.reshape(n_slides * n_q, e)
        drop = self.dropout if self.training else 0.0
        if pair is None:
            q_proj, out, amap = ops.coattn_nacagat(q2, bags, self.in_proj_weight, self.in_proj_bias,
                                                   self.out_proj.weight, self.out_proj.bias, drop, bag_relu_gate)
            total = ops.contextual_gate(q2, q_proj, self.CAG, residual=out)
            return total.view(n_slides, n_q, e), bags.split_map(amap, n_q)
        q_proj, out, amap, q_on = ops.coattn_nacagat(q2, bags, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight,
                                                     self.out_proj.bias, drop, bag_relu_gate, hand_on=True)
        total, q_on = ops.contextual_gate(q_on, q_proj, self.CAG, residual=out, dest=(pair, 0),
                                          hand_on=True)
        return total.view(n_slides, n_q, e), bags.split_map(amap, n_q), q_on.view(n_slides, n_q, e)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, **unused):
        bag = _as_bag(key, value)
        out, maps = self.forward_window(query.unsqueeze(0), BagBatch.from_list([bag]))
        return out[0], maps[0]
