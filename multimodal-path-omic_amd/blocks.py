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

    def forward_window(self, query: torch.Tensor, bags: BagBatch, need_weights: bool = False):
        """query (n_slides, N, E) -> out (n_slides, N, E), list of (N, M_b) maps or None."""
        n_slides, n_q, e = query.shape
        out, amap = ops.coattn_mcat(query.reshape(n_slides * n_q, e), bags, self.in_proj_weight, self.in_proj_bias,
                                    self.out_proj.weight, self.out_proj.bias, need_weights)
        return out.view(n_slides, n_q, e), (bags.split_map(amap, n_q) if need_weights else None)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, need_weights: bool = True,
                **unused) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        bag = _as_bag(key, value)
        out, maps = self.forward_window(query.unsqueeze(0), BagBatch.from_list([bag]), need_weights)
        return out[0], (maps[0] if need_weights else None)
