"""Set-Transformer over the N omic-guided tokens: drop-in for
nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model, nhead=8, dim_feedforward=512, dropout, 'relu'), 2)
(models/mcat/mcat.py:51-53,60-62).  Sub-classing the stock classes keeps constructor, init and
state_dict (layers.{i}.self_attn.in_proj_weight, ...linear1..., norm1...) identical; only forward()
is replaced (post-norm, torch/nn/modules/transformer.py:661 with norm_first=False)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class SetTransformerEncoder(nn.TransformerEncoder):
    def __init__(self, encoder_layer, num_layers, norm=None, **kw):
        kw.setdefault("enable_nested_tensor", False)
        super().__init__(encoder_layer, num_layers, norm=norm, **kw)
        if norm is not None:
            raise NotImplementedError("the reference uses no final norm")

    def forward(self, src: torch.Tensor, mask=None, src_key_padding_mask=None, is_causal=None) -> torch.Tensor:
        """src (T, d): one slide (the reference's call, models/mcat/mcat.py:101), or (B, T, d): a window."""
        if mask is not None or src_key_padding_mask is not None:
            raise NotImplementedError("masks are never used on this path")
        x = src if src.dim() == 3 else src.unsqueeze(0)
        x = ops.encoder(x, list(self.layers), self.training)
        return x if src.dim() == 3 else x[0]


def make_set_transformer(d_model: int, dropout: float, nhead: int = 8, dim_feedforward: int = 512, num_layers: int = 2):
    layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward,
                                       dropout=dropout, activation="relu")
    return SetTransformerEncoder(layer, num_layers=num_layers)
