"""Fusion layers of the tail (reference: models/fusion.py).  Only `concat`, the reference default
(models/mcat/config/config.yaml:43), is built so far; `bilinear` / `gated_concat` are SURVEY 8(f) rows."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class ConcatFusion(nn.Module):
    """cat -> Linear+ReLU -> Linear+ReLU; replaces models/fusion.py:7-19 (state_dict fusion_layer.0.*, .2.*).
    forward(*x): each x (d_i,) for one slide, or (B, d_i) for a window."""

    def __init__(self, dims: list, hidden_size: int = 256, output_size: int = 256):
        super().__init__()
        self.fusion_layer = nn.Sequential(nn.Linear(sum(dims), hidden_size), nn.ReLU(),
                                          nn.Linear(hidden_size, output_size), nn.ReLU())

    def forward(self, *x):
        h = torch.cat(x, dim=-1)
        h = ops.linear(h, self.fusion_layer[0].weight, self.fusion_layer[0].bias, "relu")
        return ops.linear(h, self.fusion_layer[2].weight, self.fusion_layer[2].bias, "relu")
