"""Fusion layers of the tail (reference: models/fusion.py): `concat` (the reference default,
models/mcat/config/config.yaml:43; fused with the classifier and survival head into K6 on the window path),
`gated_concat` and `bilinear`.  Every Linear runs on the HIP GEMM (`ops.linear`); the Kronecker
product and the gates are a handful of element-wise device ops.  forward(*x): each x (d_i,) for one slide -- the
reference's call, models/mcat/mcat.py:119-124 -- or (B, d_i) for a window of slides."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _lin(x, layer: nn.Linear, act: str = "none"):
    return ops.linear(x, layer.weight, layer.bias, act)


class ConcatFusion(nn.Module):
    """cat -> Linear+ReLU -> Linear+ReLU; replaces models/fusion.py:7-19 (state_dict fusion_layer.0.*, .2.*)."""

    def __init__(self, dims: list, hidden_size: int = 256, output_size: int = 256):
        super().__init__()
        self.fusion_layer = nn.Sequential(nn.Linear(sum(dims), hidden_size), nn.ReLU(),
                                          nn.Linear(hidden_size, output_size), nn.ReLU())

    def forward(self, *x):
        h = torch.cat(x, dim=-1)
        h = _lin(h, self.fusion_layer[0], "relu")
        return _lin(h, self.fusion_layer[2], "relu")


class GatedConcatFusion(nn.Module):
    """x_i * sigmoid(Linear_i(x_i)) -> cat -> the ConcatFusion MLP; replaces models/fusion.py:22-41.

    The reference keeps its gates in a plain Python list (`fusion.py:25-27`): they are never registered, never moved
    to the device, never trained and never saved.  Here they are a registered ModuleList `gates`;
    a reference checkpoint -- which has no gate entries -- still loads strictly: missing `gates.*` keys keep their
    freshly initialised values, which is what the reference model had."""

    def __init__(self, dims: list, hidden_size: int = 256, output_size: int = 256):
        super().__init__()
        self.gates = nn.ModuleList([nn.Sequential(nn.Linear(dim, 1), nn.Sigmoid()) for dim in dims])
        self.fusion_layer = nn.Sequential(nn.Linear(sum(dims), hidden_size), nn.ReLU(),
                                          nn.Linear(hidden_size, output_size), nn.ReLU())
        self._register_load_state_dict_pre_hook(self._keep_gates_if_absent)

    def _keep_gates_if_absent(self, state_dict, prefix, *args):
        for name, tensor in self.gates.state_dict().items():
            state_dict.setdefault(prefix + "gates." + name, tensor)

    def forward(self, *x):
        items = [item * _lin(item, gate[0], "sigmoid") for gate, item in zip(self.gates, x)]
        h = torch.cat(items, dim=-1)
        h = _lin(h, self.fusion_layer[0], "relu")
        return _lin(h, self.fusion_layer[2], "relu")


class BilinearFusion(nn.Module):
    """Gated bilinear (Kronecker) fusion; replaces models/fusion.py:44-113, same constructor, parameter names and
    `init_max_weights` initialisation (models/utils.py:43-48: N(0, 1/sqrt(fan_in)) weights, zero biases, Linear only)."""

    def __init__(self, dim1: int = 256, dim2: int = 256, hidden_size: int = 32, output_size: int = 64,
                 mm_hidden_size: int = 64, use_skip_connection=True, use_bilinear=True, use_gates=True, dropout=0.25):
        super().__init__()
        self.use_skip_connection, self.use_bilinear, self.use_gates = use_skip_connection, use_bilinear, use_gates
        self.linear_h1 = nn.Sequential(nn.Linear(dim1, hidden_size), nn.ReLU())
        self.linear_z1 = nn.Bilinear(dim1, dim2, hidden_size) if use_bilinear else nn.Linear(dim1 + dim2, hidden_size)
        self.linear_o1 = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.ReLU(), nn.Dropout(p=dropout))
        self.linear_h2 = nn.Sequential(nn.Linear(dim2, hidden_size), nn.ReLU())
        self.linear_z2 = nn.Bilinear(dim2, dim1, hidden_size) if use_bilinear else nn.Linear(dim2 + dim1, hidden_size)
        self.linear_o2 = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.ReLU(), nn.Dropout(p=dropout))
        self.post_fusion_dropout = nn.Dropout(p=dropout)
        self.fc1 = nn.Sequential(nn.Linear((hidden_size + 1) * (hidden_size + 1), mm_hidden_size), nn.ReLU(), nn.Dropout(p=dropout))
        self.fc2 = nn.Sequential(nn.Linear(mm_hidden_size + (hidden_size * 2) + 2, output_size), nn.ReLU(), nn.Dropout(p=dropout))
        for m in self.modules():
            if type(m) is nn.Linear:
                m.weight.data.normal_(0, 1.0 / m.weight.size(1) ** 0.5)
                m.bias.data.zero_()

    def _z(self, layer, a, b):
        """nn.Bilinear(a, b) as one GEMM: z[., k] = sum_i a[., i] (sum_j W[k, i, j] b[., j]) + bias[k]."""
        if not self.use_bilinear:
            return _lin(torch.cat((a, b), dim=-1), layer)
        k, i, j = layer.weight.shape
        y = ops.linear(b, layer.weight.view(k * i, j), None).view(-1, k, i)
        return (y * a.unsqueeze(1)).sum(-1) + layer.bias

    def _branch(self, h_layer, z_layer, o_layer, a, b):
        if self.use_gates:
            gated = torch.sigmoid(self._z(z_layer, a, b)) * _lin(a, h_layer[0], "relu")
        else:
            gated = a
        return F.dropout(_lin(gated, o_layer[0], "relu"), o_layer[2].p, self.training)

    def forward(self, *x):
        if len(x) != 2:
            raise RuntimeError("Bilinear fusion is possible only on 2 inputs")
        single = x[0].dim() == 1
        x1, x2 = (t.reshape(1, -1) if single else t for t in x)
        o1 = self._branch(self.linear_h1, self.linear_z1, self.linear_o1, x1, x2)
        o2 = self._branch(self.linear_h2, self.linear_z2, self.linear_o2, x2, x1)
        ones = torch.ones(o1.shape[0], 1, device=o1.device, dtype=o1.dtype)
        o1, o2 = torch.cat((o1, ones), 1), torch.cat((o2, ones), 1)
        out = (o1.unsqueeze(2) * o2.unsqueeze(1)).flatten(start_dim=1)          # Kronecker product, (B, 33 * 33)
        out = F.dropout(out, self.post_fusion_dropout.p, self.training)
        out = F.dropout(_lin(out, self.fc1[0], "relu"), self.fc1[2].p, self.training)
        if self.use_skip_connection:
            out = torch.cat((out, o1, o2), 1)
        out = F.dropout(_lin(out, self.fc2[0], "relu"), self.fc2[2].p, self.training)
        return out.squeeze(0) if single else out
