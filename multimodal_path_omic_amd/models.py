"""The models of the hot path, composed from the drop-in layers (host composition = row H1 of
SURVEY.md section 8(a)).  Class names, constructor arguments, attribute names (= state_dict keys) and
forward() signatures follow models/mcat/mcat.py:12-142 and models/nacagat/nacagat.py:9-138, so a
reference checkpoint loads with load_state_dict(strict=True) and the reference's train/validate/test
loops can call these models unchanged.

Beyond the reference's one-slide forward(), forward_window() pushes a whole gradient-accumulation
window of slides through the model as ONE ragged batch: slides are independent (batch size 1, no
cross-slide state: models/mcat/main.py:250), so per-slide results are unchanged while every launch is
shared by the window.  `bag_dtype=torch.bfloat16` stores the patch matrix / H_bag in bf16 (a storage
format: accumulation stays fp32, the 6 x d tail stays fp32 -- SURVEY.md section 0.7).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import AttentionNetGated, CoAttention, PreGatingContextualAttention
from .fusion import BilinearFusion, ConcatFusion, GatedConcatFusion
from . import ops
from .ops import BagBatch
from .transformer import make_set_transformer

MODEL_SIZES = {"small": [128, 128], "medium": [256, 256], "big": [512, 512]}


class _FusionModelBase(nn.Module):
    def __init__(self, omic_sizes: Sequence[int], model_size: str, n_classes: int, dropout: float, fusion: str,
                 device: str, bag_dtype: torch.dtype):
        super().__init__()
        self.n_classes = n_classes
        self.model_sizes = MODEL_SIZES[model_size]
        d0, d1 = self.model_sizes
        self.bag_dtype = bag_dtype
        # H: patch fully-connected layer (stock modules; fusing it into K1 is SURVEY 8(f) f1)
        self.H = nn.Sequential(nn.Linear(1024, d0), nn.ReLU(), nn.Dropout(dropout))
        # G: one 2-layer SNN per omic group
        self.G = nn.ModuleList([
            nn.Sequential(
                nn.Sequential(nn.Linear(s, d0), nn.ELU(), nn.AlphaDropout(p=dropout, inplace=False)),
                nn.Sequential(nn.Linear(d0, d1), nn.ELU(), nn.AlphaDropout(p=dropout, inplace=False)))
            for s in omic_sizes])
        self.co_attention = self._make_co_attention(d1)
        self.path_transformer = make_set_transformer(d1, dropout)
        self.path_attention_head = AttentionNetGated(n_classes=1, input_dim=d1, hidden_dim=d1)
        self.path_rho = nn.Sequential(nn.Linear(d1, d1), nn.ReLU(), nn.Dropout(dropout))
        self.omic_transformer = make_set_transformer(d1, dropout)
        self.omic_attention_head = AttentionNetGated(n_classes=1, input_dim=d1, hidden_dim=d1)
        self.omic_rho = nn.Sequential(nn.Linear(d1, d1), nn.ReLU(), nn.Dropout(dropout))
        self.fusion = fusion
        if fusion == "concat":
            self.fusion_layer = ConcatFusion(dims=[d1, d1], hidden_size=d1, output_size=d1).to(device=device)
        elif fusion == "bilinear":            # models/mcat/mcat.py:73-74
            self.fusion_layer = BilinearFusion(dim1=d1, dim2=d1, output_size=d1)
        elif fusion == "gated_concat":        # :75-77
            self.fusion_layer = GatedConcatFusion(dims=[d1, d1], hidden_size=d1, output_size=d1).to(device=device)
        else:
            raise RuntimeError(f"Fusion mechanism {fusion} not implemented")
        self.classifier = nn.Linear(d1, n_classes)

    # ---- pieces
    # the co-attention kernel can hand back d(H_bag) already multiplied by the ReLU/dropout derivative
    _fused_bag_gate = False

    def _patch_fc(self, bags: BagBatch) -> BagBatch:
        x = bags.data
        lin = self.H[0]
        p = self.H[2].p if self.training else 0.0
        if x.dtype == torch.bfloat16:
            h = ops.patch_fc(x, lin.weight, lin.bias, p, pre_gated_grad=self._fused_bag_gate, batch=bags)
        elif ops.patch_fc_f32_supported(x, lin.weight):
            h = ops.patch_fc_f32(x, lin.weight, lin.bias, p)        # fp32 window, 1024 -> 256: hand-written both ways
        else:                                                       # fp32 window of the small / big models: the exact-fp32 MFMA GEMM (many-row form)
            h = F.dropout(ops.linear(x.float(), lin.weight, lin.bias, "relu"), p, self.training)
        return bags.with_data(h)

    def _token_pair(self, bags: BagBatch, omics):
        return None

    def _patch_and_co_attend(self, g_bag, bags: BagBatch, inference: bool, pair=None):
        """Patch layer + co-attention (models/mcat/mcat.py:87,97) -> (co-attended tokens, map, the omic tokens to hand to
        the omic branch).  Subclasses with a fused kernel override this."""
        h, a = self._co_attend(g_bag, self._patch_fc(bags), inference)
        return h, a, g_bag

    def _omic_fc(self, omics: "List[torch.Tensor]", tokens=None) -> torch.Tensor:
        """omics: per group a (B, d_i) tensor -> G_bag (B, N, d)."""
        return ops.omic_snn(omics, self.G, self.training, tokens)

    # ---- window API
    def forward_window(self, bags: BagBatch, omics: "List[torch.Tensor]", inference: bool = False, ces_targets=None):
        """bags: raw patch features (total_rows, 1024) of the window; omics: per group (B, d_i).
        Returns hazards, survs, Y (B, C) and {'coattn': [ (N, M_b) ] | None, 'path': (B,1,N), 'omic': (B,1,N)}.
        ces_targets = (labels, censorship, slide_weight) (training step, fusion 'concat'): the `ces` loss and its backward
        ride in the head's launch (ops.fusion_head_loss_cat); the dict gains 'loss' and 'risk' (per slide), and
        backward must be driven as loss.backward(slide_weight).

        The path and the omic set-Transformer / pooling head have identical geometry and run as ONE launch sequence
        with grouped GEMMs (ops.encoder_stacked, ops.gated_pool_stacked): the token tail is a latency-bound chain of
        small launches, so the omic branch rides along in launches the path branch needs anyway."""
        pair = self._token_pair(bags, omics)
        g_bag = self._omic_fc(omics, pair)
        h_coattn, a_coattn, g_tok = self._patch_and_co_attend(g_bag, bags, inference, pair)
        stacked = pair.stack(h_coattn, g_tok).view(2, *h_coattn.shape) if pair is not None else torch.stack([h_coattn, g_tok])
        tokens = ops.encoder_stacked(stacked, [list(self.path_transformer.layers), list(self.omic_transformer.layers)],
                                     self.training)
        concat = self.fusion == "concat"
        a, h = ops.gated_pool_stacked(tokens, [self.path_attention_head, self.omic_attention_head],
                                      [self.path_rho, self.omic_rho], self.training, interleave=concat)
        att = {"coattn": a_coattn, "path": a[0], "omic": a[1]}
        if ces_targets is not None:
            if not concat:
                raise ValueError("forward_window(ces_targets=...) is built for fusion 'concat'")
            att["loss"], att["risk"], hazards, survs, y = ops.fusion_head_loss_cat(h, self.fusion_layer, self.classifier,
                                                                                   *ces_targets)
            return hazards, survs, y, att
        if concat:                              # h IS (B, [h_path | h_omic]): the pooling launch wrote it interleaved
            hazards, survs, y = ops.fusion_head_cat(h, self.fusion_layer, self.classifier)
        else:
            hazards, survs, y = self._fuse_and_head(h[0], h[1])
        return hazards, survs, y, att

    def _fuse_and_head(self, h_path, h_omic):
        """Fusion + classifier + survival head (models/mcat/mcat.py:119-138).  `concat` is one K6 call; the other fusion
        layers (row f4) run their own forward, then the classifier GEMM and the HIP head."""
        if self.fusion == "concat":
            return ops.fusion_head(h_path, h_omic, self.fusion_layer, self.classifier)
        fused = self.fusion_layer(h_path, h_omic)
        return ops.survival_head(ops.linear(fused, self.classifier.weight, self.classifier.bias))

    def _forward_one(self, wsi, omics, inference):
        """The reference's call: wsi (1,M,1024) or (M,1024); omics list of (1,d_i) or (d_i,)."""
        x = wsi.squeeze(0) if wsi.dim() == 3 else wsi
        if x.dtype != self.bag_dtype:
            x = x.to(self.bag_dtype)
        bags = BagBatch.from_list([x])
        om = [o.reshape(1, -1) for o in omics]
        hazards, survs, y, att = self.forward_window(bags, om, inference)
        co = att["coattn"][0] if att["coattn"] is not None else None
        return hazards, survs, y, {"coattn": co, "path": att["path"][0], "omic": att["omic"][0]}

    def get_trainable_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class MultimodalCoAttentionTransformer(_FusionModelBase):
    """MCAT (models/mcat/mcat.py:12).  forward(wsi, omics, inference=False)."""

    def __init__(self, omic_sizes: [], model_size: str = "medium", n_classes: int = 4, dropout: float = 0.25,
                 fusion: str = "concat", device: str = "cpu", bag_dtype: torch.dtype = torch.float32):
        super().__init__(omic_sizes, model_size, n_classes, dropout, fusion, device, bag_dtype)

    def _make_co_attention(self, d):
        return CoAttention(embed_dim=d, num_heads=1)

    _fused_bag_gate = True

    def _co_attend(self, g_bag, h_bags, inference):
        gate = 0.0
        if h_bags.data.dtype == torch.bfloat16:
            gate = getattr(h_bags.data, "_mpo_keep_scale", 1.0 / (1.0 - self.H[2].p)) if self.training else 1.0
        return self.co_attention.forward_window(g_bag, h_bags, need_weights=inference, bag_relu_gate=gate)

    def _token_pair(self, bags: BagBatch, omics):
        """Row f1 writes the co-attention output and the omic tokens straight into the (2, B, N, d) buffer the
        branch-batched encoder reads (no torch.stack copy, no gradient add for the doubly-used G_bag)."""
        n_q, e = len(omics), self.H[0].out_features
        if not ops.fused_patch_coattn_supported(bags.data, e, n_q):
            return None
        return ops.TokenPair(bags.n_slides * n_q, e, bags.data.device)

    def _patch_and_co_attend(self, g_bag, bags: BagBatch, inference: bool, pair=None):
        n_slides, n_q, e = g_bag.shape
        if pair is None:
            return super()._patch_and_co_attend(g_bag, bags, inference)
        # row f1: ONE pass over the raw patch matrix (patch layer on the MFMA, co-attention while the tile is in LDS)
        lin, co = self.H[0], self.co_attention
        out, amap, _, g_tok = ops.patch_coattn_mcat(bags.data, bags, lin.weight, lin.bias, self.H[2].p if self.training else 0.0,
                                                    g_bag.reshape(n_slides * n_q, e), co.in_proj_weight, co.in_proj_bias,
                                                    co.out_proj.weight, co.out_proj.bias, inference, pair)
        return out.view(n_slides, n_q, e), (bags.split_map(amap, n_q) if inference else None), g_tok.view(n_slides, n_q, e)

    def forward(self, wsi, omics, inference: bool = False):
        return self._forward_one(wsi, omics, inference)


class NarrowContextualAttentionGateTransformer(_FusionModelBase):
    """NaCAGaT (models/nacagat/nacagat.py:9).  forward(wsi, omics): the map is always returned (:93)."""

    def __init__(self, omic_sizes: [], model_size: str = "medium", n_classes: int = 4, dropout: float = 0.25,
                 fusion: str = "concat", device: str = "cpu", bag_dtype: torch.dtype = torch.float32):
        super().__init__(omic_sizes, model_size, n_classes, dropout, fusion, device, bag_dtype)

    def _make_co_attention(self, d):
        return PreGatingContextualAttention(embed_dim=d, num_heads=1)

    @property
    def _fused_bag_gate(self):
        # (the kernel that finishes d_bag with the patch layer's ReLU / dropout derivative is built for embed_dim <= 256; 'big'
        #  runs the column-half passes of csrc/capi.hip and leaves that derivative to the patch layer's own backward)
        return self.model_sizes[1] != 512

    def _bag_gate(self, h_bags):
        if h_bags.data.dtype == torch.bfloat16 and self._fused_bag_gate:
            return getattr(h_bags.data, "_mpo_keep_scale", 1.0 / (1.0 - self.H[2].p)) if self.training else 1.0
        return 0.0

    def _co_attend(self, g_bag, h_bags, inference):
        return self.co_attention.forward_window(g_bag, h_bags, bag_relu_gate=self._bag_gate(h_bags))

    def _token_pair(self, bags: BagBatch, omics):
        """attn_out + CAG and the omic tokens are produced straight into the (2, B, N, d) buffer the branch-batched encoder reads
        (no element-wise add, no torch.stack copy, no gradient adds for the three consumers of G_bag)."""
        return ops.TokenPair(bags.n_slides * len(omics), self.H[0].out_features, bags.data.device)

    def _patch_and_co_attend(self, g_bag, bags: BagBatch, inference: bool, pair=None):
        if pair is None:
            return super()._patch_and_co_attend(g_bag, bags, inference)
        h_bags = self._patch_fc(bags)
        return self.co_attention.forward_window(g_bag, h_bags, bag_relu_gate=self._bag_gate(h_bags), pair=pair)

    def forward(self, wsi, omics):
        return self._forward_one(wsi, omics, True)


class GeneExprNarrowContextualAttentionGateTransformer(nn.Module):
    """Row f3: the gene-expression model of models/ge_nacagat/ge_nacagat.py:9-72 -- patch layer, ONE-head self-attention
    over the M patch rows with its M x M map returned, the 2-layer / 8-head set-Transformer over the same M rows, gated
    attention-MIL pooling over L = M, rho, classifier, softmax.  Constructor, attribute names (= state_dict keys) and
    forward() follow the reference; `bag_dtype` is this package's storage switch for the patch matrix.

    Long-axis pieces: ops.bag_self_attention (no M x M state except the returned map), the encoder and the pooling head
    of the 6-token tail called with T = L = M (csrc/tail_api.hip picks the long-axis kernels)."""

    def __init__(self, model_size: str = "medium", n_classes: int = 3, dropout: float = 0.25,
                 bag_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.model_sizes = MODEL_SIZES[model_size]
        d0, d1 = self.model_sizes
        self.bag_dtype = bag_dtype
        self.H = nn.Sequential(nn.Linear(1024, d0), nn.ReLU(), nn.Dropout(dropout))
        self.self_attention = nn.MultiheadAttention(embed_dim=d1, num_heads=1)      # parameter holder
        self.path_transformer = make_set_transformer(d1, dropout)
        self.path_attention_head = AttentionNetGated(n_classes=1, input_dim=d1, hidden_dim=d1)
        self.path_rho = nn.Sequential(nn.Linear(d1, d1), nn.ReLU(), nn.Dropout(dropout))
        self.classifier = nn.Linear(d1, n_classes)

    _patch_fc = _FusionModelBase._patch_fc
    _fused_bag_gate = False

    def forward(self, wsi):
        """wsi (M, 1024) or (1, M, 1024) -> Y (n_classes,), {'attn': (M, M), 'path': (1, M)}   (ge_nacagat.py:43-72)."""
        x = wsi.squeeze(0) if wsi.dim() == 3 else wsi
        if x.dtype != self.bag_dtype:
            x = x.to(self.bag_dtype)
        h_bag = self._patch_fc(BagBatch.from_list([x])).data.float()
        h_coattn, a_coattn = ops.bag_self_attention(h_bag, self.self_attention, self.training, need_weights=True)
        path_trans = self.path_transformer(h_coattn)
        a_path, h_path = ops.gated_pool(path_trans.unsqueeze(0), self.path_attention_head, self.path_rho, self.training)
        logits = ops.linear(h_path, self.classifier.weight, self.classifier.bias)[0]
        y = torch.softmax(logits, dim=0)          # F.softmax without dim on a vector (ge_nacagat.py:67)
        return y, {"attn": a_coattn, "path": a_path[0]}

    def get_trainable_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
