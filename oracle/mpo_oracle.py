"""CPU ORACLE -- test infrastructure, NOT product code.

A plain restatement, in explicit torch-CPU matrix arithmetic (fp32 by default,
fp64 on request), of the reference's WSI-patch x omics fusion path.  It is written
from the mathematics of SURVEY.md section 8(a), one function per scope-table row,
each citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product package
(multimodal-path-omic_amd/) never does and fails loudly without its HIP library.

Pinning: tests/test_oracle_golden.py checks every function here against
tests/golden/*.npz, which tests/golden/make_golden.py generated in the authoring
container by importing the reference itself from /root/reference (torch 2.10 CPU).
Harrell's C-index is the one exception: scikit-survival is absent from the image,
so `concordance_index_censored` below restates its published algorithm and is
"parity unpinned" against the library (hand-computed cases only).

All functions are eval-mode by default (no dropout).  Where the reference drops
out in training, an explicit keep-mask (already scaled by 1/(1-p)) can be passed so
that a GPU run's own mask can be replayed exactly.

Parameters are passed as a flat {name: tensor} dict using the reference's
state_dict names, plus a prefix.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- helpers
def _lin(x, p, name):
    """y = x W^T + b with W = p[name+'.weight'] (out,in)."""
    return x @ p[name + ".weight"].t() + p[name + ".bias"]


def _layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _elu(x):
    return torch.where(x > 0, x, torch.expm1(x))


# --------------------------------------------------------------------------- H2
def _store(t, dtype):
    """Round-trip through a storage dtype (identity gradient): emulates a tensor kept in bf16."""
    return t if dtype is None else t.to(dtype).float()


def patch_fc(wsi, p, prefix="H", keep=None, storage=None, round_gemm_out=True, storage_points=("x", "w", "h")):
    """H_bag = Drop(ReLU(X W_H^T + b)); models/mcat/mcat.py:24-29,87.
    storage=torch.bfloat16 emulates the product's bf16 STORAGE points with fp32 arithmetic in between, so a bf16-stored
    run can be checked tightly: patch matrix, GEMM weight operand, H_bag -- and, with round_gemm_out, the GEMM output
    before the bias (the library-GEMM path of the small / big models; the hand-written patch-layer kernel of the medium
    models keeps the accumulator in fp32 up to the one rounding of H_bag).
    storage_points: which of the three storage points are rounded -- "x" the patch matrix, "w" the weight operand,
    "h" H_bag (and the GEMM output under round_gemm_out) -- so a test can attribute a bf16-mode error to each of them."""
    x = wsi.squeeze(0) if wsi.dim() == 3 else wsi
    x = _store(x.float(), storage if "x" in storage_points else None)
    h = x @ _store(p[prefix + ".0.weight"], storage if "w" in storage_points else None).t()
    st_h = storage if "h" in storage_points else None
    if round_gemm_out:
        h = _store(h, st_h)
    h = _store(torch.relu(h + p[prefix + ".0.bias"]), st_h)
    return h if keep is None else h * keep


def omic_fc(omics, p, prefix="G"):
    """Per-group SNN: 2 x (Linear + ELU [+ AlphaDropout]); models/mcat/mcat.py:32-45,90-92."""
    rows = []
    for i, o in enumerate(omics):
        x = o.float().reshape(1, -1)
        x = _elu(_lin(x, p, f"{prefix}.{i}.0.0"))
        x = _elu(_lin(x, p, f"{prefix}.{i}.1.0"))
        rows.append(x)
    return torch.cat(rows, 0)                                   # (N, d)


# --------------------------------------------------------------------------- H3
def mcat_coattention(query, bag, p, prefix="co_attention", need_weights=True):
    """nn.MultiheadAttention(E, heads=1) cross-attention, key is value is the bag.

    models/mcat/mcat.py:48,97; arithmetic per torch/nn/functional.py:6206-6660
    (packed in-projection, q scaled by 1/sqrt(E), softmax over patches, out_proj).
    Returns (out (N,E), A (N,M) or None).
    """
    e = query.shape[-1]
    w, b = p[prefix + ".in_proj_weight"], p[prefix + ".in_proj_bias"]
    q = query @ w[:e].t() + b[:e]
    k = bag @ w[e:2 * e].t() + b[e:2 * e]
    v = bag @ w[2 * e:].t() + b[2 * e:]
    s = (q / math.sqrt(e)) @ k.t()
    a = torch.softmax(s, dim=-1)
    out = _lin(a @ v, p, prefix + ".out_proj")
    return out, (a if need_weights else None)


# --------------------------------------------------------------------------- H4
def narrow_gated_attention(query, bag, p, prefix="co_attention", keep=None):
    """models/blocks.py:114-206 (heads = 1).

    S = (q/sqrt(E)) k^T * (tanh(q) tanh(k)^T + 1)/2 ; A = softmax(S) ; optional
    attention-weight dropout (keep mask pre-scaled) ; out = (A v) W_o^T + b_o.
    Returns (q_proj (N,E), attn_out (N,E), A (N,M)) -- A is post-dropout, as in
    the reference (:189-190, :206).
    """
    e = query.shape[-1]
    w, b = p[prefix + ".in_proj_weight"], p[prefix + ".in_proj_bias"]
    q = query @ w[:e].t() + b[:e]
    k = bag @ w[e:2 * e].t() + b[e:2 * e]
    v = bag @ w[2 * e:].t() + b[2 * e:]
    s = (q / math.sqrt(e)) @ k.t()
    gate = (torch.tanh(q) @ torch.tanh(k).t() + 1.0) / 2.0
    a = torch.softmax(s * gate, dim=-1)
    if keep is not None:
        a = a * keep
    out = _lin(a @ v, p, prefix + ".out_proj")
    return q, out, a


# --------------------------------------------------------------------------- H5
def contextual_attention_gate(q_in, q_hat, p, prefix="co_attention.CAG"):
    """models/blocks.py:232-253: G = LN(ELU(ELU(fc1 Q)+ELU(fc2 Qh))), E = LN(ELU(ELU(fc3 Qh))),
    C = ELU(fc_c(G*E)).  LayerNorm eps 1e-5, ELU alpha 1."""
    g = _elu(_elu(_lin(q_in, p, prefix + ".fc1.0")) + _elu(_lin(q_hat, p, prefix + ".fc2.0")))
    g = _layer_norm(g, p[prefix + ".G.1.weight"], p[prefix + ".G.1.bias"])
    ee = _elu(_elu(_lin(q_hat, p, prefix + ".fc3.0")))
    ee = _layer_norm(ee, p[prefix + ".E.1.weight"], p[prefix + ".E.1.bias"])
    return _elu(_lin(g * ee, p, prefix + ".fc_c.0"))


def pregating_contextual_attention(query, bag, p, prefix="co_attention", keep=None):
    """models/blocks.py:92-111: attn_out + CAG(query, q_proj), A."""
    q, out, a = narrow_gated_attention(query, bag, p, prefix, keep)
    c = contextual_attention_gate(query, q, p, prefix + ".CAG")
    return out + c, a


# --------------------------------------------------------------------------- H6
def encoder_layer(x, p, prefix, nhead=8):
    """One post-norm nn.TransformerEncoderLayer (torch/nn/modules/transformer.py:661,
    norm_first=False), ReLU FFN, dropout off.  x: (T, d) one slide, or (B, T, d)."""
    d = x.shape[-1]
    hd = d // nhead
    w, b = p[prefix + ".self_attn.in_proj_weight"], p[prefix + ".self_attn.in_proj_bias"]
    qkv = x @ w.t() + b
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]

    def heads(t):                                              # (..., T, d) -> (..., h, T, hd)
        return t.reshape(*t.shape[:-1], nhead, hd).transpose(-2, -3)

    s = heads(q) @ heads(k).transpose(-1, -2) / math.sqrt(hd)
    o = torch.softmax(s, -1) @ heads(v)                         # (..., h, T, hd)
    o = o.transpose(-2, -3).reshape(x.shape)
    x = _layer_norm(x + _lin(o, p, prefix + ".self_attn.out_proj"),
                    p[prefix + ".norm1.weight"], p[prefix + ".norm1.bias"])
    f = _lin(torch.relu(_lin(x, p, prefix + ".linear1")), p, prefix + ".linear2")
    return _layer_norm(x + f, p[prefix + ".norm2.weight"], p[prefix + ".norm2.bias"])


def set_transformer(x, p, prefix, num_layers=2, nhead=8):
    """nn.TransformerEncoder(num_layers=2), no final norm; models/mcat/mcat.py:51-53,101-102."""
    for i in range(num_layers):
        x = encoder_layer(x, p, f"{prefix}.layers.{i}", nhead)
    return x


# --------------------------------------------------------------------------- H7
def gated_attention_scores(x, p, prefix, keep_a=None, keep_b=None):
    """AttentionNetGated, models/blocks.py:13-48: A = W_c[tanh(W_a x) * sigmoid(W_b x)] (L,1)."""
    a = torch.tanh(_lin(x, p, prefix + ".attention_a.0"))
    b = torch.sigmoid(_lin(x, p, prefix + ".attention_b.0"))
    if keep_a is not None:
        a = a * keep_a
    if keep_b is not None:
        b = b * keep_b
    return _lin(a * b, p, prefix + ".attention_c")


def gated_mil_pool(x, p, head_prefix, rho_prefix):
    """Pooling idiom models/mcat/mcat.py:105-109: A^T, softmax over L, mm, rho (Linear+ReLU).
    Returns (A (1,L) raw scores, h (d,))."""
    a = gated_attention_scores(x, p, head_prefix).t()           # (1, L)
    h = torch.softmax(a, dim=1) @ x                             # (1, d)
    h = torch.relu(_lin(h, p, rho_prefix + ".0")).squeeze()
    return a, h


# --------------------------------------------------------------------------- H8
def concat_fusion(h_path, h_omic, p, prefix="fusion_layer"):
    """ConcatFusion, models/fusion.py:7-19."""
    x = torch.cat([h_path, h_omic], dim=0)
    x = torch.relu(_lin(x, p, prefix + ".fusion_layer.0"))
    return torch.relu(_lin(x, p, prefix + ".fusion_layer.2"))


def gated_concat_fusion(h_path, h_omic, p, prefix="fusion_layer"):
    """GatedConcatFusion, models/fusion.py:22-41: x_i * sigmoid(Linear_i(x_i)), cat, the ConcatFusion MLP.  The
    reference's gates live in an unregistered list (:25-27); here they are parameters `<prefix>.gates.<i>.0.*`."""
    items = []
    for i, x in enumerate((h_path, h_omic)):
        items.append(x * torch.sigmoid(_lin(x, p, f"{prefix}.gates.{i}.0")))
    x = torch.cat(items, dim=0)
    x = torch.relu(_lin(x, p, prefix + ".fusion_layer.0"))
    return torch.relu(_lin(x, p, prefix + ".fusion_layer.2"))


def bilinear_fusion(x1, x2, p, prefix="fusion_layer"):
    """BilinearFusion, models/fusion.py:44-113 with its defaults (gates, bilinear, skip connection), eval mode."""
    def bil(a, b, name):                      # nn.Bilinear: z_k = a^T W_k b + bias_k
        return torch.einsum("i,kij,j->k", a, p[f"{prefix}.{name}.weight"], b) + p[f"{prefix}.{name}.bias"]
    h1 = torch.relu(_lin(x1, p, prefix + ".linear_h1.0"))
    o1 = torch.relu(_lin(torch.sigmoid(bil(x1, x2, "linear_z1")) * h1, p, prefix + ".linear_o1.0"))       # :88-91
    h2 = torch.relu(_lin(x2, p, prefix + ".linear_h2.0"))
    o2 = torch.relu(_lin(torch.sigmoid(bil(x2, x1, "linear_z2")) * h2, p, prefix + ".linear_o2.0"))       # :95-98
    o1 = torch.cat([o1, torch.ones(1)])                                                                   # :103-106
    o2 = torch.cat([o2, torch.ones(1)])
    out = torch.outer(o1, o2).flatten()                                                                   # :107
    out = torch.relu(_lin(out, p, prefix + ".fc1.0"))                                                     # :111
    out = torch.cat([out, o1, o2])                                                                        # :112-113
    return torch.relu(_lin(out, p, prefix + ".fc2.0"))                                                    # :114


def survival_head(h, p, prefix="classifier"):
    """models/mcat/mcat.py:126-138: logits (1,C) -> hazards, survs = cumprod(1-hazards), Y = softmax."""
    logits = _lin(h, p, prefix).unsqueeze(0)
    hazards = torch.sigmoid(logits)
    survs = torch.cumprod(1 - hazards, dim=1)
    return hazards, survs, torch.softmax(logits, dim=1)


# --------------------------------------------------------------------------- H1
def _tail(h_coattn, g_bag, a_coattn, p, fusion="concat"):
    path = set_transformer(h_coattn, p, "path_transformer")
    omic = set_transformer(g_bag, p, "omic_transformer")
    a_path, h_path = gated_mil_pool(path, p, "path_attention_head", "path_rho")
    a_omic, h_omic = gated_mil_pool(omic, p, "omic_attention_head", "omic_rho")
    h = {"concat": concat_fusion, "gated_concat": gated_concat_fusion, "bilinear": bilinear_fusion}[fusion](h_path, h_omic, p)
    hazards, survs, y = survival_head(h, p)
    return hazards, survs, y, {"coattn": a_coattn, "path": a_path, "omic": a_omic}


def mcat_forward(p, wsi, omics, inference=False, bag_storage=None, fusion="concat", round_gemm_out=False,
                 storage_points=("x", "w", "h")):
    """MultimodalCoAttentionTransformer.forward, models/mcat/mcat.py:84-142 (eval mode).
    bag_storage / round_gemm_out: see patch_fc (the fused kernel of the 'medium' model rounds H_bag once)."""
    h_bag = patch_fc(wsi, p, storage=bag_storage, round_gemm_out=round_gemm_out, storage_points=storage_points)
    g_bag = omic_fc(omics, p)
    h_co, a_co = mcat_coattention(g_bag, h_bag, p, need_weights=inference)
    return _tail(h_co, g_bag, a_co, p, fusion)


def nacagat_forward(p, wsi, omics, bag_storage=None, round_gemm_out=False, storage_points=("x", "w", "h")):
    """NarrowContextualAttentionGateTransformer.forward, models/nacagat/nacagat.py:80-138 (eval mode).
    bag_storage / round_gemm_out: see patch_fc (the 'medium' model's patch-layer kernel rounds H_bag once)."""
    h_bag = patch_fc(wsi, p, storage=bag_storage, round_gemm_out=round_gemm_out, storage_points=storage_points)
    g_bag = omic_fc(omics, p)
    h_co, a_co = pregating_contextual_attention(g_bag, h_bag, p)
    return _tail(h_co, g_bag, a_co, p)


# --------------------------------------------------------------------------- f3
def bag_self_attention(x, p, prefix="self_attention", nhead=1):
    """nn.MultiheadAttention(embed, num_heads)(x, x, x) on the unbatched (M, d) bag, dropout off
    (models/ge_nacagat/ge_nacagat.py:27,49).  Returns (output (M, d), map averaged over heads (M, M))."""
    m, d = x.shape
    hd = d // nhead
    qkv = x @ p[prefix + ".in_proj_weight"].t() + p[prefix + ".in_proj_bias"]
    q, k, v = (qkv[:, i * d:(i + 1) * d].reshape(m, nhead, hd).transpose(0, 1) for i in range(3))   # (h, M, hd)
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(hd), dim=-1)                                 # (h, M, M)
    o = (a @ v).transpose(0, 1).reshape(m, d)
    return _lin(o, p, prefix + ".out_proj"), a.mean(0)


def ge_nacagat_forward(p, wsi, bag_storage=None, round_gemm_out=False):
    """GeneExprNarrowContextualAttentionGateTransformer.forward, models/ge_nacagat/ge_nacagat.py:43-72 (eval mode):
    Y (n_classes,) = softmax over the classifier logits, {'attn': (M, M), 'path': (1, M) raw pooling scores}."""
    h_bag = patch_fc(wsi, p, storage=bag_storage, round_gemm_out=round_gemm_out)
    h_co, a_co = bag_self_attention(h_bag, p)
    path = set_transformer(h_co, p, "path_transformer")
    a_path, h_path = gated_mil_pool(path, p, "path_attention_head", "path_rho")
    logits = _lin(h_path, p, "classifier")
    return torch.softmax(logits, dim=0), {"attn": a_co, "path": a_path}


def ge_ce_loss(y, target):
    """models/ge_nacagat/main.py:33 -- nn.CrossEntropyLoss applied to the ALREADY soft-maxed Y (the reference's quirk)."""
    return F.cross_entropy(y.unsqueeze(0), target.reshape(1).long())


# --------------------------------------------------------------------------- H9
def ces_loss(hazards, survs, y, c, alpha=0.75, eps=1e-7):
    """CrossEntropySurvivalLoss, models/loss.py:5-28 (batch of one slide)."""
    y = y.view(-1, 1)
    c = c.view(-1, 1).float()
    s_pad = torch.cat([torch.ones_like(c), survs], 1)
    reg = -(1 - c) * (torch.log(torch.gather(s_pad, 1, y).clamp(min=eps))
                      + torch.log(torch.gather(hazards, 1, y).clamp(min=eps)))
    s_y = torch.gather(survs, 1, y).clamp(min=eps)
    ce = -(c * torch.log(s_y) + (1 - c) * torch.log(1 - s_y))
    return ((1 - alpha) * ce + alpha * reg).mean()


def cesar_loss(hazards, survs, y, c, attention, alpha=0.75, eps=1e-7, lambda_reg=0.01):
    """CrossEntropySurvivalAttnRegLoss, models/loss.py:88-101: ces + lambda * ||A||_2."""
    attn = lambda_reg * torch.norm(attention, p=2)
    return ces_loss(hazards, survs, y, c, alpha, eps) + attn, attn


def risk_score(survs):
    """risk = -sum_j survs_j; models/mcat/main.py:56."""
    return -survs.sum(dim=1)


def concordance_index_censored(event, time, risk, tied_tol=1e-8):
    """Harrell's C as scikit-survival 0.2x defines it (the reference calls
    sksurv.metrics.concordance_index_censored, models/mcat/main.py:12,81).

    PARITY UNPINNED against the library (not installed here).  Published algorithm:
    a pair (i, j) is comparable when i had an event and time_i < time_j, or
    time_i == time_j with j censored; it is concordant when risk_i > risk_j,
    |risk_i - risk_j| <= tied_tol counts one half.  Returns the index (float).
    """
    import numpy as np
    event = np.asarray(event, dtype=bool)
    time = np.asarray(time, dtype=np.float64)
    risk = np.asarray(risk, dtype=np.float64)
    num = 0.0
    den = 0.0
    for i in range(len(time)):
        if not event[i]:
            continue
        comparable = (time > time[i]) | ((time == time[i]) & ~event)
        comparable[i] = False
        if not comparable.any():
            continue
        diff = risk[i] - risk[comparable]
        ties = np.abs(diff) <= tied_tol
        num += float(((diff > 0) & ~ties).sum()) + 0.5 * float(ties.sum())
        den += float(comparable.sum())
    if den == 0:
        raise ValueError("no comparable pairs")
    return num / den
