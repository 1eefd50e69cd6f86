"""Case tables and seeded input builders shared by make_golden.py (which runs the
REFERENCE on them) and the tests (which run the oracle and the HIP path on them).

A fixture stores only expected outputs; inputs and weights are rebuilt from the
seed with multimodal-path-omic_amd/synthetic.py, so generator and tests cannot drift.
"""
from __future__ import annotations

import torch

from multimodal_path_omic_amd import synthetic as syn

E = 256          # 'medium' model width, models/mcat/mcat.py:18-19
N_OMIC = 6

# name -> (M, weight gain, seed).  gain > 1 scales every matrix: peaky attention.
COATTN_CASES = {
    "m256": (256, 1.0, 101),
    "m2000": (2000, 1.0, 102),
    "m2000_peaky": (2000, 8.0, 103),
    "m777_ragged": (777, 2.0, 105),       # M not a multiple of any tile
    "m15000": (15000, 1.0, 104),
}
NACAGAT_CASES = {
    "m256": (256, 1.0, 201),
    "m2000": (2000, 1.0, 202),
    "m2000_peaky": (2000, 3.0, 203),
    "m777_ragged": (777, 1.5, 205),
    "m15000": (15000, 1.0, 204),
}
POOL_CASES = {"l6": (6, 301), "l3000": (3000, 302)}

MCAT_COATTN_SHAPES = {
    "co_attention.in_proj_weight": (3 * E, E), "co_attention.in_proj_bias": (3 * E,),
    "co_attention.out_proj.weight": (E, E), "co_attention.out_proj.bias": (E,),
}
CAG_SHAPES = {
    f"co_attention.CAG.{n}": s for n, s in {
        "fc1.0.weight": (E, E), "fc1.0.bias": (E,), "fc2.0.weight": (E, E), "fc2.0.bias": (E,),
        "fc3.0.weight": (E, E), "fc3.0.bias": (E,),
        "G.1.weight": (E,), "G.1.bias": (E,), "E.1.weight": (E,), "E.1.bias": (E,),
        "fc_c.0.weight": (E, E), "fc_c.0.bias": (E,)}.items()
}
NACAGAT_COATTN_SHAPES = {**MCAT_COATTN_SHAPES, **CAG_SHAPES}


def encoder_shapes(prefix, layers=2, d=E, ff=512):
    out = {}
    for i in range(layers):
        p = f"{prefix}.layers.{i}"
        out.update({
            f"{p}.self_attn.in_proj_weight": (3 * d, d), f"{p}.self_attn.in_proj_bias": (3 * d,),
            f"{p}.self_attn.out_proj.weight": (d, d), f"{p}.self_attn.out_proj.bias": (d,),
            f"{p}.linear1.weight": (ff, d), f"{p}.linear1.bias": (ff,),
            f"{p}.linear2.weight": (d, ff), f"{p}.linear2.bias": (d,),
            f"{p}.norm1.weight": (d,), f"{p}.norm1.bias": (d,),
            f"{p}.norm2.weight": (d,), f"{p}.norm2.bias": (d,)})
    return out


def pool_shapes(head, rho, d=E):
    return {
        f"{head}.attention_a.0.weight": (d, d), f"{head}.attention_a.0.bias": (d,),
        f"{head}.attention_b.0.weight": (d, d), f"{head}.attention_b.0.bias": (d,),
        f"{head}.attention_c.weight": (1, d), f"{head}.attention_c.bias": (1,),
        f"{rho}.0.weight": (d, d), f"{rho}.0.bias": (d,)}


FUSION_SHAPES = {
    "fusion_layer.fusion_layer.0.weight": (E, 2 * E), "fusion_layer.fusion_layer.0.bias": (E,),
    "fusion_layer.fusion_layer.2.weight": (E, E), "fusion_layer.fusion_layer.2.bias": (E,),
    "classifier.weight": (4, E), "classifier.bias": (4,)}


# row f4 fusion layers (models/fusion.py:22-113) with the constructor arguments of models/mcat/mcat.py:73-77
BILINEAR_SHAPES = {
    "linear_h1.0.weight": (32, E), "linear_h1.0.bias": (32,),
    "linear_z1.weight": (32, E, E), "linear_z1.bias": (32,),
    "linear_o1.0.weight": (32, 32), "linear_o1.0.bias": (32,),
    "linear_h2.0.weight": (32, E), "linear_h2.0.bias": (32,),
    "linear_z2.weight": (32, E, E), "linear_z2.bias": (32,),
    "linear_o2.0.weight": (32, 32), "linear_o2.0.bias": (32,),
    "fc1.0.weight": (64, 33 * 33), "fc1.0.bias": (64,),
    "fc2.0.weight": (E, 64 + 64 + 2), "fc2.0.bias": (E,)}
GATED_CONCAT_SHAPES = {
    "gates.0.0.weight": (1, E), "gates.0.0.bias": (1,), "gates.1.0.weight": (1, E), "gates.1.0.bias": (1,),
    "fusion_layer.0.weight": (E, 2 * E), "fusion_layer.0.bias": (E,),
    "fusion_layer.2.weight": (E, E), "fusion_layer.2.bias": (E,)}


def model_shapes(omic_sizes, nacagat: bool, d=E):
    """Full state_dict listing in the reference's registration order
    (models/mcat/mcat.py:24-82, models/nacagat/nacagat.py:20-78)."""
    s = {"H.0.weight": (d, 1024), "H.0.bias": (d,)}
    for i, w in enumerate(omic_sizes):
        s.update({f"G.{i}.0.0.weight": (d, w), f"G.{i}.0.0.bias": (d,),
                  f"G.{i}.1.0.weight": (d, d), f"G.{i}.1.0.bias": (d,)})
    s.update(NACAGAT_COATTN_SHAPES if nacagat else MCAT_COATTN_SHAPES)
    s.update(encoder_shapes("path_transformer"))
    s.update(pool_shapes("path_attention_head", "path_rho"))
    s.update(encoder_shapes("omic_transformer"))
    s.update(pool_shapes("omic_attention_head", "omic_rho"))
    s.update(FUSION_SHAPES)
    return s


# name -> (model, M, omic sizes, seed)
MODEL_CASES = {
    "mcat_cfg1": ("mcat", 256, syn.REF_TEST_OMIC_SIZES, 401),
    "mcat_m2000": ("mcat", 2000, [256] * 6, 402),
    "mcat_m15000": ("mcat", 15000, [256] * 6, 403),
    "nacagat_cfg1": ("nacagat", 256, syn.REF_TEST_OMIC_SIZES, 411),
    "nacagat_m2000": ("nacagat", 2000, [256] * 6, 412),
    "nacagat_m15000": ("nacagat", 15000, [256] * 6, 413),
}


def ge_model_shapes(d=E, n_classes=3):
    """state_dict of the gene-expression model in registration order (models/ge_nacagat/ge_nacagat.py:19-41)."""
    s = {"H.0.weight": (d, 1024), "H.0.bias": (d,),
         "self_attention.in_proj_weight": (3 * d, d), "self_attention.in_proj_bias": (3 * d,),
         "self_attention.out_proj.weight": (d, d), "self_attention.out_proj.bias": (d,)}
    s.update(encoder_shapes("path_transformer", d=d))
    s.update(pool_shapes("path_attention_head", "path_rho", d=d))
    s.update({"classifier.weight": (n_classes, d), "classifier.bias": (n_classes,)})
    return s


# row f3: name -> (M, seed).  3000 is the size of the reference's own smoke test (ge_nacagat.py:82); 333 is ragged against
# every tile size of the kernels.
GE_MODEL_CASES = {"ge_m333": (333, 431), "ge_m3000": (3000, 432)}


def ge_model_inputs(m, seed):
    return syn.make_bag(m, seed), torch.tensor([seed % 3])


# ------------------------------------------------------------------ input builders
def coattn_inputs(m, seed):
    """query (6,E) and an H_bag-like bag (M,E): ReLU'd normal, ~50 % zeros."""
    g = syn.rng(seed)
    query = syn.normal(g, (N_OMIC, E))
    bag = torch.clamp(syn.normal(g, (m, E)), min=0.0)
    probe_out = syn.normal(g, (N_OMIC, E))
    probe_a = syn.normal(g, (N_OMIC, m))
    return query, bag, probe_out, probe_a


def cag_inputs(seed=501):
    g = syn.rng(seed)
    return syn.normal(g, (N_OMIC, E)), syn.normal(g, (N_OMIC, E)), syn.normal(g, (N_OMIC, E))


def encoder_inputs(seed=601):
    g = syn.rng(seed)
    return syn.normal(g, (N_OMIC, E)), syn.normal(g, (N_OMIC, E))


def pool_inputs(l, seed):
    g = syn.rng(seed)
    return syn.normal(g, (l, E)), syn.normal(g, (E,)), syn.normal(g, (1, l))


def fusion_inputs(seed=701):
    g = syn.rng(seed)
    return syn.normal(g, (E,)), syn.normal(g, (E,)), syn.normal(g, (1, 4))


def model_inputs(m, omic_sizes, seed):
    wsi = syn.make_bag(m, seed)
    omics = syn.make_omics(omic_sizes, seed + 1000)
    label = torch.tensor([seed % 4])
    censor = torch.tensor([float(seed % 2)])
    return wsi, omics, label, censor


# cohort run: slides, M range, epochs, grad_acc_step, seed
# (SURVEY 8(c) planned 64 slides, M <= 2 048, 3 epochs, 80/20: 80 slides make the split integral -- 64 train slides = 8 whole
#  accumulation windows of 8, 16 validation slides)
COHORT = dict(n_slides=80, m_lo=256, m_hi=2048, epochs=3, grad_acc_step=8, seed=901,
              weight_seed=902, omic_sizes=[64] * 6, lr=2e-4, weight_decay=1e-5, train_frac=0.8)
