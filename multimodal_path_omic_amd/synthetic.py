"""Seeded synthetic slides, omics and weights (numpy PCG64 only).

Everything the parity tests, the golden-vector generator and bench.py feed to
either implementation comes from here, so that a fixture needs to store only a
seed and the expected outputs.  Nothing in this
module touches the GPU or the HIP library.

Shapes follow the reference's input contract (dataset/dataset.py:119-143 via
DataLoader(batch_size=1), models/mcat/mcat.py:151-152): a slide is a bag of M
patch embeddings of width 1024, omics are N vectors of per-group width.
"""
from __future__ import annotations

import numpy as np
import torch

PATCH_DIM = 1024            # models/mcat/mcat.py:25
MODEL_SIZES = {"small": 128, "medium": 256, "big": 512}   # models/mcat/mcat.py:16-21
REF_TEST_OMIC_SIZES = [100, 200, 300, 400, 500, 600]      # models/mcat/mcat.py:152


def rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def normal(gen: np.random.Generator, shape, scale: float = 1.0) -> torch.Tensor:
    return torch.from_numpy((gen.standard_normal(shape, dtype=np.float64) * scale).astype(np.float32))


def make_bag(m: int, seed: int, dim: int = PATCH_DIM) -> torch.Tensor:
    """(M, dim) fp32 patch-embedding bag, N(0,1)."""
    return normal(rng(seed), (m, dim))


def make_omics(sizes, seed: int):
    gen = rng(seed)
    return [normal(gen, (s,)) for s in sizes]


def fill_state_dict(shapes: "dict[str, tuple]", seed: int, gain: float = 1.0) -> "dict[str, torch.Tensor]":
    """Deterministic weights for a {name: shape} listing, in listing order.

    Matrices ~ N(0, gain/sqrt(fan_in)); LayerNorm weights (names ending in
    'norm1.weight', 'norm2.weight', or '.1.weight' under CAG.G / CAG.E) ~ 1 + N(0, 0.1);
    every other 1-D tensor ~ N(0, 0.1).  Non-trivial biases and LayerNorm affine
    terms are deliberate: default init (zero biases) hides bias bugs.
    """
    gen = rng(seed)
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            out[name] = normal(gen, shape, gain / np.sqrt(fan_in))
        elif _is_layernorm_weight(name):
            out[name] = 1.0 + normal(gen, shape, 0.1)
        else:
            out[name] = normal(gen, shape, 0.1)
    return out


def _is_layernorm_weight(name: str) -> bool:
    if name.endswith("norm1.weight") or name.endswith("norm2.weight"):
        return True
    return name.endswith("G.1.weight") or name.endswith("E.1.weight")


def slide_lengths(n_slides: int, lo: int, hi: int, seed: int) -> "list[int]":
    """Fixed multiset of bag lengths (cfg 4: 2k-30k patches)."""
    gen = rng(seed)
    return [int(x) for x in gen.integers(lo, hi + 1, size=n_slides)]


def make_cohort(n_slides: int, m_lo: int, m_hi: int, omic_sizes, seed: int, n_classes: int = 4):
    """Seeded synthetic survival cohort with a planted signal.

    Returns a list of dicts with keys wsi (M,1024), omics [N x (d_i,)],
    survival_months, survival_class, censorship -- the tuple layout of
    MultimodalDataset.__getitem__ (dataset/dataset.py:119-143).  The event time is
    a noisy monotone function of one direction in patch space and one in omic
    space so that a model can learn a C-index above 0.5.
    """
    gen = rng(seed)
    w_patch = gen.standard_normal(PATCH_DIM) / np.sqrt(PATCH_DIM)
    slides = []
    for i in range(n_slides):
        m = int(gen.integers(m_lo, m_hi + 1))
        z = gen.standard_normal()                        # latent risk
        wsi = gen.standard_normal((m, PATCH_DIM)) + 0.5 * z * w_patch[None, :] * np.sqrt(PATCH_DIM) / 8
        omics = [gen.standard_normal(s) + 0.5 * z for s in omic_sizes]
        months = float(np.exp(3.0 - 0.8 * z + 0.3 * gen.standard_normal()))
        slides.append(dict(
            wsi=torch.from_numpy(wsi.astype(np.float32)),
            omics=[torch.from_numpy(o.astype(np.float32)) for o in omics],
            survival_months=months,
            censorship=int(gen.random() < 0.3),
        ))
    # discretise months into n_classes quantile bins (dataset/dataset.py labels are 0..3)
    months = np.array([s["survival_months"] for s in slides])
    edges = np.quantile(months, np.linspace(0, 1, n_classes + 1)[1:-1])
    for s in slides:
        s["survival_class"] = int(np.searchsorted(edges, s["survival_months"]))
    return slides


def subsample(t: torch.Tensor, n: int = 4096) -> torch.Tensor:
    """Deterministic strided sample of a tensor's elements (fixtures store these, not full grads)."""
    flat = t.detach().reshape(-1)
    if flat.numel() <= n:
        return flat.clone()
    stride = -(-flat.numel() // n)
    return flat[::stride].clone()
