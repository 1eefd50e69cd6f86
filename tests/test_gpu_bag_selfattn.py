"""Row f3 on the GPU: self-attention over the M rows of a bag (csrc/bag_selfattn.hip), the set-Transformer and the
gated pooling head called with T = L = M, and the gene-expression model of models/ge_nacagat/ge_nacagat.py against the
reference's golden vectors.  Kernel-level checks compare with a plain torch fp32 restatement of the same op; bars: 1e-3
relative on maps (north star), 1e-3 of the largest entry on gradients."""
import math

import pytest
import torch

import cases as C
from multimodal_path_omic_amd import ops
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.models import GeneExprNarrowContextualAttentionGateTransformer
from multimodal_path_omic_amd.transformer import make_set_transformer
from multimodal_path_omic_amd.blocks import AttentionNetGated
from oracle import mpo_oracle as O

pytestmark = pytest.mark.gpu
sub = syn.subsample


def relmax(a, b, scale=None):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max().clamp_min(1e-30) if scale is None else scale))


def part_scale(ref):
    """Scale for one of dq / dk / dv: its own largest entry, but no smaller than 1e-2 of the whole gradient's (at M = 1 the
    true dq and dk are exactly zero)."""
    return lambda sl: max(float(ref[..., sl].abs().max()), 1e-2 * float(ref.abs().max()))


def attention_ref(qkv, heads, keep=None):
    """(n, M, 3d) fp64 on the CPU -> out (n, M, d), probabilities (n, h, M, M); keep: (n, h, M, M) scaled keep mask."""
    n, m, d3 = qkv.shape
    d, hd = d3 // 3, d3 // 3 // heads
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(n, m, heads, hd).transpose(1, 2) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1)
    pd = p if keep is None else p * keep
    return (pd @ v).transpose(1, 2).reshape(n, m, d), p


CASES = [  # n_bags, M, d, heads, q gain
    (1, 333, 256, 1, 1.0), (1, 1000, 256, 8, 1.0), (2, 64, 256, 8, 1.0), (1, 70, 128, 1, 1.0), (1, 130, 128, 8, 1.0),
    (1, 200, 512, 8, 1.0), (1, 257, 256, 1, 6.0), (1, 515, 256, 8, 6.0), (1, 1, 256, 8, 1.0), (1, 17, 256, 1, 1.0),
    (1, 150, 512, 1, 1.0), (2, 333, 512, 1, 6.0)]      # one head of 512 (model_size='big'): dK / dV in two column passes


def _b3_modes(d, heads):
    """Heads of width 32 (several) and 256 (one) run on three-term bf16 MFMAs by default (~16 mantissa bits per operand); the
    verification hook keeps them on the fp32 kernels.  -> [(hook value, output bar, gradient bar)]"""
    b3 = (heads > 1 and d == 32 * heads) or (heads == 1 and d == 256)
    return [(1, 1e-4, 1e-3), (0, 1e-5, 1e-4)] if b3 else [(1, 1e-5, 1e-4)]


class bf16x3:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        from multimodal_path_omic_amd import _lib as L
        self.was = L.lib().mpo_set_bag_self_attention_bf16x3(self.on)

    def __exit__(self, *exc):
        from multimodal_path_omic_amd import _lib as L
        L.lib().mpo_set_bag_self_attention_bf16x3(self.was)


@pytest.mark.parametrize("n,m,d,heads,gain", CASES)
def test_attention_core_equals_torch(dev, n, m, d, heads, gain):
    g = syn.rng(7000 + m + heads)
    qkv = syn.normal(g, (n, m, 3 * d))
    qkv[..., :d] *= gain                                       # gain 6: peaky rows (a few keys take all the mass)
    probe = syn.normal(g, (n, m, d))
    xr = qkv.double().requires_grad_(True)
    out_r, p_r = attention_ref(xr, heads)
    (out_r * probe.double()).sum().backward()
    for hook, out_bar, grad_bar in _b3_modes(d, heads):
        with bf16x3(hook):
            x = qkv.to(dev).requires_grad_(True)
            out, amap = ops.BagSelfAttentionFn.apply(x, heads, 0.0, heads == 1)
            (out * probe.to(dev)).sum().backward()
        assert relmax(out, out_r) < out_bar, (hook, relmax(out, out_r))
        assert relmax(x.grad, xr.grad) < grad_bar, (hook, relmax(x.grad, xr.grad))
        for part in range(3):                                  # dq, dk, dv separately: none hides behind a larger one
            sl = slice(part * d, (part + 1) * d)
            assert relmax(x.grad[..., sl], xr.grad[..., sl], part_scale(xr.grad)(sl)) < grad_bar, (hook, part)
        if heads == 1:
            assert amap.shape == (n, m, m)
            ref = p_r[:, 0].detach()
            big = ref > 1e-30
            err = ((amap.double().cpu() - ref).abs() / ref.clamp_min(1e-30))[big].max().item()
            print(f"[self-attention map] M={m} d={d} gain={gain} mode={hook}: max relative error {err:.2e}")
            assert err < 1e-3, (hook, err)                     # north-star bar on maps: elementwise relative, both arithmetics
            assert float((amap.sum(-1) - 1).abs().max()) < 1e-4
        else:
            assert amap is None


def _recover_keep(dev, qkv, heads, p, offset):
    """The kernel's own (scaled) keep mask, read back through V = identity blocks: out[q][h hd + c] = P_drop[h][q][b hd + c]."""
    n, m, d3 = qkv.shape
    d, hd = d3 // 3, d3 // 3 // heads
    _, p_ref = attention_ref(qkv.double(), heads)
    keep = torch.zeros(n, heads, m, m, dtype=torch.float64)
    for b in range((m + hd - 1) // hd):
        probe = qkv.clone()
        v = torch.zeros(n, m, heads, hd)
        rows = torch.arange(b * hd, min(m, (b + 1) * hd))
        v[:, rows, :, rows - b * hd] = 1.0
        probe[..., 2 * d:] = v.reshape(n, m, d)
        ops._rng_calls = offset
        out, _ = ops.BagSelfAttentionFn.apply(probe.to(dev), heads, p, False)
        pd = out.cpu().double().reshape(n, m, heads, hd).permute(0, 2, 1, 3)          # (n, h, q, c)
        keep[..., rows] = pd[..., : len(rows)] / p_ref[..., rows].clamp_min(1e-300)
    return keep


@pytest.mark.parametrize("m,d,heads", [(96, 256, 8), (200, 256, 1), (70, 512, 1)])
def test_attention_dropout_mask_is_shared_by_forward_and_backward(dev, m, d, heads):
    """Dropout on the probabilities: the mask is never stored.  It is read back here through identity-block values, must be
    {0, 1/(1-p)} at the stated rate, and a torch restatement using exactly that mask must reproduce the forward and all
    three gradients -- i.e. forward, dQ and dK/dV kernels regenerate the same mask."""
    for hook, out_bar, grad_bar in _b3_modes(d, heads):
        with bf16x3(hook):
            _dropout_mask_check(dev, m, d, heads, out_bar, grad_bar)


def _dropout_mask_check(dev, m, d, heads, out_bar, grad_bar):
    g = syn.rng(7100 + m)
    qkv = syn.normal(g, (1, m, 3 * d)) * 0.5
    probe = syn.normal(g, (1, m, d))
    p, offset = 0.25, 12345
    keep = _recover_keep(dev, qkv, heads, p, offset)
    vals = keep.round(decimals=4).unique()
    assert all(min(abs(float(v)), abs(float(v) - 1 / (1 - p))) < 1e-3 for v in vals), vals
    rate = float((keep < 0.5).double().mean())
    assert abs(rate - p) < 4 * math.sqrt(p * (1 - p) / keep.numel()) + 1e-3, rate
    if heads > 1:
        assert not torch.equal(keep[0, 0] > 0.5, keep[0, 1] > 0.5)                    # heads draw their own masks
    ops._rng_calls = offset
    x = qkv.to(dev).requires_grad_(True)
    out, _ = ops.BagSelfAttentionFn.apply(x, heads, p, False)
    (out * probe.to(dev)).sum().backward()
    xr = qkv.double().requires_grad_(True)
    out_r, _ = attention_ref(xr, heads, keep=(keep > 0.5).double() / (1 - p))
    (out_r * probe.double()).sum().backward()
    assert relmax(out, out_r) < out_bar
    for part in range(3):
        sl = slice(part * d, (part + 1) * d)
        assert relmax(x.grad[..., sl], xr.grad[..., sl], part_scale(xr.grad)(sl)) < grad_bar, part
    # another offset = another mask
    ops._rng_calls = offset + 1
    out2, _ = ops.BagSelfAttentionFn.apply(qkv.to(dev), heads, p, False)
    assert not torch.equal(out2, out.detach())


def _leaf(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


@pytest.mark.parametrize("t", [17, 200, 1000])
def test_set_transformer_over_bag_rows_equals_oracle(dev, t):
    """nn.TransformerEncoder over T = M rows (ge_nacagat.py:30-33,53): the 6-token tail's launch sequence with the
    long-axis attention kernels, against the oracle's encoder and its autograd gradients."""
    shapes = C.encoder_shapes("enc")
    sd = syn.fill_state_dict(shapes, 7200 + t)
    enc = make_set_transformer(256, 0.25)
    enc.load_state_dict({k[len("enc."):]: v for k, v in sd.items()}, strict=True)
    enc = enc.to(dev).eval()
    g = syn.rng(7300 + t)
    x, probe = syn.normal(g, (t, 256)), syn.normal(g, (t, 256))
    xd = x.to(dev).requires_grad_(True)
    y = enc(xd)
    (y * probe.to(dev)).sum().backward()
    p = _leaf(sd)
    xo = x.clone().requires_grad_(True)
    yo = O.set_transformer(xo, p, "enc")
    (yo * probe).sum().backward()
    assert relmax(y, yo) < 1e-4
    assert relmax(xd.grad, xo.grad) < 1e-3
    for n, prm in enc.named_parameters():
        ref = p["enc." + n].grad
        assert relmax(prm.grad, ref) < 2e-3, (n, relmax(prm.grad, ref))


@pytest.mark.parametrize("b,l", [(1, 300), (2, 100), (1, 65), (1, 4099)])
def test_gated_pooling_over_bag_rows_equals_oracle(dev, b, l):
    """AttentionNetGated + softmax pooling + rho over L = M rows (ge_nacagat.py:56-60) against the oracle."""
    shapes = C.pool_shapes("head", "rho")
    sd = syn.fill_state_dict(shapes, 7400 + l)
    head = AttentionNetGated(n_classes=1, input_dim=256, hidden_dim=256)
    rho = torch.nn.Sequential(torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Dropout(0.25))
    head.load_state_dict({k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}, strict=True)
    rho.load_state_dict({k[len("rho."):]: v for k, v in sd.items() if k.startswith("rho.")}, strict=True)
    head, rho = head.to(dev).eval(), rho.to(dev).eval()
    g = syn.rng(7500 + l)
    x, probe_h, probe_a = syn.normal(g, (b, l, 256)), syn.normal(g, (b, 256)), syn.normal(g, (b, 1, l))
    xd = x.to(dev).requires_grad_(True)
    a, h = ops.gated_pool(xd, head, rho, False)
    ((h * probe_h.to(dev)).sum() + (a * probe_a.to(dev)).sum()).backward()
    p = _leaf(sd)
    xo = x.clone().requires_grad_(True)
    outs = [O.gated_mil_pool(xo[i], p, "head", "rho") for i in range(b)]
    ao, ho = torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs])
    ((ho * probe_h).sum() + (ao * probe_a).sum()).backward()
    assert relmax(a, ao) < 1e-4 and relmax(h, ho) < 1e-4
    assert relmax(xd.grad, xo.grad) < 1e-3
    for mod, pre in ((head, "head."), (rho, "rho.")):
        for n, prm in mod.named_parameters():
            assert relmax(prm.grad, p[pre + n].grad) < 2e-3, (pre + n)


def build_ge(dev, seed, bag_dtype=torch.float32):
    model = GeneExprNarrowContextualAttentionGateTransformer(model_size="medium", bag_dtype=bag_dtype)
    sd = syn.fill_state_dict(C.ge_model_shapes(), seed)
    model.load_state_dict(sd, strict=True)                     # the reference's state_dict layout loads as is
    return model.to(dev).eval(), sd


@pytest.mark.parametrize("case", list(C.GE_MODEL_CASES))
def test_ge_model_matches_reference_golden(dev, golden, case):
    g = golden("ge_models")
    m, seed = C.GE_MODEL_CASES[case]
    model, sd = build_ge(dev, seed)
    wsi, target = C.ge_model_inputs(m, seed + 1)
    y, att = model(wsi=wsi.to(dev))
    assert y.shape == (3,) and att["attn"].shape == (m, m) and att["path"].shape == (1, m)
    y_b, _ = model(wsi=wsi.to(dev).unsqueeze(0))               # DataLoader convention
    assert torch.equal(y, y_b)
    assert float((y.detach().cpu() - g[f"{case}/Y"]).abs().max()) < 1e-4
    assert relmax(att["path"], g[f"{case}/A_path"]) < 1e-3
    for got, key in ((sub(att["attn"]), "A_attn_sub"), (att["attn"].max(dim=1).values, "A_attn_rowmax"),
                     (att["attn"].diagonal(), "A_attn_diag")):
        ref = g[f"{case}/{key}"]
        assert ((got.cpu() - ref).abs() / ref.clamp_min(1e-30)).max().item() < 1e-3, key
    loss = torch.nn.functional.cross_entropy(y.unsqueeze(0), target.to(dev))         # models/ge_nacagat/main.py:33
    assert abs(loss.item() - float(g[f"{case}/loss"])) < 1e-4
    loss.backward()
    for n, prm in model.named_parameters():
        ref = g[f"{case}/grad/{n}"]
        got = sub(prm.grad if prm.grad is not None else torch.zeros_like(prm), 256).cpu()
        scale = max(float(ref.abs().max()), 1e-5)      # shift-invariant biases have ~1e-8 'gradients'
        assert float((got - ref).abs().max()) / scale < 5e-3, (n, float((got - ref).abs().max()) / scale)


def test_ge_model_bf16_bag_against_the_oracle(dev):
    """bf16 storage of the patch matrix / H_bag: Y within 1e-3 of the fp32 oracle, and close to the oracle fed the same
    stored values."""
    m, seed = 2000, 4343
    model, sd = build_ge(dev, seed, bag_dtype=torch.bfloat16)
    wsi, _ = C.ge_model_inputs(m, seed + 1)
    y, att = model(wsi=wsi.to(dev))
    y32, _ = O.ge_nacagat_forward(sd, wsi)
    assert float((y.cpu() - y32).abs().max()) < 1e-3
    y16, att16 = O.ge_nacagat_forward(sd, wsi, bag_storage=torch.bfloat16)
    assert float((y.cpu() - y16).abs().max()) < 2e-4
    assert relmax(att["path"], att16["path"]) < 2e-3


def test_ge_model_training_step_at_15000_rows(dev):
    """The long-bag shape itself (M = 15 000, training mode with every dropout on): one forward + backward; the map's
    rows are distributions, Y is a distribution, every parameter receives a finite gradient."""
    m, seed = 15000, 4444
    model, _ = build_ge(dev, seed, bag_dtype=torch.bfloat16)
    model.train()
    wsi, target = C.ge_model_inputs(m, seed + 1)
    y, att = model(wsi=wsi.to(dev))
    assert att["attn"].shape == (m, m) and att["path"].shape == (1, m)
    assert abs(float(y.sum()) - 1.0) < 1e-5
    assert float((att["attn"].sum(-1) - 1).abs().max()) < 1e-4
    torch.nn.functional.cross_entropy(y.unsqueeze(0), target.to(dev)).backward()
    for n, prm in model.named_parameters():
        assert prm.grad is not None and bool(torch.isfinite(prm.grad).all()), n
        assert float(prm.grad.abs().max()) > 0, n


@pytest.mark.parametrize("size,d", [("small", 128), ("big", 512)])
def test_ge_model_small_and_big_equal_oracle(dev, size, d):
    """model_size='small' (embed 128: head dimensions 128 and 16) and 'big' (embed 512: one head of 512, eight of 64;
    models/ge_nacagat/ge_nacagat.py:12-17) against the oracle; a head dimension the kernels are not built for is refused
    with a message -- not run some other way."""
    m, seed = 300, 4545
    shapes = C.ge_model_shapes(d=d)
    sd = syn.fill_state_dict(shapes, seed)
    model = GeneExprNarrowContextualAttentionGateTransformer(model_size=size)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    wsi, target = C.ge_model_inputs(m, seed + 1)
    y, att = model(wsi=wsi.to(dev))
    p = _leaf(sd)
    yo, atto = O.ge_nacagat_forward(p, wsi)
    assert float((y.detach().cpu() - yo).abs().max()) < 1e-4
    assert relmax(att["path"], atto["path"]) < 1e-3
    assert ((att["attn"].cpu() - atto["attn"]).abs() / atto["attn"].clamp_min(1e-30)).max().item() < 1e-3
    torch.nn.functional.cross_entropy(y.unsqueeze(0), target.to(dev)).backward()
    O.ge_ce_loss(yo, target).backward()
    for n, prm in model.named_parameters():
        ref = p[n].grad
        scale = max(float(ref.abs().max()), 1e-5)
        assert float((prm.grad.cpu() - ref).abs().max()) / scale < 5e-3, n
    qkv = torch.zeros(1, 32, 3 * 96, device=dev)
    with pytest.raises(RuntimeError, match="head dimension 96"):
        ops.BagSelfAttentionFn.apply(qkv, 1, 0.0, True)
