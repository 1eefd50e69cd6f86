"""GPU parity of the token-tail kernels (rows H5-H8: K3 CAG is in test_gpu_coattn_nacagat.py, here
K4 set-Transformer, K5 gated attention-MIL pooling, K6 fusion + survival head) against the
reference's golden vectors, plus training-mode consistency of forward and backward dropout masks."""
import pytest
import torch
import torch.nn as nn

import cases as C
from multimodal_path_omic_amd import ops
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.blocks import AttentionNetGated
from multimodal_path_omic_amd.fusion import ConcatFusion
from multimodal_path_omic_amd.transformer import make_set_transformer

pytestmark = pytest.mark.gpu
sub = syn.subsample


def relerr(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def check_grads(g, prefix, names, grads, tol=2e-3):
    for n, gr in zip(names, grads):
        ref = g[prefix + n]
        scale = max(float(ref.abs().max()), 1e-5)
        err = float((sub(gr).cpu() - ref).abs().max()) / scale
        assert err < tol, (n, err)


def test_encoder_matches_golden(dev, golden):
    g = golden("encoder")
    sd = syn.fill_state_dict(C.encoder_shapes("path_transformer"), 600)
    enc = make_set_transformer(C.E, 0.25)
    enc.load_state_dict({k[len("path_transformer."):]: v for k, v in sd.items()}, strict=True)
    enc.to(dev).eval()
    x, probe = C.encoder_inputs()
    xd = x.to(dev).requires_grad_(True)
    y = enc(xd)                                             # (T, d): the reference's unbatched call
    assert relerr(y, g["y"]) < 1e-4
    params = dict(enc.named_parameters())
    tensors = [xd] + [params[k[len("path_transformer."):]] for k in sd]
    grads = torch.autograd.grad((y * probe.to(dev)).sum(), tensors)
    check_grads(g, "grad/", ["x"] + list(sd), grads)
    # a window of slides: every slide gets the single-slide result
    xb = torch.stack([x, 2 * x, -x]).to(dev)
    yb = enc(xb)
    assert relerr(yb[0], y) < 1e-5 and relerr(yb[1], enc(2 * x.to(dev))) < 1e-5


@pytest.mark.parametrize("case", list(C.POOL_CASES))
def test_gated_pool_matches_golden(dev, golden, case):
    g = golden("pool")
    l, seed = C.POOL_CASES[case]
    sd = syn.fill_state_dict(C.pool_shapes("path_attention_head", "path_rho"), seed)
    head = AttentionNetGated(n_classes=1, input_dim=C.E, hidden_dim=C.E)
    rho = nn.Sequential(nn.Linear(C.E, C.E), nn.ReLU(), nn.Dropout(0.25))
    head.load_state_dict({k[len("path_attention_head."):]: v for k, v in sd.items() if k.startswith("path_attention_head.")})
    rho.load_state_dict({k[len("path_rho."):]: v for k, v in sd.items() if k.startswith("path_rho.")})
    head.to(dev).eval(); rho.to(dev).eval()
    x, probe_h, probe_a = C.pool_inputs(l, seed + 1)
    xd = x.to(dev).requires_grad_(True)
    a, h = ops.gated_pool(xd.unsqueeze(0), head, rho, training=False)
    assert relerr(a[0], g[f"{case}/A"]) < 1e-4
    assert relerr(h[0], g[f"{case}/h"]) < 1e-4
    # the module's stand-alone forward() keeps the reference signature: (A (L,1), x)
    a2, x2 = head(xd)
    assert a2.shape == (l, 1) and x2 is xd and relerr(a2.t(), g[f"{case}/A"]) < 1e-4
    hp, rp = dict(head.named_parameters()), dict(rho.named_parameters())
    names = ["x"] + list(sd)
    tensors = [xd] + [hp[k[len("path_attention_head."):]] if k.startswith("path_attention_head.") else rp[k[len("path_rho."):]]
                      for k in sd]
    loss = (h[0] * probe_h.to(dev)).sum() + (a[0] * probe_a.to(dev)).sum()
    check_grads(g, f"{case}/grad/", names, torch.autograd.grad(loss, tensors))


def test_fusion_head_matches_golden(dev, golden):
    g = golden("fusion")
    sd = syn.fill_state_dict(C.FUSION_SHAPES, 700)
    fus = ConcatFusion(dims=[C.E, C.E], hidden_size=C.E, output_size=C.E)
    cls = nn.Linear(C.E, 4)
    fus.load_state_dict({k[len("fusion_layer."):]: v for k, v in sd.items() if k.startswith("fusion_layer.")})
    cls.load_state_dict({k[len("classifier."):]: v for k, v in sd.items() if k.startswith("classifier.")})
    fus.to(dev); cls.to(dev)
    hp, ho, probe = C.fusion_inputs()
    hpd, hod = hp.to(dev).requires_grad_(True), ho.to(dev).requires_grad_(True)
    assert relerr(fus(hpd, hod), g["h"]) < 1e-4             # stand-alone ConcatFusion.forward(*x)
    hz, sv, y = ops.fusion_head(hpd.unsqueeze(0), hod.unsqueeze(0), fus, cls)
    for t, k in ((hz, "hazards"), (sv, "survs"), (y, "Y")):
        assert float((t.cpu() - g[k]).abs().max()) < 1e-5
    p = probe.to(dev)
    loss = (hz * p).sum() + (sv * p.flip(1)).sum() + (y * p * 0.5).sum()
    fp, cp = dict(fus.named_parameters()), dict(cls.named_parameters())
    names = ["h_path", "h_omic"] + list(sd)
    tensors = [hpd, hod] + [fp[k[len("fusion_layer."):]] if k.startswith("fusion_layer.") else cp[k[len("classifier."):]]
                            for k in sd]
    check_grads(g, "grad/", names, torch.autograd.grad(loss, tensors))


def _directional(fn, x, v, eps=1e-2):
    return (fn(x + eps * v) - fn(x - eps * v)) / (2 * eps)


@pytest.mark.parametrize("which", ["encoder", "pool"])
def test_training_dropout_forward_backward_use_the_same_masks(dev, which):
    """With the dropout counter pinned, the forward is a deterministic function of its input, so the
    analytic gradient (masks regenerated in backward) must match a finite difference."""
    torch.manual_seed(5)
    gsyn = syn.rng(31)
    b, t = 3, C.N_OMIC
    x = syn.normal(gsyn, (b, t, C.E)).to(dev)
    v = syn.normal(gsyn, (b, t, C.E)).to(dev)
    probe = syn.normal(gsyn, (b, t, C.E)).to(dev)
    if which == "encoder":
        mod = make_set_transformer(C.E, 0.25).to(dev).train()

        def f(inp):
            ops._rng_calls = 1234
            return (mod(inp) * probe).sum().double()
    else:
        head = AttentionNetGated(n_classes=1, input_dim=C.E, hidden_dim=C.E).to(dev).train()
        rho = nn.Sequential(nn.Linear(C.E, C.E), nn.ReLU(), nn.Dropout(0.25)).to(dev).train()

        def f(inp):
            ops._rng_calls = 1234
            a, h = ops.gated_pool(inp, head, rho, training=True)
            return ((h * probe[:, 0]).sum() + a.sum() * 0.1).double()
    xr = x.clone().requires_grad_(True)
    out = f(xr)
    out.backward()
    ana = float((xr.grad * v).sum())
    num = float(_directional(f, x, v))
    assert abs(ana - num) < 5e-2 * max(1.0, abs(num)), (ana, num)
    # dropout is really on: another counter gives another value
    ops._rng_calls = 99999
    other = float((mod(x) * probe).sum()) if which == "encoder" else float(ops.gated_pool(x, head, rho, True)[1].sum())
    ops._rng_calls = 1234
    same = float((mod(x) * probe).sum()) if which == "encoder" else float(ops.gated_pool(x, head, rho, True)[1].sum())
    ops._rng_calls = 99999
    again = float((mod(x) * probe).sum()) if which == "encoder" else float(ops.gated_pool(x, head, rho, True)[1].sum())
    assert other == again and other != same


def test_omic_snn_matches_stock_modules_and_alpha_dropout(dev):
    """self.G (models/mcat/mcat.py:32-45): grouped-GEMM family == the stock nn.Sequential stack in eval mode
    (values and gradients); in training mode AlphaDropout keeps ~(1-p) of the activations, replaces the rest by
    the constant a*alpha'+b, and forward/backward use the same mask (finite differences, counter pinned)."""
    import torch.nn as nn
    sizes = [100, 31, 256, 8, 300, 64]
    torch.manual_seed(3)
    G = nn.ModuleList([nn.Sequential(
        nn.Sequential(nn.Linear(s, C.E), nn.ELU(), nn.AlphaDropout(p=0.25, inplace=False)),
        nn.Sequential(nn.Linear(C.E, C.E), nn.ELU(), nn.AlphaDropout(p=0.25, inplace=False))) for s in sizes]).to(dev)
    g = syn.rng(41)
    b = 5
    xs = [syn.normal(g, (b, s)).to(dev) for s in sizes]
    probe = syn.normal(g, (b, len(sizes), C.E)).to(dev)
    G.eval()
    ref = torch.stack([m(x) for m, x in zip(G, xs)], dim=1)
    got = ops.omic_snn(xs, G, training=False)
    assert relerr(got, ref) < 1e-5
    gr_ref = torch.autograd.grad((ref * probe).sum(), list(G.parameters()))
    gr_got = torch.autograd.grad((got * probe).sum(), list(G.parameters()))
    for a, r in zip(gr_got, gr_ref):
        assert float((a - r).abs().max()) / max(float(r.abs().max()), 1e-6) < 1e-4
    # training mode
    G.train()
    ops._rng_calls = 777
    y = ops.omic_snn(xs, G, training=True)
    p = 0.25
    a_c = 1.0 / ((1 - p) * (1 + p * 1.7580993408473766 ** 2)) ** 0.5
    const = a_c * (-1.7580993408473766) + (a_c * 1.7580993408473766 * p)
    frac = float(((y - const).abs() < 1e-6).float().mean())
    assert abs(frac - p) < 0.03, frac
    v = [syn.normal(g, tuple(q.shape)).to(dev) * 0.5 for q in G.parameters()]

    def f(scale):
        ops._rng_calls = 777
        with torch.no_grad():
            for q, d in zip(G.parameters(), v):
                q.add_(scale * d)
        out = float((ops.omic_snn(xs, G, True) * probe).sum().double())
        with torch.no_grad():
            for q, d in zip(G.parameters(), v):
                q.sub_(scale * d)
        return out
    ops._rng_calls = 777
    yy = ops.omic_snn(xs, G, True)
    grads = torch.autograd.grad((yy * probe).sum(), list(G.parameters()))
    ana = sum(float((gq * d).sum()) for gq, d in zip(grads, v))
    eps = 1e-3
    num = (f(eps) - f(-eps)) / (2 * eps)
    assert abs(ana - num) < 3e-2 * max(1.0, abs(num)), (ana, num)


def test_ces_loss_kernel_matches_reference_kat_golden_and_oracle_gradients(dev, golden):
    """'ces' loss in one launch each way (mpo_ces_loss_*): the reference's own known answers (models/loss.py:104-123),
    the golden per-slide values, and gradients against the oracle incl. the clamp(min=eps) edges and the
    broadcast (loss.sum()/acc) upstream gradient."""
    from oracle import mpo_oracle as O
    hz = torch.tensor([[0.51, 0.52, 0.49, 0.48]], device=dev)
    s = torch.tensor([[0.5, 0.4, 0.2, 0.1]], device=dev)
    y0 = torch.tensor([0], device=dev)
    assert ops.ces_loss(hz, s, y0, torch.tensor([0.0], device=dev))[0].item() == pytest.approx(0.6782951951026917, abs=1e-6)
    assert ops.ces_loss(hz, s, y0, torch.tensor([1.0], device=dev))[0].item() == pytest.approx(0.1732867956161499, abs=1e-6)
    g = golden("loss")
    hz_all = g["hazards"].reshape(8, 4)
    sv_all = torch.cumprod(1 - hz_all, dim=1)
    y = torch.arange(8) % 4
    c = (torch.arange(8) // 4).float()
    per, risk = ops.ces_loss(hz_all.to(dev), sv_all.to(dev), y.to(dev), c.to(dev))
    for i in range(8):
        assert per[i].item() == pytest.approx(float(g[f"ces/{i}"]), rel=1e-5, abs=1e-6)
    torch.testing.assert_close(risk.cpu(), -sv_all.sum(1), rtol=1e-6, atol=1e-6)
    # gradients: random slides plus rows that sit below eps (clamped: no gradient flows there)
    gen = torch.Generator().manual_seed(5)
    hz_r = torch.rand(32, 4, generator=gen) * 0.98 + 0.01
    hz_r[3, 2] = 1e-9
    hz_r[7] = torch.tensor([1.0 - 1e-9, 0.5, 0.5, 0.5])           # S collapses below eps from class 0 on
    y_r = torch.randint(0, 4, (32,), generator=gen)
    y_r[3], y_r[7] = 2, 1
    c_r = torch.randint(0, 2, (32,), generator=gen).float()
    w = torch.rand(32, generator=gen)
    for weights in (w, None):
        hz_o = hz_r.clone().requires_grad_(True)
        sv_o = torch.cumprod(1 - hz_r, dim=1).requires_grad_(True)
        per_o = torch.stack([O.ces_loss(hz_o[i:i + 1], sv_o[i:i + 1], y_r[i:i + 1], c_r[i:i + 1]) for i in range(32)])
        ((per_o * weights).sum() if weights is not None else per_o.sum() / 8).backward()
        hz_d = hz_r.to(dev).requires_grad_(True)
        sv_d = sv_o.detach().to(dev).requires_grad_(True)
        per_d, _ = ops.ces_loss(hz_d, sv_d, y_r.to(dev), c_r.to(dev))
        torch.testing.assert_close(per_d.cpu(), per_o.detach(), rtol=1e-5, atol=1e-6)
        ((per_d * weights.to(dev)).sum() if weights is not None else per_d.sum() / 8).backward()
        torch.testing.assert_close(hz_d.grad.cpu(), hz_o.grad, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(sv_d.grad.cpu(), sv_o.grad, rtol=1e-5, atol=1e-7)


def test_fused_head_and_loss_launch_equals_the_separate_launches(dev):
    """Training-step form (mpo_fusion_head_loss_*): head, `ces` loss and both backward passes in one launch must give
    the separate launches' values -- same arithmetic, registers instead of memory in between -- and the same
    parameter / input gradients; a gradient other than the announced slide weights is refused."""
    from oracle import mpo_oracle as O
    torch.manual_seed(11)
    b = 32
    fus = ConcatFusion(dims=[C.E, C.E], hidden_size=C.E, output_size=C.E).to(dev)
    cls = nn.Linear(C.E, 4).to(dev)
    hcat = torch.randn(b, 2 * C.E, device=dev)
    y = torch.randint(0, 4, (b,), device=dev)
    c = torch.randint(0, 2, (b,), device=dev).float()
    w = torch.full((b,), 1.0 / 32, device=dev)
    params = list(fus.parameters()) + list(cls.parameters())

    h1 = hcat.clone().requires_grad_(True)
    hz, sv, yy = ops.fusion_head_cat(h1, fus, cls)
    per, risk = ops.ces_loss(hz, sv, y, c)
    g_ref = torch.autograd.grad(per, [h1] + params, grad_outputs=w)

    h2 = hcat.clone().requires_grad_(True)
    per2, risk2, hz2, sv2, yy2 = ops.fusion_head_loss_cat(h2, fus, cls, y, c, w)
    for a, r in ((per2, per), (risk2, risk), (hz2, hz), (sv2, sv), (yy2, yy)):     # (same formulas; the compiler may contract
        torch.testing.assert_close(a, r, rtol=1e-6, atol=1e-7)                      #  multiply-adds differently in the fused body)
    assert not hz2.requires_grad and not risk2.requires_grad
    g_new = torch.autograd.grad(per2, [h2] + params, grad_outputs=w, retain_graph=True)
    for a, r in zip(g_new, g_ref):
        torch.testing.assert_close(a, r, rtol=1e-5, atol=1e-8)
    # against the oracle's loss on the kernel's own hazards (the head itself is covered by the golden test above)
    per_o = torch.stack([O.ces_loss(hz2[i:i + 1].cpu(), sv2[i:i + 1].cpu(), y[i:i + 1].cpu(), c[i:i + 1].cpu()) for i in range(b)])
    torch.testing.assert_close(per2.cpu(), per_o, rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError, match="slide_weight"):
        torch.autograd.grad(per2, [h2], grad_outputs=w.clone())


def test_interleaved_pool_output_is_the_concatenation(dev):
    """gated_pool_stacked(interleave=True) writes h as (B, [branch 0 | branch 1]) -- the row ConcatFusion reads -- and
    takes its gradient in that layout: values and every gradient equal the (branches, B, d) form."""
    torch.manual_seed(3)
    b, l, d = 5, 6, C.E
    heads = [AttentionNetGated(n_classes=1, input_dim=d, hidden_dim=d).to(dev).eval() for _ in range(2)]
    rhos = [nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Dropout(0.25)).to(dev).eval() for _ in range(2)]
    tokens = torch.randn(2, b, l, d, device=dev)
    probe = torch.randn(b, 2 * d, device=dev)
    params = [p for m in heads + rhos for p in m.parameters()]

    t0 = tokens.clone().requires_grad_(True)
    sc0, h0 = ops.gated_pool_stacked(t0, heads, rhos, False)
    g0 = torch.autograd.grad((h0.transpose(0, 1).reshape(b, -1) * probe).sum() + sc0.sum(), [t0] + params)
    t1 = tokens.clone().requires_grad_(True)
    sc1, h1 = ops.gated_pool_stacked(t1, heads, rhos, False, interleave=True)
    assert h1.shape == (b, 2 * d)
    assert torch.equal(h1, h0.transpose(0, 1).reshape(b, -1)) and torch.equal(sc1, sc0)
    g1 = torch.autograd.grad((h1 * probe).sum() + sc1.sum(), [t1] + params)
    for a, r in zip(g1, g0):
        torch.testing.assert_close(a, r, rtol=1e-6, atol=1e-7)
    # training: dropout on the interleaved rows at the configured rate, zeros where dropped, kept values scaled
    for m in heads + rhos:
        m.train()
    _, ht = ops.gated_pool_stacked(tokens, heads, rhos, True, interleave=True)
    dropped = float((ht == 0).float().mean())
    assert 0.5 < dropped < 0.75                      # relu zeros (~1/2) plus a quarter of the rest


def test_fast_gemm_body_equals_the_general_body(dev):
    """The branch-free GEMM body (regular products: every set-Transformer / pooling / fusion product of the model) against
    the general body through a whole training-mode forward + backward of the branch-batched encoder and pooling head:
    dropout epilogues, ReLU / tanh / sigmoid value gates and regenerated-dropout gates, dx and dW layouts.  Same
    products in the same order and the same random streams: equal up to how the compiler contracts multiply-adds
    (bias-gradient sums differ in the last bit), far below any parity tolerance -- a wrong mask or index would not be."""
    from multimodal_path_omic_amd import _lib as L
    torch.manual_seed(5)
    b, l, d = 32, 6, C.E
    enc = [make_set_transformer(d, dropout=0.25).to(dev).train() for _ in range(2)]
    heads = [AttentionNetGated(n_classes=1, input_dim=d, hidden_dim=d).to(dev).train() for _ in range(2)]
    rhos = [nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Dropout(0.25)).to(dev).train() for _ in range(2)]
    x = torch.randn(2, b, l, d, device=dev)
    probe = torch.randn(b, 2 * d, device=dev)
    # the omic SNNs (AlphaDropout + ELU derivative gates, 32-row weight gradients) and the fusion MLP (K = 32 products)
    G = nn.ModuleList([nn.Sequential(nn.Sequential(nn.Linear(d, d), nn.ELU(), nn.AlphaDropout(0.25)),
                                     nn.Sequential(nn.Linear(d, d), nn.ELU(), nn.AlphaDropout(0.25))) for _ in range(l)]).to(dev).train()
    fus = ConcatFusion(dims=[d, d], hidden_size=d, output_size=d).to(dev)
    cls = nn.Linear(d, 4).to(dev)
    omics = [torch.randn(b, d, device=dev) for _ in range(l)]
    params = [p for m in enc + heads + rhos + [G, fus, cls] for p in m.parameters()]

    def run():
        ops._rng_calls = 1000                                   # same dropout streams in both runs
        t = x.clone().requires_grad_(True)
        g_bag = ops.omic_snn(omics, G, True)
        tok = ops.encoder_stacked(t + torch.stack([g_bag, g_bag]), [list(e.layers) for e in enc], True)
        sc, h = ops.gated_pool_stacked(tok, heads, rhos, True, interleave=True)
        hz, sv, y = ops.fusion_head_cat(h, fus, cls)
        g = torch.autograd.grad((h * probe).sum() + sc.sum() + (hz * sv).sum() + y[:, 0].sum(), [t] + params)
        return [h.detach().clone(), sc.detach().clone(), hz.detach().clone()] + [gi.clone() for gi in g]

    fast = run()
    was = L.lib().mpo_set_gemm_fast_path(0)
    try:
        general = run()
    finally:
        L.lib().mpo_set_gemm_fast_path(was)
    assert was == 1
    for i, (a, r) in enumerate(zip(fast, general)):
        torch.testing.assert_close(a, r, rtol=1e-5, atol=1e-6 * float(r.abs().max()), msg=lambda m, i=i: f"tensor {i}: {m}")


def test_step_counters_bump_is_one_launch_for_both(dev):
    e = torch.tensor([41], dtype=torch.int64, device=dev)
    t = torch.tensor([6], dtype=torch.int32, device=dev)
    ops.bump_step_counters(e, t)
    ops.bump_step_counters(e, None)
    ops.bump_step_counters(None, t)
    assert e.item() == 43 and t.item() == 8


@pytest.mark.parametrize("kind", ["bilinear", "gated_concat"])
def test_fusion_next_rows_match_golden(dev, golden, kind):
    """Row f4 on the GPU: BilinearFusion / GatedConcatFusion (HIP GEMMs + element-wise device ops) against the
    reference's golden outputs and gradients, per slide (the reference's 1-D call) and for a window of slides."""
    from multimodal_path_omic_amd.fusion import BilinearFusion, GatedConcatFusion
    g = golden("fusion_next")
    if kind == "bilinear":
        fus = BilinearFusion(dim1=C.E, dim2=C.E, output_size=C.E)
        sd = syn.fill_state_dict(C.BILINEAR_SHAPES, 710, 3.0)
    else:
        fus = GatedConcatFusion(dims=[C.E, C.E], hidden_size=C.E, output_size=C.E)
        sd = syn.fill_state_dict(C.GATED_CONCAT_SHAPES, 720)
    assert {k: tuple(v.shape) for k, v in fus.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    fus.load_state_dict(sd, strict=True)
    fus.to(dev).eval()
    hp, ho, _ = C.fusion_inputs()
    hpd, hod = hp.to(dev).requires_grad_(True), ho.to(dev).requires_grad_(True)
    y = fus(hpd, hod)
    assert y.shape == g[f"{kind}/out"].shape
    assert relerr(y, g[f"{kind}/out"]) < 1e-4
    probe = syn.normal(syn.rng(711), tuple(y.shape)).to(dev)
    names = ["h_path", "h_omic"] + list(sd)
    params = dict(fus.named_parameters())
    gs = torch.autograd.grad((y * probe).sum(), [hpd, hod] + [params[k] for k in sd])
    check_grads(g, f"{kind}/grad/", names, gs)
    # window form: rows are independent slides
    hw = torch.stack([hp, ho * 0.5, -hp]).to(dev), torch.stack([ho, hp, ho * 2.0]).to(dev)
    yw = fus(*hw)
    for b in range(3):
        assert relerr(yw[b], fus(hw[0][b], hw[1][b])) < 1e-5


def test_gated_concat_loads_a_reference_checkpoint_without_gate_entries(dev):
    """The reference never registers its gates (models/fusion.py:25-27), so its checkpoints have no gates.* keys."""
    from multimodal_path_omic_amd.fusion import GatedConcatFusion
    fus = GatedConcatFusion(dims=[C.E, C.E], hidden_size=C.E, output_size=C.E)
    before = {k: v.clone() for k, v in fus.gates.state_dict().items()}
    ref_like = {k: v for k, v in syn.fill_state_dict(C.GATED_CONCAT_SHAPES, 720).items() if not k.startswith("gates.")}
    fus.load_state_dict(ref_like, strict=True)
    for k, v in fus.gates.state_dict().items():
        assert torch.equal(v, before[k])


@pytest.mark.parametrize("fusion", ["bilinear", "gated_concat"])
def test_model_with_other_fusion_matches_oracle(dev, fusion):
    """Whole MCAT with `fusion=` bilinear / gated_concat (models/mcat/mcat.py:69-79): hazards and gradients vs the oracle."""
    from multimodal_path_omic_amd.harness import ces_loss
    from multimodal_path_omic_amd.models import MultimodalCoAttentionTransformer
    from oracle import mpo_oracle as O
    omic_sizes, m, seed = [64, 100, 256, 31, 8, 300], 900, 7170
    model = MultimodalCoAttentionTransformer(omic_sizes=omic_sizes, fusion=fusion)
    sd = syn.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    wsi, omics, _, _ = C.model_inputs(m, omic_sizes, seed + 1)
    hz, sv, y, _ = model(wsi=wsi.to(dev), omics=[o.to(dev) for o in omics])
    label, censor = torch.tensor([2]), torch.tensor([0.0])
    ces_loss(hz, sv, label.to(dev), censor.to(dev)).backward()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hz_o, sv_o, _, _ = O.mcat_forward(p, wsi, omics, fusion=fusion)
    assert float((hz.cpu() - hz_o).abs().max()) < 1e-4
    O.ces_loss(hz_o, sv_o, label, censor).backward()
    for n, prm in model.named_parameters():
        ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
        scale = max(float(ref.abs().max()), 1e-4)
        assert float((prm.grad.cpu() - ref).abs().max()) / scale < (1e-2 if n.startswith("H.") else 2e-3), n


@pytest.mark.parametrize("rows,k,n,act", [(1500, 256, 768, "none"), (515, 512, 256, "relu"), (2049, 192, 64, "tanh"), (1024, 64, 512, "sigmoid"),
                                          (4100, 256, 512, "relu"), (15000, 512, 256, "none")])
def test_many_row_products_equal_torch(dev, rows, k, n, act):
    """Products with >= 512 rows (the gene-expression model's set-Transformer / pooling over the rows of a bag) run on 32 x 64
    tiles (csrc/gemm_f32_rows.hip): forward and the input gradient, ragged row counts; from 2 048 rows the weight and bias
    gradients run with the row axis cut over the grid (csrc/gemm_f32_longk.hip).  Against torch."""
    g = syn.rng(9100 + rows)
    x, w, b, probe = syn.normal(g, (rows, k)), syn.normal(g, (n, k)) * 0.1, syn.normal(g, (n,)), syn.normal(g, (rows, n))
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xd, wd, bd, act)
    (y * probe.to(dev)).sum().backward()
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    yr = {"none": lambda t: t, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act](yr)
    (yr * probe.double()).sum().backward()

    def rel(a, r):
        return float((a.detach().double().cpu() - r).abs().max() / r.abs().max())
    assert rel(y, yr) < 1e-5
    assert rel(xd.grad, xr.grad) < 1e-5 and rel(wd.grad, wr.grad) < 1e-4 and rel(bd.grad, br.grad) < 1e-4
