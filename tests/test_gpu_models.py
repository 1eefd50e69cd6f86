"""Whole-model GPU parity (row H1): hazards / survs / Y / attention maps and the parameter
gradients of the `ces` loss against the reference's golden vectors (fp32 bag) and the oracle
(bf16-stored bag).  North-star bar: 1e-3 on hazards, relative on attention maps."""
import pytest
import torch

import cases as C
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.harness import ces_loss
from multimodal_path_omic_amd.models import (MultimodalCoAttentionTransformer,
                                             NarrowContextualAttentionGateTransformer)
from multimodal_path_omic_amd.ops import BagBatch
from oracle import mpo_oracle as O

pytestmark = pytest.mark.gpu
sub = syn.subsample


def relerr(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def build(kind, omic_sizes, seed, dev, bag_dtype=torch.float32):
    cls = MultimodalCoAttentionTransformer if kind == "mcat" else NarrowContextualAttentionGateTransformer
    model = cls(omic_sizes=omic_sizes, bag_dtype=bag_dtype)
    sd = syn.fill_state_dict(C.model_shapes(omic_sizes, kind == "nacagat"), seed)
    model.load_state_dict(sd, strict=True)              # reference state_dict layout loads as is
    return model.to(dev).eval(), sd


MCAT_CASES = [c for c in C.MODEL_CASES if c.startswith("mcat")]
NACAGAT_CASES = [c for c in C.MODEL_CASES if c.startswith("nacagat")]


@pytest.mark.parametrize("case", MCAT_CASES + NACAGAT_CASES)
def test_model_matches_reference_golden(dev, golden, case):
    g = golden("models")
    kind, m, omic_sizes, seed = C.MODEL_CASES[case]
    model, sd = build(kind, omic_sizes, seed, dev)
    wsi, omics, label, censor = C.model_inputs(m, omic_sizes, seed + 1)
    wsi_d, om_d = wsi.to(dev), [o.to(dev) for o in omics]
    kw = dict(inference=True) if kind == "mcat" else {}
    hz, sv, y, att = model(wsi=wsi_d, omics=om_d, **kw)
    assert hz.shape == (1, 4) and att["path"].shape == (1, len(omic_sizes)) and att["coattn"].shape == (len(omic_sizes), m)
    # the DataLoader convention gives the same result
    hz_b, *_ = model(wsi=wsi_d.unsqueeze(0), omics=[o.unsqueeze(0) for o in om_d], **kw)
    assert torch.equal(hz, hz_b)
    if kind == "mcat":
        assert model(wsi=wsi_d, omics=om_d)[3]["coattn"] is None      # training-style call: no map
    assert float((hz.detach().cpu() - g[f"{case}/hazards"]).abs().max()) < 1e-4
    assert float((sv.cpu() - g[f"{case}/survs"]).abs().max()) < 1e-4
    assert float((y.cpu() - g[f"{case}/Y"]).abs().max()) < 1e-4
    assert relerr(att["path"], g[f"{case}/A_path"]) < 1e-3
    assert relerr(att["omic"], g[f"{case}/A_omic"]) < 1e-3
    ga = g[f"{case}/A_coattn_sub"]
    assert ((sub(att["coattn"]).cpu() - ga).abs() / ga.clamp_min(1e-30)).max().item() < 1e-3        # the north-star map bar
    loss = ces_loss(hz, sv, label.to(dev), censor.to(dev))
    assert abs(loss.item() - float(g[f"{case}/loss"])) < 1e-4
    loss.backward()
    for n, p in model.named_parameters():
        ref = g[f"{case}/grad/{n}"]
        got = sub(p.grad if p.grad is not None else torch.zeros_like(p), 256).cpu()
        scale = max(float(ref.abs().max()), 1e-5)      # shift-invariant biases have ~1e-8 'gradients'
        assert float((got - ref).abs().max()) / scale < 5e-3, (n, float((got - ref).abs().max()) / scale)


@pytest.mark.parametrize("kind", ["mcat", "nacagat"])
def test_model_bf16_bag_within_north_star(dev, kind):
    """bf16 storage of the patch matrix and H_bag: hazards stay within 1e-3 of the fp32 oracle."""
    omic_sizes, m, seed = [256] * 6, 3000, 4242
    model, sd = build(kind, omic_sizes, seed, dev, bag_dtype=torch.bfloat16)
    wsi, omics, _, _ = C.model_inputs(m, omic_sizes, seed + 1)
    hz, sv, y, att = model(wsi=wsi.to(dev), omics=[o.to(dev) for o in omics])
    fwd = O.mcat_forward if kind == "mcat" else O.nacagat_forward
    hz_o, sv_o, y_o, _ = fwd(sd, wsi, omics)
    assert float((hz.cpu() - hz_o).abs().max()) < 1e-3
    assert float((sv.cpu() - sv_o).abs().max()) < 1e-3
    # Gradients: bf16 is a STORAGE format, so the oracle is fed the same stored values (patch matrix, H-layer
    # GEMM operands/output, H_bag rounded to bf16; fp32 arithmetic) and must then agree closely.  What remains
    # is the bf16 rounding of d(H_bag) on its way into dW_H.
    label, censor = torch.tensor([2]), torch.tensor([0.0])
    from multimodal_path_omic_amd import ops
    handoffs = ops.stats["colsum_handoffs"]
    ces_loss(hz, sv, label.to(dev), censor.to(dev)).backward()
    if kind == "mcat":          # H.0.bias's gradient must have come out of the co-attention backward kernel itself
        assert ops.stats["colsum_handoffs"] == handoffs + 1
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hz_s, sv_s, _, _ = fwd(p, wsi, omics, bag_storage=torch.bfloat16)
    assert float((hz.cpu() - hz_s).abs().max()) < 2e-4
    O.ces_loss(hz_s, sv_s, label, censor).backward()
    for n, prm in model.named_parameters():
        ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
        scale = max(float(ref.abs().max()), 1e-4)
        err = float((prm.grad.cpu() - ref).abs().max()) / scale
        assert err < (2e-2 if n.startswith("H.") else 1e-2), (n, err)


@pytest.mark.parametrize("case", ["mcat_m15000", "nacagat_m15000", "mcat_m2000", "nacagat_m2000"])
def test_model_bf16_bag_vs_fp32_reference_golden(dev, golden, case, record_property):
    """The BENCHMARKED storage mode at the benchmarked bag length against the fp32 REFERENCE's own outputs (fixtures
    generated by the imported reference, tests/golden/make_golden.py): bf16 patch matrix, bf16 patch-layer operands and
    bf16 H_bag, fp32 accumulation, fp32 6 x d tail.
      * hazards / survs / Y: the north_star bar, 1e-3 absolute.
      * co-attention map: element-wise RELATIVE error.  Rounding the 15 000 x 1024 patch matrix to bf16
        ALONE -- fp32 arithmetic everywhere else, measured with the CPU oracle on these fixtures -- already moves the map
        by 3.1e-3 (MCAT) / 2.9e-2 (NaCAGaT: the narrow gate multiplies the logit error), so 1e-3 on the map is not
        reachable in this storage mode by any kernel.  What the kernels must NOT do is add to the storage rounding: the
        error against the fp32 reference is held to the error the ORACLE makes when it is fed the same stored values
        (bag_storage=bf16; fp32 arithmetic), +25 %, and to a fraction of it against that same-storage oracle.  Where that
        floor comes from, storage point by storage point (X / W_H / H_bag: 3.1e-3 / 3.2e-3 / 3.6e-3 alone, 5.2e-3 together
        for MCAT at M = 15 000), is pinned on the CPU by test_oracle_golden.py::test_bf16_storage_map_error_by_storage_point.
      * the pooling map over the 6 co-attended tokens ('path'): the same rule -- 1e-3 where the storage floor allows it
        (MCAT: 1.5e-4), else the same-storage oracle's own error + 25 % (NaCAGaT: 8.3e-3 from storage alone).
      * fixed ceilings pinned to the measured values on top of the relative-to-floor bars: a change that moved the oracle's
        storage emulation together with the kernels would still be caught.
    The measured margins are printed (pytest -rA) and quoted in DESIGN.md section 4."""
    g = golden("models")
    kind, m, omic_sizes, seed = C.MODEL_CASES[case]
    model, sd = build(kind, omic_sizes, seed, dev, bag_dtype=torch.bfloat16)
    wsi, omics, label, censor = C.model_inputs(m, omic_sizes, seed + 1)
    kw = dict(inference=True) if kind == "mcat" else {}
    hz, sv, y, att = model(wsi=wsi.to(dev), omics=[o.to(dev) for o in omics], **kw)
    ga = g[f"{case}/A_coattn_sub"]

    def map_rel(a, ref):
        return float(((a - ref).abs() / ref.clamp_min(1e-30)).max())
    e_h = float((hz.detach().cpu() - g[f"{case}/hazards"]).abs().max())
    e_s = float((sv.cpu() - g[f"{case}/survs"]).abs().max())
    e_y = float((y.cpu() - g[f"{case}/Y"]).abs().max())
    e_a = map_rel(sub(att["coattn"]).cpu(), ga)
    e_p = relerr(att["path"], g[f"{case}/A_path"])
    fwd = O.mcat_forward if kind == "mcat" else O.nacagat_forward
    with torch.no_grad():
        _, _, _, att_s = fwd(sd, wsi, omics, bag_storage=torch.bfloat16, **kw)       # same stored values, fp32 arithmetic
        _, _, _, att_x = fwd(sd, wsi.bfloat16().float(), omics, **kw)                # ONLY the patch matrix rounded
    floor_s, floor_x = map_rel(sub(att_s["coattn"]), ga), map_rel(sub(att_x["coattn"]), ga)
    floor_p = relerr(att_s["path"], g[f"{case}/A_path"])
    e_same = map_rel(sub(att["coattn"]).cpu(), sub(att_s["coattn"]))
    print(f"[bf16 vs fp32 reference] {case}: hazards {e_h:.2e} survs {e_s:.2e} Y {e_y:.2e} path map rel {e_p:.2e} (same-storage oracle "
          f"{floor_p:.2e}) | coattn map rel "
          f"{e_a:.2e} (oracle on the same stored values {floor_s:.2e}; patch matrix rounded alone {floor_x:.2e}); vs same-storage oracle {e_same:.2e}")
    for k, v in (("hazards", e_h), ("survs", e_s), ("Y", e_y), ("coattn_rel", e_a), ("coattn_rel_storage_floor", floor_s),
                 ("coattn_rel_patch_rounding_only", floor_x), ("coattn_rel_vs_same_storage_oracle", e_same), ("path_rel", e_p),
                 ("path_rel_storage_floor", floor_p)):
        record_property(f"{case}/{k}", v)
    assert e_h < 1e-3 and e_s < 1e-3 and e_y < 1e-3, (e_h, e_s, e_y)
    assert e_a < 1.25 * floor_s + 2e-4, (e_a, floor_s)
    # Against the same-storage oracle what is left are elements of H_bag that the GPU and the CPU GEMM, summing in
    # different orders, round to different bf16 neighbours (a 2^-8 step of one element of a 256-term logit): a fraction of
    # the storage rounding itself
    assert e_same < 0.6 * floor_s + 2e-4, (e_same, floor_s)
    assert e_p < max(1e-3, 1.25 * floor_p + 2e-4), (e_p, floor_p)
    # fixed ceilings (measured r02/r03: MCAT map 5.2e-3, path 1.5e-4 / 4.1e-4; NaCAGaT map 5.3e-2, path 9.4e-3 / 2.8e-3)
    assert e_a < (6.5e-3 if kind == "mcat" else 6.5e-2), e_a
    assert e_p < (1e-3 if kind == "mcat" else 1.3e-2), e_p


@pytest.mark.parametrize("kind", ["mcat", "nacagat"])
@pytest.mark.parametrize("lengths", [[300, 1, 2048, 77], [30000] + [40] * 15 + [7] * 16], ids=["mixed", "one_giant_31_tiny"])
def test_window_equals_per_slide(dev, kind, lengths):
    """One ragged window launch == the reference's slide-by-slide loop (values and summed grads); the second window is
    the work plan's worst case: one 30 000-patch slide owns almost every workgroup, 31 slides of 40 / 7 patches one each."""
    omic_sizes, seed = [64, 100, 256, 31, 8, 300], 777
    model, _ = build(kind, omic_sizes, seed, dev)
    g = syn.rng(seed)
    wsis = [syn.normal(g, (m, 1024)).to(dev) for m in lengths]
    omics = [[syn.normal(g, (s,)).to(dev) for s in omic_sizes] for _ in lengths]
    labels = (torch.arange(len(lengths)) % 4).to(dev)
    cens = (torch.arange(len(lengths)) % 2).float().to(dev)
    bags = BagBatch.from_list(wsis)
    om_w = [torch.stack([omics[b][i] for b in range(len(lengths))]) for i in range(len(omic_sizes))]
    hz_w, sv_w, y_w, att_w = model.forward_window(bags, om_w, inference=True)
    ces_loss(hz_w, sv_w, labels, cens, reduction="sum").backward()
    grads_w = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    for b in range(len(lengths)):
        kw = dict(inference=True) if kind == "mcat" else {}
        hz, sv, y, att = model(wsi=wsis[b], omics=omics[b], **kw)
        assert relerr(hz_w[b], hz[0]) < 1e-5
        assert relerr(att_w["coattn"][b], att["coattn"]) < 1e-4
        ces_loss(hz, sv, labels[b:b + 1], cens[b:b + 1]).backward()
    for n, p in model.named_parameters():
        scale = max(float(p.grad.abs().max()), 1e-3)            # shift-invariant biases: ~1e-8 noise
        assert float((grads_w[n] - p.grad).abs().max()) / scale < 2e-4, n


@pytest.mark.parametrize("kind", ["mcat", "nacagat"])
def test_full_size_bf16_training_window_equals_per_slide(dev, kind):
    """BASELINE's bag length (15 000 patches, bf16 storage) through the TRAINING-step path of the harness -- the fused
    patch layer + co-attention (MCAT) / patch-layer kernel + K2 (NaCAGaT), token pair, interleaved pooling, head + loss in
    one launch, hand-written dW_H -- against the same slides one at a time through the module's forward() and the separate
    loss launches.  Size-independent property: a window is the sum of its slides (losses equal, gradients add up)."""
    from multimodal_path_omic_amd import harness
    omic_sizes, seed, n, m = [256] * 6, 991, 6, 15000
    model, _ = build(kind, omic_sizes, seed, dev, bag_dtype=torch.bfloat16)
    g = syn.rng(seed)
    wsis = [syn.normal(g, (m, 1024)).to(dev).to(torch.bfloat16) for _ in range(n)]
    omics = [[syn.normal(g, (s,)).to(dev) for s in omic_sizes] for _ in range(n)]
    labels = (torch.arange(n) % 4).to(dev)
    cens = (torch.arange(n) % 2).float().to(dev)
    bags = BagBatch.from_list(wsis)
    om_w = [torch.stack([omics[b][i] for b in range(n)]) for i in range(len(omic_sizes))]
    per_slide, risk = harness.train_window(model, bags, om_w, labels, cens, grad_acc_step=n)
    grads_w = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    for b in range(n):
        hz, sv, y, att = model(wsi=wsis[b], omics=omics[b])
        loss = ces_loss(hz, sv, labels[b:b + 1], cens[b:b + 1])
        assert abs(float(loss) - float(per_slide[b])) < 2e-5 * max(1.0, abs(float(loss))), (b, float(loss), float(per_slide[b]))
        assert abs(float(-sv.sum()) - float(risk[b])) < 1e-5
        (loss / n).backward()
    for k, p in model.named_parameters():
        scale = max(float(p.grad.abs().max()), 1e-3)
        # H.*: the patch layer's pre-activation gradient travels in bf16 and is heavy-tailed (median 4e-9 against a maximum 1e5
        # times larger).  The two runs' fp32 tails differ in the last bit, so 0.1-0.3 % of the gradient's elements round to the
        # other bf16 neighbour (measured, tools/gpu_diag_window_vs_slide.py); when one of the few dominant elements is among
        # them a row of dW_H moves by up to one bf16 ulp of it, 2^-8 = 3.9e-3 (seen: 4.7e-4 with the r02 two-pass K2 gradient,
        # 1.7e-3 with the one-pass kernel -- other elements flipped, same count).  Everything else is fp32 all the way.
        bar = 4e-3 if k.startswith("H.") else 5e-4                # (one bf16 ulp of a dominant element, 2^-8, is the ceiling of that mechanism)
        assert float((grads_w[k] - p.grad).abs().max()) / scale < bar, k


def test_whole_model_at_100k_fp32_patches_matches_oracle(dev):
    """BASELINE config 5 as ONE model: a 100 000 x 1024 fp32 slide through MCAT medium -- the fp32 patch
    layer (fp16-split MFMA products, patch_fc_f32.hip), K1's forward over the fp32 H_bag, the tail, `ces`, K1's vector backward,
    the one-pass fp32 weight gradient -- against the CPU oracle on the same slide: hazards, the co-attention map (sampled
    columns), a sample of dW_H rows and the patch layer's bias gradient.  (The per-kernel tests hold each of these kernels at
    this size alone; the bench runs the chain unchecked.)"""
    sizes, m = [256] * 6, 100_000
    wsi, omics, label, censor = C.model_inputs(m, sizes, 51)
    sd = syn.fill_state_dict(C.model_shapes(sizes, False), 52)
    model = MultimodalCoAttentionTransformer(omic_sizes=sizes)
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    hz, sv, y, att = model(wsi=wsi.to(dev), omics=[o.to(dev) for o in omics], inference=True)
    ces_loss(hz, sv, label.to(dev), censor.to(dev)).backward()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hz_o, sv_o, _, att_o = O.mcat_forward(p, wsi, omics, inference=True)
    O.ces_loss(hz_o, sv_o, label, censor).backward()
    e_h = float((hz.detach().cpu() - hz_o).abs().max())
    cols = torch.randint(0, m, (4096,), generator=torch.Generator().manual_seed(3))
    a, a_o = att["coattn"].detach().cpu()[:, cols], att_o["coattn"].detach()[:, cols]
    e_a = float(((a - a_o).abs() / a_o.abs().clamp_min(1e-30)).max())
    g, g_o = model.H[0].weight.grad.cpu(), p["H.0.weight"].grad
    rows = torch.arange(0, 256, 8)
    e_w = float((g[rows] - g_o[rows]).abs().max() / g_o.abs().max())
    gb, gb_o = model.H[0].bias.grad.cpu(), p["H.0.bias"].grad
    e_b = float((gb - gb_o).abs().max() / gb_o.abs().max())
    print(f"[cfg5 whole model] hazards {e_h:.1e}, map rel {e_a:.1e}, dW_H rows {e_w:.1e}, db_H {e_b:.1e}")
    assert e_h < 1e-4 and e_a < 1e-3 and e_w < 5e-3 and e_b < 5e-3, (e_h, e_a, e_w, e_b)


@pytest.mark.parametrize("kind", ["mcat", "nacagat"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_small_model_size_matches_oracle(dev, kind, dtype):
    """model_size='small' (d = 128, models/mcat/mcat.py:16-17): the E = 128 instantiations of every bag kernel, forward
    and gradients against the oracle (fed the same stored values for the bf16 bag)."""
    omic_sizes, m, seed = [64, 100, 256, 31, 8, 300], 1500, 5150
    cls = MultimodalCoAttentionTransformer if kind == "mcat" else NarrowContextualAttentionGateTransformer
    model = cls(omic_sizes=omic_sizes, model_size="small", bag_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes["H.0.weight"] == (128, 1024)
    sd = syn.fill_state_dict(shapes, seed)
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    wsi, omics, _, _ = C.model_inputs(m, omic_sizes, seed + 1)
    hz, sv, y, att = model(wsi=wsi.to(dev), omics=[o.to(dev) for o in omics])
    label, censor = torch.tensor([1]), torch.tensor([0.0])
    ces_loss(hz, sv, label.to(dev), censor.to(dev)).backward()
    fwd = O.mcat_forward if kind == "mcat" else O.nacagat_forward
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = dict(bag_storage=torch.bfloat16) if dtype == torch.bfloat16 else {}
    hz_o, sv_o, _, _ = fwd(p, wsi, omics, **kw)
    assert float((hz.cpu() - hz_o).abs().max()) < 2e-4
    O.ces_loss(hz_o, sv_o, label, censor).backward()
    for n, prm in model.named_parameters():
        ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
        scale = max(float(ref.abs().max()), 1e-4)
        err = float((prm.grad.cpu() - ref).abs().max()) / scale
        tol = 2e-3 if dtype == torch.float32 else (2e-2 if n.startswith("H.") else 1e-2)
        assert err < tol, (n, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_big_mcat_matches_oracle(dev, dtype):
    """MCAT model_size='big' (d = 512, models/mcat/mcat.py:20-21): the E = 512 instantiations of K1 forward and backward
    (2 / 1 waves per workgroup to fit LDS; functional, not tuned) against the oracle."""
    omic_sizes, m, seed = [64, 100, 256, 31, 8, 300], 1200, 6160
    model = MultimodalCoAttentionTransformer(omic_sizes=omic_sizes, model_size="big", bag_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes["H.0.weight"] == (512, 1024)
    sd = syn.fill_state_dict(shapes, seed)
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    wsi, omics, _, _ = C.model_inputs(m, omic_sizes, seed + 1)
    hz, sv, y, att = model(wsi=wsi.to(dev), omics=[o.to(dev) for o in omics], inference=True)
    label, censor = torch.tensor([3]), torch.tensor([0.0])
    ces_loss(hz, sv, label.to(dev), censor.to(dev)).backward()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = dict(bag_storage=torch.bfloat16) if dtype == torch.bfloat16 else {}   # (every width runs the one patch-layer kernel: H_bag rounded once)
    hz_o, sv_o, _, att_o = O.mcat_forward(p, wsi, omics, inference=True, **kw)
    assert float((hz.cpu() - hz_o).abs().max()) < 2e-4
    a, a_o = att["coattn"].cpu(), att_o["coattn"].detach()
    assert ((a - a_o).abs() / a_o.clamp_min(1e-30)).max().item() < 1e-3
    O.ces_loss(hz_o, sv_o, label, censor).backward()
    for n, prm in model.named_parameters():
        ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
        scale = max(float(ref.abs().max()), 1e-4)
        err = float((prm.grad.cpu() - ref).abs().max()) / scale
        # H.*: d(pre-activation) passes through ReLU's kink -- of the 614 400 pre-activations a few lie within the 6e-6
        # forward difference between the GPU and CPU fp32 GEMMs and flip their mask; one flipped element moves a row of
        # dW_H by |dH| |x| ~ 3e-5, i.e. 6e-3 of this fixture's tiny gradient scale (tools/gpu_diag_big4.py: the gradient
        # ARRIVING at H_bag agrees to 4e-6)
        tol = (1e-2 if n.startswith("H.") else 2e-3) if dtype == torch.float32 else (2e-2 if n.startswith("H.") else 1e-2)
        assert err < tol, (n, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_big_nacagat_matches_oracle(dev, dtype):
    """NaCAGaT model_size='big' (d = 512, models/nacagat/nacagat.py:17-18): K2's bag passes run once per column half of the
    split-halves bag layout on the 256-wide kernels (csrc/capi.hip; functional, not tuned) -- forward, map and every parameter
    gradient against the oracle, on a ragged two-slide window so that the halves' strided copies see several query rows."""
    omic_sizes, seed = [64, 100, 256, 31, 8, 300], 6262
    lengths = [1200, 77]
    model = NarrowContextualAttentionGateTransformer(omic_sizes=omic_sizes, model_size="big", bag_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes["H.0.weight"] == (512, 1024) and shapes["co_attention.in_proj_weight"] == (1536, 512)
    sd = syn.fill_state_dict(shapes, seed)
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    g = syn.rng(seed + 1)
    wsis = [syn.normal(g, (m, 1024)) for m in lengths]
    omics = [[syn.normal(g, (s,)) for s in omic_sizes] for _ in lengths]
    labels, cens = torch.tensor([3, 1]), torch.tensor([0.0, 1.0])
    bags = BagBatch.from_list([w.to(dev).to(dtype) for w in wsis])
    om_w = [torch.stack([omics[b][i] for b in range(len(lengths))]).to(dev) for i in range(len(omic_sizes))]
    hz_w, sv_w, _, att_w = model.forward_window(bags, om_w, inference=True)
    ces_loss(hz_w, sv_w, labels.to(dev), cens.to(dev), reduction="sum").backward()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw = dict(bag_storage=torch.bfloat16) if dtype == torch.bfloat16 else {}   # (every width runs the one patch-layer kernel: H_bag rounded once)
    for b, m in enumerate(lengths):
        hz_o, sv_o, _, att_o = O.nacagat_forward(p, wsis[b], omics[b], **kw)
        assert float((hz_w[b].cpu() - hz_o[0]).abs().max()) < 2e-4
        a, a_o = att_w["coattn"][b].cpu(), att_o["coattn"].detach()
        rel = ((a - a_o).abs() / a_o.clamp_min(1e-30)).max().item()
        print(f"[big nacagat] {dtype} slide {b} ({m} rows): co-attention map rel err {rel:.2e}")
        # bf16: H_bag elements whose fp32 value sits near a rounding boundary land one bf16 step (2^-8 of the element) apart
        # under the kernel's and the oracle's summation orders; K = H W_k^T carries each into the exponent (measured 3.5e-3
        # at d = 512, printed above; the fp32 leg of this test holds the same kernels to 1e-3)
        assert rel < (1e-3 if dtype == torch.float32 else 6e-3), rel
        O.ces_loss(hz_o, sv_o, labels[b:b + 1], cens[b:b + 1]).backward()
    for n, prm in model.named_parameters():
        ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
        scale = max(float(ref.abs().max()), 1e-4)
        err = float((prm.grad.cpu() - ref).abs().max()) / scale
        tol = (1e-2 if n.startswith("H.") else 3e-3) if dtype == torch.float32 else (2e-2 if n.startswith("H.") else 1e-2)
        assert err < tol, (n, err)


@pytest.mark.parametrize("kind", ["mcat", "nacagat"])
def test_cesar_window_loss_matches_oracle(dev, kind):
    """harness.train_window(loss='cesar'): ces + lambda ||A_b||_2 per slide (models/loss.py:88-101) over a ragged window,
    with the norm's gradient entering the co-attention kernels through the flat map; against the oracle slide by slide."""
    from multimodal_path_omic_amd import harness
    from multimodal_path_omic_amd.dp import FlatGradBucket
    omic_sizes, seed, lam = [64, 100, 256, 31, 8, 300], 888, 0.05
    lengths = [300, 33, 1500, 77]
    model, sd = build(kind, omic_sizes, seed, dev)
    g = syn.rng(seed)
    wsis = [syn.normal(g, (m, 1024)) for m in lengths]
    omics = [[syn.normal(g, (s,)) for s in omic_sizes] for _ in lengths]
    labels, cens = torch.tensor([0, 1, 2, 3]), torch.tensor([0., 1., 0., 1.])
    bags = BagBatch.from_list([w.to(dev) for w in wsis])
    om_w = [torch.stack([omics[b][i] for b in range(len(lengths))]).to(dev) for i in range(len(omic_sizes))]
    bucket = FlatGradBucket(list(model.parameters()))
    bucket.begin()
    per_slide, risk = harness.train_window(model, bags, om_w, labels.to(dev), cens.to(dev), 4, loss="cesar", lambda_reg=lam)
    bucket.finish()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fwd = O.mcat_forward if kind == "mcat" else O.nacagat_forward
    for b in range(len(lengths)):
        kw = dict(inference=True) if kind == "mcat" else {}
        hz, sv, _, att = fwd(p, wsis[b], omics[b], **kw)
        loss_b, _ = O.cesar_loss(hz, sv, labels[b:b + 1], cens[b:b + 1], att["coattn"], lambda_reg=lam)
        assert per_slide[b].item() == pytest.approx(loss_b.item(), rel=2e-4, abs=2e-5)
        (loss_b / 4).backward()
    for n, prm in model.named_parameters():
        ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
        scale = max(float(ref.abs().max()), 1e-4)
        err = float((prm.grad.cpu() - ref).abs().max()) / scale
        assert err < (1e-2 if n.startswith("H.") else 3e-3), (n, err)


@pytest.mark.parametrize("kind", ["mcat", "nacagat"])
@pytest.mark.parametrize("n_groups", [1, 3, 7, 8, 9, 15, 16])
def test_other_omic_group_counts_and_tiny_bags(dev, kind, n_groups):
    """The number of omic queries is a model argument (len(omic_sizes), models/mcat/mcat.py:32-45), not a constant 6: 1, 3, 9
    and the MFMA-column maximum 16 groups, over a ragged window whose bags straddle every tile boundary (1, 2, 31, 32, 33,
    65 patches) -- window forward == per-slide oracle, gradients included."""
    omic_sizes = [16 + 8 * i for i in range(n_groups)]
    lengths = [1, 2, 31, 32, 33, 65]
    seed = 1000 + n_groups
    cls = MultimodalCoAttentionTransformer if kind == "mcat" else NarrowContextualAttentionGateTransformer
    model = cls(omic_sizes=omic_sizes)
    sd = syn.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    g = syn.rng(seed)
    wsis = [syn.normal(g, (m, 1024)) for m in lengths]
    omics = [[syn.normal(g, (s,)) for s in omic_sizes] for _ in lengths]
    labels, cens = torch.arange(len(lengths)) % 4, (torch.arange(len(lengths)) % 2).float()
    bags = BagBatch.from_list([w.to(dev) for w in wsis])
    om_w = [torch.stack([omics[b][i] for b in range(len(lengths))]).to(dev) for i in range(n_groups)]
    hz_w, sv_w, _, att_w = model.forward_window(bags, om_w, inference=True)
    ces_loss(hz_w, sv_w, labels.to(dev), cens.to(dev), reduction="sum").backward()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fwd = O.mcat_forward if kind == "mcat" else O.nacagat_forward
    for b, m in enumerate(lengths):
        kw = dict(inference=True) if kind == "mcat" else {}
        hz, sv, _, att = fwd(p, wsis[b], omics[b], **kw)
        assert float((hz_w[b].cpu() - hz[0]).abs().max()) < 1e-4, (b, m)
        a, a_o = att_w["coattn"][b].cpu(), att["coattn"].detach()
        assert a.shape == (n_groups, m)
        assert ((a - a_o).abs() / a_o.clamp_min(1e-30)).max().item() < 1e-3, (b, m)
        O.ces_loss(hz, sv, labels[b:b + 1], cens[b:b + 1]).backward()
    worst = (0.0, "")
    for n, prm in model.named_parameters():
        ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
        scale = max(float(ref.abs().max()), 1e-4)
        err = float((prm.grad.cpu() - ref).abs().max()) / scale
        worst = max(worst, (err, n))
        # bar 1e-2: with up to 16 tokens x 512 FFN units x 2 layers x 2 encoders x 6 slides (~2e5 ReLU pre-activations) one
        # that lies within rounding of zero can flip its mask against the CPU oracle and move a row of that layer's
        # weight gradient (seen once: nacagat, 16 groups, 7.9e-3 at path_transformer.layers.0.linear1.weight, while the
        # co-attention module alone agrees to 1.5e-5 at 16 queries, tools/gpu_diag_nq16.py); everything else is < 1e-3
        assert err < 1e-2, (n, err)
    print(f"worst gradient error [{kind}, {n_groups} groups]: {worst[0]:.2e} at {worst[1]}")


def test_more_than_16_omic_groups_fails_loudly(dev):
    """The omic queries are the 16 columns of one MFMA block: 17 groups is refused with a message, never mis-computed."""
    omic_sizes = [8] * 17
    model = MultimodalCoAttentionTransformer(omic_sizes=omic_sizes).to(dev).eval()
    wsi = torch.randn(64, 1024, device=dev)
    with pytest.raises(RuntimeError, match="omic queries must be in 1..16"):
        model(wsi=wsi, omics=[torch.randn(8, device=dev) for _ in omic_sizes])
