"""Row f1: the patch layer fused with MCAT's co-attention forward (mpo_patch_coattn_mcat_forward, one pass over the raw
patch matrix) against
  * the CPU oracle fed the same stored values (bf16 patch matrix / weight operand / H_bag, fp32 arithmetic) -- the parity
    check proper, incl. the reference's golden co-attention cases re-used as H-bag consumers;
  * the unfused HIP path (library GEMM + epilogue kernel + K1), which it must reproduce up to the ONE rounding it removes
    (the unfused path rounds the GEMM output to bf16 before the bias);
  * itself on ragged windows (window == per-slide; block / tile / slide edges: 1, 31, 127, 128, 129, 257 ... rows);
  * dropout: realised rate, keep scale, determinism per (seed, offset), fresh masks per epoch, backward mask consistency.
Models reach this kernel through MultimodalCoAttentionTransformer with a bf16 bag (tests/test_gpu_models.py,
test_gpu_graph.py, test_gpu_cohort.py, test_gpu_dp.py)."""
import math

import pytest
import torch

import cases as C
from multimodal_path_omic_amd import ops
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.ops import BagBatch
from oracle import mpo_oracle as O

pytestmark = pytest.mark.gpu
E, N_Q = 256, 6


def _params(seed, gain=1.0):
    shapes = {"H.0.weight": (E, 1024), "H.0.bias": (E,), **C.MCAT_COATTN_SHAPES}
    return syn.fill_state_dict(shapes, seed, gain)


def _inputs(lengths, seed):
    g = syn.rng(seed)
    bags = [syn.normal(g, (m, 1024)) for m in lengths]
    query = syn.normal(g, (len(lengths) * N_Q, E))
    return bags, query


def _fused(p, bags, query, dev, need_weights=True, drop_p=0.0, n_q=N_Q):
    batch = BagBatch.from_list([b.to(dev).to(torch.bfloat16) for b in bags])
    d = {k: v.to(dev).requires_grad_(True) for k, v in p.items()}
    q = query.to(dev).requires_grad_(True)
    out, amap, h, _ = ops.patch_coattn_mcat(batch.data, batch, d["H.0.weight"], d["H.0.bias"], drop_p, q,
                                            d["co_attention.in_proj_weight"], d["co_attention.in_proj_bias"],
                                            d["co_attention.out_proj.weight"], d["co_attention.out_proj.bias"], need_weights)
    return out, amap, h, d, q, batch


def _unfused(p, bags, query, dev, need_weights=True):
    batch = BagBatch.from_list([b.to(dev).to(torch.bfloat16) for b in bags])
    d = {k: v.to(dev).requires_grad_(True) for k, v in p.items()}
    q = query.to(dev).requires_grad_(True)
    h = ops.patch_fc(batch.data, d["H.0.weight"], d["H.0.bias"], 0.0, pre_gated_grad=True)
    out, amap = ops.coattn_mcat(q, batch.with_data(h), d["co_attention.in_proj_weight"], d["co_attention.in_proj_bias"],
                                d["co_attention.out_proj.weight"], d["co_attention.out_proj.bias"], need_weights, 1.0)
    return out, amap, h, d, q, batch


def _oracle(p, bag, query_rows):
    """One slide through the oracle on the stored values the fused kernel sees: bf16 X and W_H operands, fp32 accumulate,
    bias + ReLU in fp32, ONE rounding of H_bag to bf16; co-attention in fp32 on that H_bag."""
    x = bag.bfloat16().float()
    w = p["H.0.weight"].bfloat16().float()
    h = torch.relu(x @ w.t() + p["H.0.bias"]).bfloat16().float()
    out, a = O.mcat_coattention(query_rows, h, p)
    return out, a, h


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("lengths,gain", [([256], 1.0), ([2000], 1.0), ([777], 2.0), ([15000], 1.0), ([2000], 4.0)],
                         ids=["m256", "m2000", "m777_ragged", "m15000", "m2000_peaky"])
def test_fused_forward_and_gradients_match_oracle(dev, lengths, gain):
    p = _params(811, gain)
    bags, query = _inputs(lengths, 812)
    out, amap, h, d, q, batch = _fused(p, bags, query, dev)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    qr = query.clone().requires_grad_(True)
    out_o, a_o, h_o = _oracle(pr, bags[0], qr)
    # H_bag: the kernel accumulates in another order than the CPU GEMM: an element can land on the other side of a bf16
    # rounding boundary (one ulp = 2^-8 relative), nothing more
    hd = (h.float().cpu() - h_o.detach()).abs()
    assert float((hd / h_o.detach().abs().clamp_min(1e-2)).max()) <= 2.0 ** -7
    assert float((hd > 0).float().mean()) < 0.02
    # The co-attention itself is held to the oracle ON THE KERNEL'S OWN H_bag (the stored values it consumed): the handful
    # of one-ulp differences above would otherwise show up in the map through |qk| (~1e-3 of a logit at gain 1, ~2e-2 at
    # gain 4), which says nothing about the attention arithmetic.
    out_k, a_k = O.mcat_coattention(query, h.float().cpu(), p)
    assert relmax(out.detach().cpu(), out_k) < 1e-3
    a = amap.view(N_Q, lengths[0]).detach().cpu()
    rel = float(((a - a_k).abs() / a_k.clamp_min(1e-30)).max())
    print(f"[f1 map] {lengths[0]} rows, gain {gain}: rel err {rel:.2e} vs the oracle on the kernel's H_bag")
    assert rel < 1e-3, rel
    # The COMPOSED check, end to end against the same-storage oracle (its own H_bag): the only admissible extra error is what
    # the one-ulp H_bag differences bounded above can do to a logit.  With s[n][m] = qk[n] . H[m] (qk = (q W_q^T + b_q) W_k /
    # sqrt(E), the folded query) a perturbation dH moves logit (n, m) by at most sum_d |qk[n][d]| |dH[m][d]| =: ds[n][m], and a
    # softmax entry by at most exp(2 max ds) - 1 relative (its own logit up, the normaliser down); on top of that the
    # arithmetic bar of the check above.
    w_in, b_in = p["co_attention.in_proj_weight"], p["co_attention.in_proj_bias"]
    qk = ((query @ w_in[:E].t() + b_in[:E]) / math.sqrt(E)) @ w_in[E:2 * E]
    ds_max = float((qk.abs() @ hd.t()).max())
    rel_e2e = float(((a - a_o.detach()).abs() / a_o.detach().clamp_min(1e-30)).max())
    bound = math.expm1(2.0 * ds_max) + 1e-3
    print(f"[f1 map, composed] {lengths[0]} rows, gain {gain}: rel err {rel_e2e:.2e} vs the same-storage oracle end to end; "
          f"bound from the H_bag one-ulp differences {bound:.2e} (max logit shift {ds_max:.2e})")
    assert rel_e2e < bound, (rel_e2e, bound)
    assert relmax(out.detach().cpu(), out_o.detach()) < 2e-3 * gain * gain
    probe_o, probe_a = syn.normal(syn.rng(5), (N_Q, E)), syn.normal(syn.rng(6), (N_Q, lengths[0]))
    ((out * probe_o.to(dev)).sum() + (amap.view(N_Q, -1) * probe_a.to(dev)).sum()).backward()
    ((out_o * probe_o).sum() + (a_o * probe_a).sum()).backward()
    for k in p:
        ref = pr[k].grad
        tol = 2e-2 if k.startswith("H.") else 5e-3            # dH leaves K1's backward in bf16 on its way into dW_H
        if k == "co_attention.in_proj_bias":
            ref = ref.clone()
            ref[E:2 * E] = 0                                    # the key bias cancels in the softmax (exactly zero here)
        assert relmax(d[k].grad.cpu(), ref) < tol, (k, relmax(d[k].grad.cpu(), ref))
    assert relmax(q.grad.cpu(), qr.grad) < 5e-3


@pytest.mark.parametrize("lengths", [[1], [31, 32, 33], [127, 128, 129, 257, 1], [300, 1, 2048, 77], [5000] + [40] * 15 + [7] * 16],
                         ids=["one_row", "tile_edges", "block_edges", "mixed", "one_giant_31_tiny"])
def test_fused_equals_unfused_on_ragged_windows(dev, lengths):
    p = _params(821)
    bags, query = _inputs(lengths, 822)
    out_f, map_f, h_f, d_f, q_f, batch = _fused(p, bags, query, dev)
    out_u, map_u, h_u, d_u, q_u, _ = _unfused(p, bags, query, dev)
    # the unfused path rounds the GEMM output to bf16 BEFORE the bias (two roundings), the fused one once
    # (its pre-bias rounding moves an element by up to 2^-9 |x W^T|, i.e. < 0.02 absolute at these magnitudes, and can
    # push an element across the ReLU kink)
    hd = (h_f.float() - h_u.float()).abs().detach()
    assert float((hd - 2.0 ** -7 * h_u.float().abs().detach()).max()) < 0.01 and float(hd.mean()) < 1e-3
    assert relmax(out_f.detach(), out_u.detach()) < 5e-3
    assert float(((map_f - map_u).abs() / map_u.clamp_min(1e-30)).max()) < 2e-2
    # every row of every slide sums to one; slide b's block sits at n_q * cu[b]
    off = 0
    for m in lengths:
        blk = map_f[N_Q * off:N_Q * (off + m)].view(N_Q, m)
        torch.testing.assert_close(blk.sum(1), torch.ones(N_Q, device=dev), rtol=1e-4, atol=1e-4)
        off += m
    probe = syn.normal(syn.rng(7), tuple(out_f.shape)).to(dev)
    (out_f * probe).sum().backward()
    # window == slide by slide (same kernel, one-slide plans): H_bag bit for bit, outputs, and the SUMMED gradients (the
    # unfused path is no reference for gradients: its extra rounding moves elements across the ReLU kink)
    off, gsum = 0, {k: torch.zeros_like(v) for k, v in d_f.items()}
    for b, m in enumerate(lengths):
        o1, m1, h1, d1, *_ = _fused(p, [bags[b]], query[N_Q * b:N_Q * (b + 1)], dev)
        assert relmax(o1.detach(), out_f.detach()[N_Q * b:N_Q * (b + 1)]) < 1e-4
        assert torch.equal(h1, h_f[off:off + m])
        (o1 * probe[N_Q * b:N_Q * (b + 1)]).sum().backward()
        for k in p:
            gsum[k] += d1[k].grad
        off += m
    for k in p:
        assert relmax(d_f[k].grad, gsum[k]) < 2e-3, (k, relmax(d_f[k].grad, gsum[k]))


def test_fused_fewer_and_more_queries(dev):
    """1 and 8 omic queries (the kernel's range); 9 and more fall back to the unfused path in the model."""
    p = _params(831)
    for n_q in (1, 8):
        g = syn.rng(832 + n_q)
        bag, query = syn.normal(g, (700, 1024)), syn.normal(g, (n_q, E))
        out, amap, h, *_ = _fused(p, [bag], query, dev, n_q=n_q)
        out_o, a_o, _ = _oracle(p, bag, query)
        assert relmax(out.detach().cpu(), out_o) < 2e-3
        assert float(((amap.view(n_q, 700).cpu() - a_o).abs() / a_o.clamp_min(1e-30)).max()) < 3e-3
    assert not ops.fused_patch_coattn_supported(torch.empty(4, 1024, dtype=torch.bfloat16), 256, 9)
    assert not ops.fused_patch_coattn_supported(torch.empty(4, 1024, dtype=torch.float32), 256, 6)


def test_fused_dropout_masks(dev):
    p = _params(841)
    bags, query = _inputs([3000, 500], 842)
    torch.manual_seed(1234)
    ops.set_rng_epoch(None)
    _, _, h0, *_ = _fused(p, bags, query, dev, need_weights=False, drop_p=0.0)
    torch.manual_seed(1234)
    calls = ops._rng_calls
    out1, _, h1, d1, q1, _ = _fused(p, bags, query, dev, need_weights=False, drop_p=0.25)
    ops._rng_calls = calls                                       # same (seed, offset) -> the same mask
    _, _, h2, *_ = _fused(p, bags, query, dev, need_weights=False, drop_p=0.25)
    assert torch.equal(h1, h2)
    _, _, h3, *_ = _fused(p, bags, query, dev, need_weights=False, drop_p=0.25)      # next offset -> another mask
    assert not torch.equal(h1, h3)
    pos = h0 > 0
    dropped = pos & (h1 == 0)
    rate = float(dropped.sum()) / float(pos.sum())
    assert abs(rate - 0.25) < 0.005, rate                        # ~4e5 positives: sigma ~ 7e-4
    kept = pos & (h1 != 0)
    ratio = (h1[kept].float() / h0[kept].float())
    assert float((ratio - 4.0 / 3.0).abs().max()) < 0.02         # 1/(1-p) up to the bf16 rounding of both sides
    # rows and columns are dropped independently (no stripe of a shared draw): per-column and per-row rates
    col_rate = dropped.float().sum(0) / pos.float().sum(0).clamp_min(1)
    assert float((col_rate - 0.25).abs().max()) < 0.06
    # the epoch moves every stream
    ep = torch.zeros(1, dtype=torch.int64, device=dev)
    ops.set_rng_epoch(ep)
    ops._rng_calls = calls
    _, _, h4, *_ = _fused(p, bags, query, dev, need_weights=False, drop_p=0.25)
    assert torch.equal(h4, h1)                                   # epoch 0 == no epoch
    ep += 1
    ops._rng_calls = calls
    _, _, h5, *_ = _fused(p, bags, query, dev, need_weights=False, drop_p=0.25)
    assert not torch.equal(h5, h1)
    ops.set_rng_epoch(None)
    # backward sees the same mask: d(sum out)/dW_H must equal the oracle's gradient with THIS mask replayed
    out1.sum().backward()
    keep = (h1 != 0).float().cpu() * (4.0 / 3.0)
    keep[(h0 <= 0).cpu()] = 4.0 / 3.0                            # where relu is 0 the mask value is irrelevant
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    outs, off = [], 0
    for b, bag in enumerate(bags):
        m = bag.shape[0]
        x = bag.bfloat16().float()
        # (gradient flows to the fp32 master weight through the rounded operand as identity, like the kernel's)
        h = torch.relu(x @ (pr["H.0.weight"] + (pr["H.0.weight"].bfloat16().float() - pr["H.0.weight"]).detach()).t()
                       + pr["H.0.bias"]) * keep[off:off + m]
        o, _ = O.mcat_coattention(query[N_Q * b:N_Q * (b + 1)], h, pr, need_weights=False)
        outs.append(o)
        off += m
    torch.cat(outs).sum().backward()
    assert relmax(d1["H.0.weight"].grad.cpu(), pr["H.0.weight"].grad) < 2e-2
    assert relmax(d1["H.0.bias"].grad.cpu(), pr["H.0.bias"].grad) < 2e-2


class _ProduceInto(torch.autograd.Function):
    """Test stand-in for the omic SNN: a producer that writes its output into the pair's second half."""

    @staticmethod
    def forward(ctx, q, pair):
        dst = pair.slot(1, q.shape)
        dst.copy_(q)
        return dst

    @staticmethod
    def backward(ctx, g):
        return g, None


def test_token_pair_hands_both_token_sets_over_without_copies(dev):
    """ops.TokenPair: the co-attention output and the (doubly used) query land in one (2, R, E) buffer and the query's
    second gradient is folded into the op's own backward (d_query accumulate): values and gradients must equal the
    torch.stack formulation."""
    lengths = [700, 333]
    p = _params(31)
    bags, query = _inputs(lengths, 77)
    probe = syn.normal(syn.rng(5), (2, len(lengths) * N_Q, E)).to(dev)

    def run(use_pair):
        batch = BagBatch.from_list([b.to(dev).to(torch.bfloat16) for b in bags])
        d = {k: v.to(dev).requires_grad_(True) for k, v in p.items()}
        q = query.to(dev).requires_grad_(True)
        pair = ops.TokenPair(q.shape[0], E, dev) if use_pair else None
        q_in = _ProduceInto.apply(q, pair) if use_pair else q * 1.0
        out, _, _, q_tok = ops.patch_coattn_mcat(batch.data, batch, d["H.0.weight"], d["H.0.bias"], 0.0, q_in,
                                                 d["co_attention.in_proj_weight"], d["co_attention.in_proj_bias"],
                                                 d["co_attention.out_proj.weight"], d["co_attention.out_proj.bias"], False, pair)
        stacked = pair.stack(out, q_tok) if use_pair else torch.stack([out, q_tok])
        (stacked * probe).sum().backward()
        return stacked.detach().clone(), q.grad.clone(), {k: v.grad.clone() for k, v in d.items()}

    s0, gq0, g0 = run(False)
    s1, gq1, g1 = run(True)
    assert torch.equal(s0, s1)
    torch.testing.assert_close(gq1, gq0, rtol=1e-5, atol=1e-6)
    for k in g0:
        torch.testing.assert_close(g1[k], g0[k], rtol=1e-5, atol=1e-6)


def test_patch_layer_alone_is_the_fused_kernel_without_its_co_attention(dev):
    """mpo_patch_fc_forward (the fused kernel with its co-attention slices off): H_bag must be bit for bit what the fused
    forward writes -- eval and, with the same dropout stream, training mode -- on a ragged window with partial blocks; the
    autograd wrapper (ops.patch_fc with a batch) must give the patch layer's gradients."""
    lengths = [1, 127, 129, 700, 2049]
    p = _params(41)
    bags, query = _inputs(lengths, 42)
    batch = BagBatch.from_list([b.to(dev).to(torch.bfloat16) for b in bags])
    w, bias = p["H.0.weight"].to(dev).requires_grad_(True), p["H.0.bias"].to(dev).requires_grad_(True)
    for drop_p in (0.0, 0.25):
        ops._rng_calls = 500
        _, _, h_fused, _ = ops.patch_coattn_mcat(batch.data, batch, w, bias, drop_p, query.to(dev),
                                                 p["co_attention.in_proj_weight"].to(dev), p["co_attention.in_proj_bias"].to(dev),
                                                 p["co_attention.out_proj.weight"].to(dev), p["co_attention.out_proj.bias"].to(dev), False)
        ops._rng_calls = 500
        h = ops.patch_fc(batch.data, w, bias, drop_p, batch=batch)
        assert torch.equal(h, h_fused)
        if drop_p > 0:
            assert abs(h._mpo_keep_scale - 1.0 / 0.75) < 1e-12
    # eval-mode values against the oracle on the same stored operands, gradients against autograd on those
    x = batch.data.float()
    wr, br = p["H.0.weight"].to(dev).bfloat16().float().requires_grad_(True), p["H.0.bias"].to(dev).clone().requires_grad_(True)
    ref = torch.relu(x @ wr.t() + br)
    h = ops.patch_fc(batch.data, w, bias, 0.0, batch=batch)
    assert relmax(h.float(), ref.detach()) < 2.0 ** -7
    probe = torch.randn_like(ref) * 0.01
    gw, gb = torch.autograd.grad((h.float() * probe).sum(), [w, bias])
    (ref * probe.bfloat16().float()).sum().backward()
    assert relmax(gw, wr.grad) < 5e-3 and relmax(gb, br.grad) < 5e-3


@pytest.mark.parametrize("embed", [128, 512])
def test_patch_layer_widths_of_the_small_and_big_models(dev, embed):
    """model_size 'small' / 'big' (models/mcat/mcat.py:16-21: Linear(1024, 128 | 512)) on the headline kernel: 128 = the upper
    half of the 256-column block is zero weight rows whose stores are dropped, 512 = one pass per column half.  Values against
    the fp32 product of the stored operands (H_bag is bf16: 2^-7 of the largest entry), the realised dropout rate and scale,
    window == slide by slide bit for bit, gradients against autograd, and nothing written outside H_bag."""
    lengths = [1, 127, 129, 700, 2049]
    gen = torch.Generator().manual_seed(embed)
    bags = [torch.randn(m, 1024, generator=gen) for m in lengths]
    batch = BagBatch.from_list([b.to(dev).to(torch.bfloat16) for b in bags])
    w = (torch.randn(embed, 1024, generator=gen) / 32).to(dev).requires_grad_(True)
    bias = (torch.randn(embed, generator=gen) * 0.1).to(dev).requires_grad_(True)
    x = batch.data.float()
    wr, br = w.detach().bfloat16().float().requires_grad_(True), bias.detach().clone().requires_grad_(True)
    ref = torch.relu(x @ wr.t() + br)
    h = ops.patch_fc(batch.data, w, bias, 0.0, batch=batch)
    assert h.shape == (sum(lengths), embed) and h.dtype == torch.bfloat16
    assert relmax(h.float(), ref.detach()) < 2.0 ** -7
    for b, m in enumerate(lengths):                                     # the window's rows == the slide on its own
        r0 = sum(lengths[:b])
        alone = ops.patch_fc(batch.data[r0:r0 + m].contiguous(), w, bias, 0.0)
        assert torch.equal(alone, h[r0:r0 + m])
    probe = torch.randn_like(ref) * 0.01
    gw, gb = torch.autograd.grad((h.float() * probe).sum(), [w, bias])
    (ref * probe.bfloat16().float()).sum().backward()
    assert relmax(gw, wr.grad) < 5e-3 and relmax(gb, br.grad) < 5e-3
    # training mode: zeros where dropped, survivors scaled by 1 / (1 - realised rate); same stream -> same mask
    ops._rng_calls = 900
    hd = ops.patch_fc(batch.data, w, bias, 0.25, batch=batch)
    ops._rng_calls = 900
    hd2 = ops.patch_fc(batch.data, w, bias, 0.25, batch=batch)
    assert torch.equal(hd, hd2)
    pos = ref.detach() > 0.05
    kept = (hd != 0) & pos
    rate = 1.0 - kept.sum().item() / pos.sum().item()
    assert abs(rate - 0.25) < 0.01, rate
    assert abs(hd._mpo_keep_scale - 1.0 / 0.75) < 1e-12
    assert relmax(hd.float()[kept], (ref.detach() / 0.75)[kept]) < 2.0 ** -7
    # per-column and per-row keep rates: the counter covers every (row, 16-column group) of the wider / narrower row once
    col_rate = 1.0 - (kept.sum(0).float() / pos.sum(0).clamp_min(1).float())
    assert float((col_rate - 0.25).abs().max()) < 0.06, float((col_rate - 0.25).abs().max())


def test_patch_layer_refuses_other_widths(dev):
    x = torch.zeros(100, 1024, device=dev, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="patch layer"):
        ops.patch_fc(x, torch.zeros(384, 1024, device=dev), torch.zeros(384, device=dev), 0.0)
