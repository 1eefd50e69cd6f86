"""The patch layer of an fp32-stored window on the hand-written kernels (mpo_patch_fc_f32_*, csrc/patch_fc_f32.hip;
models/mcat/mcat.py:24-29,87 and its backward): forward against the fp64 product of the same fp32 operands, the one-pass
backward (ReLU / dropout derivative read off H_bag, dW_H = g^T X, db_H = colsum g) against fp64 as well.  The products run
as three half-precision MFMA terms of hi / lo operand splits -- fp16 splits forward (the ReLU mask must not flip against the
fp32 reference more often than an fp32 GEMM's would), bf16 splits in the weight gradient: the bars below are what that
arithmetic is held to (a plain bf16 product would sit at ~4e-3, bf16 splits at ~5e-6 on the forward)."""
import pytest
import torch

from multimodal_path_omic_amd import ops

pytestmark = pytest.mark.gpu


def _operands(rows, dev, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(rows, 1024, device=dev, generator=gen)
    w = (torch.rand(256, 1024, device=dev, generator=gen) - 0.5) / 16
    b = torch.randn(256, device=dev, generator=gen) * 0.1
    return x, w, b, gen


@pytest.mark.parametrize("rows", [1, 15, 127, 128, 129, 1000, 33001])
def test_forward_and_backward_match_fp64(dev, rows):
    x, w, b, gen = _operands(rows, dev, rows)
    wp, bp = torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())
    # operands carved out of NaN-padded buffers: nothing outside [0, rows) may leak into a sum
    xbuf = torch.full((rows + 200, 1024), float("nan"), device=dev)
    xbuf[:rows] = x
    h = ops.patch_fc_f32(xbuf[:rows], wp, bp, 0.0)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    err = float((h.double() - ref).abs().max())
    assert err < 3e-6, err                                     # max over up to 8M elements; |x w| ~ 0.03, 1024 terms, ~2^-22 per product: an fp32 GEMM's own level
    probe = torch.randn(rows, 256, device=dev, generator=gen)
    (h * probe).sum().backward()
    g = probe.double() * (ref > 0)
    # (an element within the forward error of the ReLU kink may flip its mask against the fp64 reference: compare dW with
    #  the mask the kernel itself used)
    g_k = probe.double() * (h.detach() > 0)
    dw_ref, db_ref = g_k.t() @ x.double(), g_k.sum(0)
    assert float((h.detach() > 0).ne(ref > 0).float().mean()) < 5e-6      # mask flips against fp64: pre-activations within ~1e-6 of zero
    e_w = float((wp.grad.double() - dw_ref).abs().max() / dw_ref.abs().max().clamp_min(1e-30))
    e_b = float((bp.grad.double() - db_ref).abs().max() / db_ref.abs().max().clamp_min(1e-30))
    assert e_w < 1e-4 and e_b < 1e-5, (e_w, e_b)
    # the two masks differ on the handful of flipped elements only
    assert float(((g - g_k) != 0).float().mean()) < 5e-6


@pytest.mark.parametrize("x_gain,w_gain", [(1e-4, 1.0), (1e4, 1.0), (1.0, 3000.0), (1e-4, 3000.0), (3e5, 1e-3)],
                         ids=["features_1e-4", "features_1e4", "weights_100", "tiny_features_large_weights", "features_3e5"])
def test_forward_keeps_fp32_accuracy_at_any_operand_scale(dev, x_gain, w_gain):
    """the fp16 operand splits must not depend on the operands' own scale -- features of 1e-4 (fp16 subnormals
    when unscaled), of 1e4 and 3e5 (past fp16's maximum when unscaled), weights of ~100 (W x 1024 overflowed): each operand is
    scaled by the power of two that puts its maximum at 2^14..2^15, so the relative error against fp64 is the same everywhere."""
    rows = 3000
    x, w, b, gen = _operands(rows, dev, 77)
    x, w = x * x_gain, w * w_gain
    b = b * (x_gain * w_gain)
    h = ops.patch_fc_f32(x, torch.nn.Parameter(w), torch.nn.Parameter(b), 0.0)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    err = float((h.double() - ref).abs().max() / ref.abs().max())
    assert err < 3e-6, err                                     # (N(0,1) x U(+-1/32) operands: 3e-6 absolute at |ref| ~ 1)
    assert float((h > 0).ne(ref > 0).float().mean()) < 2e-5


def test_non_finite_features_stay_non_finite(dev):
    """A NaN or an infinity in a patch feature makes its row of H_bag non-finite, as in the reference's fp32 product
    (models/mcat/mcat.py:87) -- not a clamped finite value -- and leaves every other row alone."""
    rows = 500
    x, w, b, gen = _operands(rows, dev, 78)
    clean = ops.patch_fc_f32(x.clone(), torch.nn.Parameter(w), torch.nn.Parameter(b), 0.0)
    x2 = x.clone()
    x2[7, 100] = float("nan")
    x2[300, 5] = float("inf")
    h = ops.patch_fc_f32(x2, torch.nn.Parameter(w), torch.nn.Parameter(b), 0.0)
    assert not torch.isfinite(h[7]).any() and not torch.isfinite(h[300]).any()
    keep = torch.ones(rows, dtype=torch.bool, device=dev)
    keep[7] = keep[300] = False
    assert torch.isfinite(h[keep]).all()
    # (not bit-equal: with an infinity in the window feature_scale() falls back to 1, the clean window is scaled by its maximum)
    torch.testing.assert_close(h[keep], clean[keep], rtol=0, atol=5e-6)


def test_dropout_mask_lives_in_the_output_and_gates_the_backward(dev):
    rows, p = 20000, 0.25
    x, w, b, gen = _operands(rows, dev, 9)
    wp, bp = torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())
    torch.manual_seed(5)
    h = ops.patch_fc_f32(x, wp, bp, p)
    base = torch.relu(x.double() @ w.double().t() + b.double()).float()
    pos = base > 0.05
    kept = (h > 0) & pos
    rate = 1.0 - float(kept.sum()) / float(pos.sum())
    assert abs(rate - 0.25) < 5e-3, rate                     # 64 / 256 exactly representable
    scale = (h[kept] / base[kept])
    assert float((scale - 4.0 / 3.0).abs().max()) < 1e-3
    h2 = ops.patch_fc_f32(x, wp, bp, p)                        # a second call reserves new counters: a fresh mask
    assert float(((h > 0) != (h2 > 0)).float().mean()) > 0.05
    probe = torch.randn(rows, 256, device=dev, generator=gen)
    (h * probe).sum().backward()
    g = probe.double() * (h.detach() > 0) * (4.0 / 3.0)
    dw_ref, db_ref = g.t() @ x.double(), g.sum(0)
    assert float((wp.grad.double() - dw_ref).abs().max() / dw_ref.abs().max()) < 1e-4
    assert float((bp.grad.double() - db_ref).abs().max() / db_ref.abs().max()) < 1e-5


def test_other_geometries_are_not_claimed(dev):
    assert not ops.patch_fc_f32_supported(torch.empty(4, 1024, device=dev), torch.empty(128, 1024, device=dev))
    assert not ops.patch_fc_f32_supported(torch.empty(4, 1024, device=dev, dtype=torch.bfloat16), torch.empty(256, 1024, device=dev))
    assert ops.patch_fc_f32_supported(torch.empty(4, 1024, device=dev), torch.empty(256, 1024, device=dev))
