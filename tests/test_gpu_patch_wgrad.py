"""Weight gradient of the patch layer, dW_H = g^T X (models/mcat/mcat.py:24-29 backward), hand-written kernel
(mpo_patch_weight_grad) against the fp32 product of the same bf16 operands: the products are exact in fp32, only the
summation order differs.  Row counts cover the chunk (32), range (64 ranges) and tail edges; rows past the end of the
operands must not leak in (the kernel reads clamped rows and clears them)."""
import pytest
import torch

from multimodal_path_omic_amd import ops

pytestmark = pytest.mark.gpu


def _ref(g, x):
    return g.double().t() @ x.double()


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 2047, 2048, 2049, 6000, 48001])
@pytest.mark.parametrize("patch_dim", [1024, 512, 256])
def test_patch_weight_grad_matches_fp32_product(dev, rows, patch_dim):
    gen = torch.Generator(device=dev).manual_seed(rows + patch_dim)
    # operands carved out of larger buffers whose neighbours hold NaN: nothing outside [0, rows) may be read into the sum
    gbuf = torch.full((rows + 64, 256), float("nan"), device=dev, dtype=torch.bfloat16)
    xbuf = torch.full((rows + 64, patch_dim), float("nan"), device=dev, dtype=torch.bfloat16)
    g, x = gbuf[:rows], xbuf[:rows]
    g.copy_(torch.randn(rows, 256, device=dev, generator=gen) * (torch.rand(rows, 256, device=dev, generator=gen) > 0.3))
    x.copy_(torch.randn(rows, patch_dim, device=dev, generator=gen))
    out = torch.full((256, patch_dim), float("nan"), device=dev)
    ops.patch_weight_grad(g, x, out)
    ref = _ref(g, x)
    err = float((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    assert err < 2e-6, err


def test_patch_weight_grad_equals_the_library_split_k_product(dev):
    gen = torch.Generator(device=dev).manual_seed(3)
    rows = 32 * 1500
    g = (torch.randn(rows, 256, device=dev, generator=gen) * 0.01).to(torch.bfloat16)
    x = torch.randn(rows, 1024, device=dev, generator=gen).to(torch.bfloat16)
    a, b = torch.empty(256, 1024, device=dev), torch.empty(256, 1024, device=dev)
    ops.patch_weight_grad(g, x, a)
    ops._splitk_tn(g, x, b)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


def test_unsupported_shapes_take_the_library_path(dev):
    g = torch.randn(500, 128, device=dev).to(torch.bfloat16)           # embed 128 ('small'): not built, library product
    x = torch.randn(500, 1024, device=dev).to(torch.bfloat16)
    out = torch.empty(128, 1024, device=dev)
    ops.patch_weight_grad(g, x, out)
    torch.testing.assert_close(out.double(), _ref(g, x), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("patch_dim", [768, 1280])
def test_patch_dims_the_launcher_refuses_take_the_library_path(dev, patch_dim):
    """embed 256 but a patch width the kernel's 256-row-range split does not divide (the launcher wants 256 % (k / 256) == 0):
    the Python gate must route these to the library product instead of raising in backward."""
    gen = torch.Generator(device=dev).manual_seed(patch_dim)
    g = (torch.randn(3000, 256, device=dev, generator=gen) * 0.1).to(torch.bfloat16)
    x = torch.randn(3000, patch_dim, device=dev, generator=gen).to(torch.bfloat16)
    out = torch.empty(256, patch_dim, device=dev)
    ops.patch_weight_grad(g, x, out)
    ref = _ref(g, x)
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 1e-4
