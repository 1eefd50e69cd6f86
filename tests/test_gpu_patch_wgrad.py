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


@pytest.mark.parametrize("embed,patch_dim", [(128, 1024), (512, 1024), (128, 128), (512, 512)])
@pytest.mark.parametrize("rows", [1, 33, 2049, 48001])
def test_small_and_big_embed_widths(dev, embed, patch_dim, rows):
    """model_size 'small' (embed 128: the g image's upper half repeats the lower, those rows of dW are not written) and
    'big' (512: one pass per 256 columns of g, row pitch 1024 B) on the same kernel (models/mcat/mcat.py:16-21); the square
    cases are NaCAGaT's key-projection weight gradient d_k^T H_bag at those sizes (models/nacagat/nacagat.py:17-18)."""
    gen = torch.Generator(device=dev).manual_seed(rows + embed)
    gbuf = torch.full((rows + 64, embed), float("nan"), device=dev, dtype=torch.bfloat16)
    xbuf = torch.full((rows + 64, patch_dim), float("nan"), device=dev, dtype=torch.bfloat16)
    g, x = gbuf[:rows], xbuf[:rows]
    g.copy_(torch.randn(rows, embed, device=dev, generator=gen) * (torch.rand(rows, embed, device=dev, generator=gen) > 0.3))
    x.copy_(torch.randn(rows, patch_dim, device=dev, generator=gen))
    guard = torch.full((embed + 8, patch_dim), float("nan"), device=dev)
    out = guard[:embed]
    ops.patch_weight_grad(g, x, out)
    ref = _ref(g, x)
    err = float((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    assert err < 2e-6, err
    assert torch.isnan(guard[embed:]).all()                 # nothing written past dW's rows


def test_a_patch_matrix_of_more_than_4_gib_goes_in_row_segments(dev):
    """The kernel's DMA offsets are 32-bit: 2 200 000 x 1024 bf16 rows (4.5 GB) are two launches whose second accumulates."""
    rows = 2_200_000
    gen = torch.Generator(device=dev).manual_seed(9)
    x = torch.empty(rows, 1024, device=dev, dtype=torch.bfloat16)
    g = torch.empty(rows, 256, device=dev, dtype=torch.bfloat16)
    for r0 in range(0, rows, 200_000):                      # (filled in pieces: no multi-GB fp32 temporaries)
        r1 = min(rows, r0 + 200_000)
        x[r0:r1] = torch.randn(r1 - r0, 1024, device=dev, generator=gen)
        g[r0:r1] = torch.randn(r1 - r0, 256, device=dev, generator=gen) * 0.05
    out = torch.full((256, 1024), float("nan"), device=dev)
    ops.patch_weight_grad(g, x, out)
    ref = torch.zeros(256, 1024, device=dev, dtype=torch.float64)
    for r0 in range(0, rows, 100_000):
        r1 = min(rows, r0 + 100_000)
        ref += g[r0:r1].double().t() @ x[r0:r1].double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    assert err < 5e-6, err


@pytest.mark.parametrize("embed,patch_dim", [(256, 768), (256, 1280), (64, 1024)])
def test_geometries_the_kernel_is_not_built_for_raise(dev, embed, patch_dim):
    """One code path: no library product behind the kernel -- a width it does not cover is an error, not a slow detour."""
    g = torch.zeros(3000, embed, device=dev, dtype=torch.bfloat16)
    x = torch.zeros(3000, patch_dim, device=dev, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="patch weight gradient"):
        ops.patch_weight_grad(g, x, torch.empty(embed, patch_dim, device=dev))
