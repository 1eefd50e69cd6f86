"""Diagnostic (not a test): the full-size bf16 NaCAGaT training window against the same slides one at a time
(tests/test_gpu_models.py::test_full_size_bf16_training_window_equals_per_slide), per-parameter gradient differences, with the
patch-side gradient of K2 on the one-pass kernel (csrc/k2_patchgrad.hip) and on the r02 path (library GEMM + epilogue pass);
plus the two paths against each other on identical inputs and, for scale, against an fp32 torch evaluation of the same d_bag."""
import sys

import torch

sys.path[:0] = [".", "tests", "tests/golden"]
from multimodal_path_omic_amd import harness, ops, synthetic as syn  # noqa: E402
from multimodal_path_omic_amd.ops import BagBatch  # noqa: E402
from test_gpu_models import build, ces_loss  # noqa: E402

dev = torch.device("cuda:0")
omic_sizes, seed, n, m = [256] * 6, 991, 6, 15000
g = syn.rng(seed)
wsis = [syn.normal(g, (m, 1024)).to(dev).to(torch.bfloat16) for _ in range(n)]
omics = [[syn.normal(g, (s,)).to(dev) for s in omic_sizes] for _ in range(n)]
labels = (torch.arange(n) % 4).to(dev)
cens = (torch.arange(n) % 2).float().to(dev)
om_w = [torch.stack([omics[b][i] for b in range(n)]) for i in range(len(omic_sizes))]
res = {}
for fused in (True, False):
    ops.k2_fused_patch_grad = fused
    model, _ = build("nacagat", omic_sizes, seed, dev, bag_dtype=torch.bfloat16)
    harness.train_window(model, BagBatch.from_list(wsis), om_w, labels, cens, grad_acc_step=n)
    gw = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad()
    for b in range(n):
        hz, sv, y, att = model(wsi=wsis[b], omics=omics[b])
        (ces_loss(hz, sv, labels[b:b + 1], cens[b:b + 1]) / n).backward()
    gs = {k: p.grad.clone() for k, p in model.named_parameters()}
    res[fused] = (gw, gs)
    worst = sorted(((float((gw[k] - gs[k]).abs().max()) / max(float(gs[k].abs().max()), 1e-3), k) for k in gw), reverse=True)[:5]
    print(f"fused={fused}: window vs per-slide, worst relative differences: " + ", ".join(f"{k} {e:.2e}" for e, k in worst), flush=True)
for which, name in ((0, "window"), (1, "per-slide")):
    a, b = res[True][which], res[False][which]
    worst = sorted(((float((a[k] - b[k]).abs().max()) / max(float(b[k].abs().max()), 1e-3), k) for k in a), reverse=True)[:4]
    print(f"one-pass kernel vs r02 path ({name}): " + ", ".join(f"{k} {e:.2e}" for e, k in worst), flush=True)
ops.k2_fused_patch_grad = True

# Where do window and per-slide part ways?  Record the operands of the two weight-gradient products of the backward (d_k, then
# d_h = the patch layer's pre-activation gradient) in both runs and count the elements that differ.
rec = []
orig = ops.patch_weight_grad


def spy(g_, x_, out_):
    rec.append(g_.detach().float().clone())
    return orig(g_, x_, out_)


ops.patch_weight_grad = spy
for fused in (True, False):
    ops.k2_fused_patch_grad = fused
    model, _ = build("nacagat", omic_sizes, seed, dev, bag_dtype=torch.bfloat16)
    rec.clear()
    harness.train_window(model, BagBatch.from_list(wsis), om_w, labels, cens, grad_acc_step=n)
    dk_w, dh_w = rec[0], rec[1]
    rec.clear()
    model.zero_grad()
    for b in range(n):
        hz, sv, y, att = model(wsi=wsis[b], omics=omics[b])
        (ces_loss(hz, sv, labels[b:b + 1], cens[b:b + 1]) / n).backward()
    dk_s, dh_s = torch.cat(rec[0::2]), torch.cat(rec[1::2])
    for name, a, b_ in (("d_k", dk_w, dk_s), ("d_h", dh_w, dh_s)):
        diff = (a - b_).abs()
        nz = a.abs() > 0
        print(f"fused={fused} {name}: {float((diff > 0).float().mean()):.4f} of the elements differ between window and per-slide; "
              f"max |diff| / max |value| {float(diff.max() / a.abs().max()):.2e}; median |value| {float(a[nz].abs().median()):.2e}", flush=True)
ops.patch_weight_grad = orig
ops.k2_fused_patch_grad = True
