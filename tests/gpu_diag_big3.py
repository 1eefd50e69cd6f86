"""Diagnostic: fp32 library GEMM of the 'big' patch layer on the GPU vs CPU fp32 / fp64."""
import sys, torch
sys.path[:0] = [".", "tests", "tests/golden"]
import cases as C
from multimodal_path_omic_amd import synthetic as syn
dev = torch.device("cuda:0")
for d in (256, 512):
    g = syn.rng(1)
    x = syn.normal(g, (1200, 1024)); w = syn.normal(g, (d, 1024)) / 32; b = syn.normal(g, (d,)) * 0.1
    ref64 = torch.nn.functional.linear(x.double(), w.double(), b.double())
    cpu32 = torch.nn.functional.linear(x, w, b)
    gpu32 = torch.nn.functional.linear(x.to(dev), w.to(dev), b.to(dev)).cpu()
    print(d, "cpu32 vs 64", float((cpu32 - ref64).abs().max()), "gpu32 vs 64", float((gpu32 - ref64).abs().max()),
          "sign flips gpu vs cpu", int(((gpu32 > 0) != (cpu32 > 0)).sum()), "allow_tf32", torch.backends.cuda.matmul.allow_tf32,
          torch.get_float32_matmul_precision(), flush=True)
    gy = syn.normal(g, (1200, d))
    dw_cpu = gy.t() @ x
    dw_gpu = (gy.to(dev).t() @ x.to(dev)).cpu()
    dw64 = gy.double().t() @ x.double()
    print("   dW: cpu32 vs 64", float((dw_cpu - dw64).abs().max() / dw64.abs().max()), "gpu32 vs 64", float((dw_gpu - dw64).abs().max() / dw64.abs().max()))
