"""Diagnostic timing of the K1 module (whole C-ABI call, not single kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import torch
from multimodal_path_omic_amd.blocks import CoAttention
from multimodal_path_omic_amd.ops import BagBatch
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us
mod = CoAttention(256, 1).to(dev)
for (B, M, dt) in [(1, 15000, torch.bfloat16), (32, 15000, torch.bfloat16), (64, 15000, torch.bfloat16), (1, 100000, torch.float32), (8, 100000, torch.float32), (32, 15000, torch.float32)]:
    bags = [torch.relu(torch.randn(M, 256, device=dev)).to(dt) for _ in range(B)]
    batch = BagBatch.from_list(bags); del bags
    data = batch.data.requires_grad_(True)
    q = torch.randn(B, 6, 256, device=dev, requires_grad=True)
    b2 = batch.with_data(data)
    esz = 2 if dt == torch.bfloat16 else 4
    bytes_f = B * M * 256 * esz
    with torch.no_grad():
        t_f = timeit(lambda: mod.forward_window(q, b2, False))
        t_fa = timeit(lambda: mod.forward_window(q, b2, True))
    def fb():
        out, _ = mod.forward_window(q, b2, False)
        out.sum().backward()
    t_fb = timeit(fb)
    print(f"B={B} M={M} {dt}: fwd {t_f:.1f} us ({bytes_f/t_f/1e6:.2f} TB/s alg) | fwd+map {t_fa:.1f} us | fwd+bwd {t_fb:.1f} us ({3*bytes_f/t_fb/1e6:.2f} TB/s alg)")
