"""Diagnostic timing: dW_H = g^T X (480 000 x 256 bf16, 480 000 x 1024 bf16) hand-written kernel against the library's batched split-K product of the same operands (the r01 form; the product code has no library product any more),
two resident operand sets alternated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
T = 32 * 15000
gs = [(torch.randn(T, 256, device=dev) * 0.01).to(torch.bfloat16) for _ in range(2)]
xs = [torch.randn(T, 1024, device=dev).to(torch.bfloat16) for _ in range(2)]
out = torch.empty(256, 1024, device=dev)
def timeit(fn, n=20):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_new = timeit(lambda i: ops.patch_weight_grad(gs[i & 1], xs[i & 1], out))
def splitk(g, x, o, s=64):          # 64 batches of 7 500 rows: the library's 256 x 256 tiles fill the 256 CUs once
    c = g.shape[0] // s
    torch.sum(torch.bmm(g[:s * c].view(s, c, -1).transpose(1, 2), x[:s * c].view(s, c, -1), out_dtype=torch.float32), 0, out=o)
t_lib = timeit(lambda i: splitk(gs[i & 1], xs[i & 1], out))
b = T * (256 + 1024) * 2
print(f"hand-written {t_new:.1f} us ({b / t_new / 1e6:.2f} TB/s alg) | library split-K {t_lib:.1f} us ({b / t_lib / 1e6:.2f} TB/s alg)")
