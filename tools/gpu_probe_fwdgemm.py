"""Probe (not a test): forms of the patch-layer forward GEMM  Y[480k,256] = X[480k,1024] W[256,1024]^T  (bf16)."""
import torch
dev = torch.device("cuda:0")
T = 480000
x = torch.randn(T, 1024, device=dev).to(torch.bfloat16)
w = (torch.randn(256, 1024, device=dev) / 32).to(torch.bfloat16)
b = torch.randn(256, device=dev).to(torch.bfloat16)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
wt = w.t().contiguous()
print("mm(x, w.t())            ", f"{timeit(lambda: torch.mm(x, w.t())):.1f} us", flush=True)
print("mm(x, wt contiguous)    ", f"{timeit(lambda: torch.mm(x, wt)):.1f} us", flush=True)
print("linear(x, w, b)         ", f"{timeit(lambda: torch.nn.functional.linear(x, w, b)):.1f} us", flush=True)
try:
    print("_addmm_activation relu  ", f"{timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)):.1f} us", flush=True)
except Exception as e:
    print("_addmm_activation failed", repr(e)[:200])
for nb in (8, 32, 64, 128, 256):
    xb = x.view(nb, T // nb, 1024)
    print(f"bmm {nb:4d} batches         ", f"{timeit(lambda: torch.matmul(xb, w.t())):.1f} us", flush=True)
    # (torch.bmm with a stride-0 expanded weight faulted the GPU on ROCm 7.2 / torch 2.10: never pass one)
y = torch.mm(x, w.t())
print("copy 245MB (epilogue-like)", f"{timeit(lambda: y.add_(1)):.1f} us")
