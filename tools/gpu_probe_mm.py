"""Probe (not a test): which mixed-precision GEMM forms torch 2.10/ROCm offers, and their timings."""
import torch, time
dev = torch.device("cuda:0")
M, K, N = 480000, 256, 256
h = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = torch.randn(N, K, device=dev) / 16
whi = w.to(torch.bfloat16); wlo = (w - whi.float()).to(torch.bfloat16)
b = torch.randn(N, device=dev)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
ref = torch.nn.functional.linear(h.float(), w, b)
print("fp32 linear (incl. bag->fp32 copy) us", timeit(lambda: torch.nn.functional.linear(h.float(), w, b)))
try:
    k1 = torch.mm(h, whi.t(), out_dtype=torch.float32)
    print("mm out_dtype ok; us", timeit(lambda: torch.mm(h, whi.t(), out_dtype=torch.float32)))
except Exception as e:
    print("mm out_dtype FAILED", repr(e)[:200])
try:
    out = b.expand(M, N).contiguous()
    k = torch.addmm(out, h, whi.t(), out_dtype=torch.float32)
    print("addmm out_dtype ok", (k - (ref - 0)).abs().max().item())
    def two():
        k = torch.addmm(out, h, whi.t(), out_dtype=torch.float32)
        return torch.addmm(k, h, wlo.t(), out_dtype=torch.float32)
    k = two()
    print("hi/lo addmm err", (k - ref).abs().max().item(), "rel", ((k - ref).abs().max() / ref.abs().max()).item(), "us", timeit(two))
except Exception as e:
    print("addmm out_dtype FAILED", repr(e)[:300])
try:
    w2 = torch.stack([whi, wlo])                      # (2, N, K)
    h2 = h.unsqueeze(0).expand(2, M, K)
    kk = torch.bmm(h2, w2.transpose(1, 2), out_dtype=torch.float32)
    print("bmm expand ok; us", timeit(lambda: torch.bmm(h2, w2.transpose(1, 2), out_dtype=torch.float32)))
except Exception as e:
    print("bmm expand FAILED", repr(e)[:300])
# concatenated-N form: one GEMM -> [K_hi | K_lo]  (then a fused add elsewhere)
wcat = torch.cat([whi, wlo], 0)
print("cat-N mm us", timeit(lambda: torch.mm(h, wcat.t(), out_dtype=torch.float32)))
