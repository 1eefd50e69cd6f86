import torch, time
dev = torch.device("cuda:0")
M, K, N = 480000, 1024, 256
x = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
b = torch.randn(N, device=dev).to(torch.bfloat16)
wt = w.t().contiguous()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
import torch.nn.functional as F
for name, fn in [
    ("F.linear(x,w,b)", lambda: F.linear(x, w, b)),
    ("F.linear(x,w)", lambda: F.linear(x, w)),
    ("x @ wt (NN)", lambda: x @ wt),
    ("addmm(b, x, wt)", lambda: torch.addmm(b, x, wt)),
    ("(w @ x.t()) transposed out", lambda: w @ x.t()),
    ("chunks of 65536 rows F.linear", lambda: [F.linear(x[i:i+65536], w, b) for i in range(0, M, 65536)]),
]:
    t = timeit(fn)
    print(f"{name:40s} {t*1e3:8.1f} us  {2*M*K*N/t/1e9:7.1f} TF/s")
try:
    torch.backends.cuda.preferred_blas_library("cublas")
    print("rocblas F.linear", timeit(lambda: F.linear(x, w, b))*1e3)
    torch.backends.cuda.preferred_blas_library("cublaslt")
except Exception as e:
    print("pref", e)
# dW shape: (N x M) @ (M x K)
dh = torch.randn(M, N, device=dev).to(torch.bfloat16)
print("dW = dh.t() @ x", timeit(lambda: dh.t() @ x)*1e3, "us")
# fp32 K projection for NaCAGaT
h32 = torch.randn(M, N, device=dev)
wk = torch.randn(N, N, device=dev) * 0.05
print("K fp32 linear", timeit(lambda: F.linear(h32, wk))*1e3, "us")
