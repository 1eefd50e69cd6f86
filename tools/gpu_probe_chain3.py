"""Probe: per-node cost of this library's own trivial kernel (counters_bump) in a captured chain, alone and alternating
with other kernels."""
import sys, time, torch
sys.path[:0] = ["."]
from multimodal_path_omic_amd import ops
dev = torch.device("cuda:0")
e = torch.zeros(1, dtype=torch.int64, device=dev)
t = torch.zeros(1, dtype=torch.int32, device=dev)
x = torch.randn(192, 256, device=dev)
w = torch.randn(256, 256, device=dev) / 16
b = torch.zeros(256, device=dev)
def A(n):
    for _ in range(n): ops.bump_step_counters(e, t)
def B(n):
    y = x
    for _ in range(n // 2):
        ops.bump_step_counters(e, t); y = y + 1.0
    return y
def C(n):
    y = x
    for _ in range(n // 2):
        ops.bump_step_counters(e, t); y = ops.linear(y, w, b)
    return y
def D(n):
    y = x
    for _ in range(n): y = ops.linear(y, w, b, "relu") if _ % 2 else ops.linear(y, w, b)
    return y
for name, fn in (("A bump x96", A), ("B bump/add alternating", B), ("C bump/gemm alternating", C), ("D gemm/gemm(relu) same kernel", D)):
    n = 96
    with torch.no_grad():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn(n); fn(n)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn(n)
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"{name:34s}: {dt / n * 1e6:6.2f} us per kernel", flush=True)
