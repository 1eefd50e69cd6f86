"""Probe: why a trivial kernel costs 4.7 us inside the captured window step but 1.7 us in a chain of identical kernels."""
import sys, time, torch
sys.path[:0] = ["."]
from multimodal_path_omic_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(192, 256, device=dev)
big = torch.randn(256 * 1024 * 1024, device=dev)       # 1 GiB
big2 = torch.empty_like(big)
w = torch.randn(256, 256, device=dev) / 16
b = torch.zeros(256, device=dev)
fns = [lambda y: y + 1.0, lambda y: y * 1.001, lambda y: y - 0.5, lambda y: torch.neg(y), lambda y: torch.abs(y),
       lambda y: torch.relu(y), lambda y: torch.sigmoid(y), lambda y: torch.tanh(y), lambda y: torch.sin(y),
       lambda y: torch.cos(y), lambda y: torch.clamp(y, -1, 1), lambda y: torch.sqrt(torch.abs(y))]
def A(n):
    y = x
    for _ in range(n): y = y + 1.0
    return y
def B(n):
    big2.copy_(big)
    y = x
    for _ in range(n): y = y + 1.0
    return y
def C(n):
    y = x
    for i in range(n): y = fns[i % len(fns)](y)
    return y
def D(n):
    y = x
    for i in range(n // 3):
        y = ops.linear(y, w, b); y = torch.nn.functional.layer_norm(y, (256,)); y = y + 1.0
    return y
def E(n):   # every trivial kernel separated by a 64 MB copy
    y = x
    for i in range(n):
        big2[:16 * 1024 * 1024].copy_(big[:16 * 1024 * 1024]); y = y + 1.0
    return y
for name, fn, extra in (("A same trivial", A, 0), ("B 1GiB copy + same trivial", B, 1), ("C 12 distinct trivial", C, 0),
                        ("D gemm/ln/add mix", D, 0), ("E trivial after 64MB copies", E, 0)):
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
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(f"{name:30s} graph: {dt * 1e6:9.1f} us total, {dt / n * 1e6:6.2f} us per chain element", flush=True)
