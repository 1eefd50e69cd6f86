"""Probe (not a test): per-kernel cost of a dependent chain inside a HIP graph -- trivial kernels vs the small GEMM."""
import sys, time, torch
sys.path[:0] = ["."]
from multimodal_path_omic_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(192, 256, device=dev)
w = torch.randn(256, 256, device=dev) / 16
w2 = torch.randn(512, 256, device=dev) / 16
w3 = torch.randn(256, 512, device=dev) / 16
b = torch.zeros(256, device=dev); b2 = torch.zeros(512, device=dev)
def chain_trivial(n):
    y = x
    for _ in range(n): y = y + 1.0
    return y
def chain_gemm(n):
    y = x
    for _ in range(n): y = ops.linear(y, w, b)
    return y
def chain_ffn(n):
    y = x
    for _ in range(n // 2): y = ops.linear(ops.linear(y, w2, b2), w3, b)
    return y
def chain_ln(n):
    y = x
    for _ in range(n): y = torch.nn.functional.layer_norm(y, (256,))
    return y
for name, fn in (("trivial add", chain_trivial), ("gemm 192x256x256", chain_gemm), ("ffn 256->512->256", chain_ffn), ("torch layer_norm", chain_ln)):
    n = 100
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
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        print(f"{name:24s} graph: {dt / n * 1e6:6.2f} us per kernel", flush=True)
