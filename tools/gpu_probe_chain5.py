"""Probe: does an elementwise kernel cost more when its input was just written by the tiled GEMM kernel?"""
import sys, time, torch
sys.path[:0] = ["."]
from multimodal_path_omic_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(384, 256, device=dev)
w = torch.randn(256, 256, device=dev) / 16
b = torch.zeros(256, device=dev)
m = torch.randn(384, 256, device=dev) * 0.01 + 1.0
def run(name, fn, n=96):
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
        print(f"{name:50s}: {dt / n * 1e6:6.2f} us per kernel", flush=True)
def gemm_only(n):
    y = x
    for _ in range(n): y = ops.linear(y, w, b)
    return y
def mul_only(n):
    y = x
    for _ in range(n): y = y * m
    return y
def gemm_mul_dep(n):
    y = x
    for _ in range(n // 2): y = ops.linear(y, w, b) * m
    return y
def gemm_mul_indep(n):
    y, z = x, x
    for _ in range(n // 2):
        y = ops.linear(y, w, b); z = z * m
    return y, z
def gemm_ln_dep(n):
    y = x
    for _ in range(n // 2): y = torch.nn.functional.layer_norm(ops.linear(y, w, b), (256,))
    return y
run("gemm 384x256x256 chain", gemm_only)
run("torch mul 98304 chain", mul_only)
run("gemm -> mul (dependent) alternating", gemm_mul_dep)
run("gemm, mul (independent data) alternating", gemm_mul_indep)
run("gemm -> torch layer_norm (dependent) alternating", gemm_ln_dep)
