"""Probe: chain cost of elementwise kernels on the tail's tensor sizes (98 304 and 393 216 floats) in a captured graph."""
import sys, time, torch
dev = torch.device("cuda:0")
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
        print(f"{name:44s}: {dt / n * 1e6:6.2f} us per kernel", flush=True)
for numel in (49152, 98304, 393216, 1572864):
    a = torch.randn(numel, device=dev); b = torch.randn(numel, device=dev) * 0.01 + 1.0
    def mul_chain(n, a=a, b=b):
        y = a
        for _ in range(n): y = y * b
        return y
    run(f"torch mul chain, {numel} floats", mul_chain)
    x2 = torch.randn(numel // 256, 256, device=dev)
    def ln_chain(n, x2=x2):
        y = x2
        for _ in range(n): y = torch.nn.functional.layer_norm(y, (256,))
        return y
    run(f"torch layer_norm chain, {numel // 256} x 256", ln_chain)
