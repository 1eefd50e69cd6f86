"""Probe: the fast small-row GEMM (rows x 256 x 256) in a captured chain of 48 dependent launches, rows swept."""
import sys, time, torch
sys.path[:0] = ["."]
from multimodal_path_omic_amd import ops
dev = torch.device("cuda:0")
w = torch.randn(256, 256, device=dev) / 16
b = torch.zeros(256, device=dev)
for rows in (16, 64, 192, 384, 768, 1536, 3072, 6144, 12288):
    x = torch.randn(rows, 256, device=dev)
    def fn(n):
        y = x
        for _ in range(n): y = ops.linear(y, w, b)
        return y
    n = 48
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
    wgs = rows // 16 * 16
    print(f"rows {rows:6d} ({wgs:6d} workgroups): {dt / n * 1e6:6.2f} us per launch, {2 * rows * 256 * 256 / (dt / n) / 1e12:6.2f} TFLOP/s", flush=True)
