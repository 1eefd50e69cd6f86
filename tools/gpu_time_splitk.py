import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd.ops import _splitk_tn
dev = torch.device("cuda:0")
T = 480000
g = torch.randn(T, 256, device=dev).to(torch.bfloat16); x = torch.randn(T, 1024, device=dev).to(torch.bfloat16)
out = torch.empty(256, 1024, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for chunk in (2048, 3000, 3750, 4096, 5000, 7500, 8192, 15000, 16384, 32768):
    print(chunk, "rows/chunk ->", T // max(1, T // chunk), "x", max(1, T // chunk), f"{timeit(lambda: _splitk_tn(g, x, out, chunk)):.1f} us", flush=True)
db = torch.empty(256, device=dev)
print("bias sum", f"{timeit(lambda: torch.sum(g, 0, dtype=torch.float32, out=db)):.1f} us")
ones = torch.ones(1, T, device=dev, dtype=torch.bfloat16)
print("ones @ g", f"{timeit(lambda: torch.mm(ones, g)):.1f} us")
print("g.view(58,-1,256).sum(1) bf16->f32 2-stage", f"{timeit(lambda: g.view(1875, 256, 256).sum(1, dtype=torch.float32).sum(0)):.1f} us")
