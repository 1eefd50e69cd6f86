import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd import _lib as L
dev = torch.device("cuda:0")
lib = L.lib()
rows, cols = 480000, 256
h = [torch.randn(rows, cols, device=dev).to(torch.bfloat16) for _ in range(2)]
bias = torch.randn(cols, device=dev)
def timeit(fn, n=10):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
for p in (0.0, 0.25):
    t = timeit(lambda i: L.check(lib.mpo_patch_epilogue_forward(L.ptr(h[i & 1]), L.ptr(bias), rows, cols, p, 1, 2, None, st), "e"))
    print(f"epilogue p={p}: {t:.1f} us  {2*rows*cols*2/t/1e6:.2f} TB/s")
t = timeit(lambda i: h[i & 1].mul_(1.0))
print(f"torch in-place mul bf16: {t:.1f} us")
t = timeit(lambda i: torch.relu_(h[i & 1]))
print(f"torch relu_: {t:.1f} us")
g = torch.empty_like(h[0])
t = timeit(lambda i: L.check(lib.mpo_patch_epilogue_backward(L.ptr(h[i & 1]), L.ptr(h[1 - (i & 1)]), L.ptr(g), h[0].numel(), 0.25, st), "b"))
print(f"epilogue bwd: {t:.1f} us  {3*rows*cols*2/t/1e6:.2f} TB/s")
