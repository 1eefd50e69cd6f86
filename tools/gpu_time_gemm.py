import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd import _lib as L
dev = torch.device("cuda:0"); lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (R, I, O) in [(192, 256, 256), (192, 256, 768), (192, 512, 256), (32, 256, 256), (192, 256, 512)]:
    x = torch.randn(R, I, device=dev); w = torch.randn(O, I, device=dev) * 0.05; b = torch.randn(O, device=dev)
    y = torch.empty(R, O, device=dev); dy = torch.randn(R, O, device=dev); dx = torch.empty(R, I, device=dev)
    dw = torch.empty(O, I, device=dev); db = torch.empty(O, device=dev)
    tf = timeit(lambda: lib.mpo_linear_forward(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), R, I, O, 1.0, 1, st))
    ti = timeit(lambda: lib.mpo_linear_backward_input(L.ptr(dy), L.ptr(w), L.ptr(dx), R, I, O, 1.0, 0, st))
    tw = timeit(lambda: lib.mpo_linear_backward_weight(L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(db), R, I, O, 1.0, st))
    tt = timeit(lambda: torch.nn.functional.linear(x, w, b))
    print(f"R={R} I={I} O={O}: fwd {tf:.1f} us | bwd_input {ti:.1f} us | bwd_weight {tw:.1f} us | torch F.linear {tt:.1f} us")
