"""Diagnostic timing of the K2 key projection kernel alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd import _lib as L
dev = torch.device("cuda:0")
rows, E = 480000, 256
hs = [torch.relu(torch.randn(rows, E, device=dev)).to(torch.bfloat16) for _ in range(2)]
w = torch.randn(E, E, device=dev) / 16
b = torch.randn(E, device=dev)
outs = [torch.empty(rows, E, device=dev) for _ in range(2)]
lib = L.lib()
s = torch.cuda.current_stream().cuda_stream
def run(i):
    L.check(lib.mpo_key_projection(L.ptr(hs[i & 1]), rows, E, L.ptr(w), L.ptr(b), L.ptr(outs[i & 1]), s), "kp")
for i in range(4): run(i)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(20): run(i)
e.record(); torch.cuda.synchronize()
us = a.elapsed_time(e) / 20 * 1e3
print(f"key projection 480000 x 256: {us:.1f} us  ({rows * E * 6 / us / 1e6:.2f} TB/s of 737 MB)", flush=True)
