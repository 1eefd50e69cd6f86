"""Diagnostic timing of K2's forward bag pass (bag_rowdot_gated, fp32 keys): HIP-event time per launch over two
alternating resident 32 x 15000 x 256 windows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd import _lib as L
from multimodal_path_omic_amd.ops import BagBatch, make_cu
dev = torch.device("cuda:0")
torch.manual_seed(0)
E, n_q, B, M = 256, 6, 32, 15000
lengths = [M] * B
cu = make_cu(lengths, dev)
bags = [torch.randn(B * M, E, device=dev) for _ in range(2)]
batch = BagBatch(bags[0], cu, lengths)
plan = batch.plan()
qs2 = torch.randn(B * n_q, E, device=dev) * 0.05
tq = torch.tanh(torch.randn(B * n_q, E, device=dev))
maps = torch.empty(2, n_q * B * M, device=dev)
lib = L.lib()
s = torch.cuda.current_stream().cuda_stream
def run(i):
    L.check(lib.mpo_nacagat_fwd_bagpass(L.ptr(bags[i & 1]), L.ptr(cu), B, E, L.ptr(qs2), L.ptr(tq), L.ptr(maps[0]), L.ptr(maps[1]),
                                        n_q, M, plan, s), "k2 fwd bagpass")
for i in range(4): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 40
e0.record()
for i in range(n): run(i)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print(f"probe {os.environ.get('MPO_K2_PROBE', '0')}: {us:.1f} us per launch = {B * M * E * 4 / us / 1e6:.2f} TB/s")
