"""rocprofv3 workload: K2's forward bag pass (bag_rowdot_gated) on a 32 x 15000 x 256 fp32 key window (491.52 MB),
two distinct resident windows alternated so every launch streams from HBM."""
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
batch = BagBatch(bags[0], cu, lengths)      # keep the batch alive: it owns the plan's device array
plan = batch.plan()
qs2 = torch.randn(B * n_q, E, device=dev) * 0.05
tq = torch.tanh(torch.randn(B * n_q, E, device=dev))
maps = torch.empty(2, n_q * B * M, device=dev)
lib = L.lib()
s = torch.cuda.current_stream().cuda_stream
for i in range(12):
    L.check(lib.mpo_nacagat_fwd_bagpass(L.ptr(bags[i & 1]), L.ptr(cu), B, E, L.ptr(qs2), L.ptr(tq), L.ptr(maps[0]), L.ptr(maps[1]),
                                        n_q, M, plan, s), "k2 fwd bagpass")
torch.cuda.synchronize()
