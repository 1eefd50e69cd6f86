"""rocprofv3 workload: only the two K1 bag-pass kernels on a 32 x 15000 x 256 bf16 window (245.76 MB),
two distinct resident windows alternated so every launch streams from HBM (working set >> 256 MB L3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd import _lib as L
from multimodal_path_omic_amd.ops import make_cu
dev = torch.device("cuda:0")
torch.manual_seed(0)
E, n_q, B, M = 256, 6, 32, 15000
cu = make_cu([M] * B, dev)
bags = [torch.relu(torch.randn(B * M, E, device=dev)).to(torch.bfloat16) for _ in range(2)]
dbag = torch.empty_like(bags[0])
qk2 = torch.randn(B * n_q, E, device=dev) * 0.05
dctx = torch.randn(B * n_q, E, device=dev) * 0.05
lib = L.lib()
splits = lib.mpo_coattn_target_workgroups() // B + 1
part_ml = torch.empty(B * splits * 32, device=dev)
part_ctx = torch.empty(B * splits * n_q * E, device=dev)
lse2 = torch.full((B * n_q,), 14.0, device=dev)
delta = torch.zeros(B * n_q, device=dev)
s = torch.cuda.current_stream().cuda_stream
for i in range(10):
    L.check(lib.mpo_coattn_fwd_bagpass(L.ptr(bags[i & 1]), 1, L.ptr(cu), B, E, L.ptr(qk2), L.ptr(part_ml), L.ptr(part_ctx),
                                       None, n_q, M, None, s), "fwd")
    L.check(lib.mpo_coattn_bwd_bagpass(L.ptr(bags[i & 1]), 1, L.ptr(cu), B, E, L.ptr(qk2), L.ptr(lse2), L.ptr(dctx),
                                       L.ptr(delta), None, L.ptr(dbag), L.ptr(part_ctx), n_q, M, None, s), "bwd")
torch.cuda.synchronize()
