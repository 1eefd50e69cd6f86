"""The patch-layer bag pass (rows H2 / f1; K1=1: followed by K1's forward bag pass) timed alone on a 32 x 15000 x 1024 bf16 window (two resident
windows alternated); also the rocprofv3 workload of profiles/r02_f1_*.  DROP=<p> sets the dropout rate (default 0.25)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from multimodal_path_omic_amd import _lib as L
from multimodal_path_omic_amd.ops import BagBatch, make_cu

dev = torch.device("cuda:0")
window, patches, E, n_q = 32, 15000, 256, 6
lib = L.lib()
lengths = [patches] * window
cu = make_cu(lengths, dev)
xs = [torch.randn(window * patches, 1024, device=dev).to(torch.bfloat16) for _ in range(2)]
batch = BagBatch(xs[0], cu, lengths)
plan = batch.plan()
w = torch.randn(E, 1024, device=dev) / 32
wb = torch.empty(E, 1024, device=dev, dtype=torch.bfloat16)
stream = torch.cuda.current_stream(dev)
L.check(lib.mpo_pack_patch_weight(L.ptr(w), L.ptr(wb), E, 1024, stream.cuda_stream), "pack")
bias = torch.randn(E, device=dev) * 0.1
h_out = torch.empty(window * patches, E, device=dev, dtype=torch.bfloat16)
parts = lib.mpo_coattn_target_workgroups() + window
part_ml = torch.empty(parts * 32, device=dev)
part_ctx = torch.empty(parts * n_q * E, device=dev)
qk2 = torch.randn(window * n_q, E, device=dev) * 0.05
drop = float(os.environ.get("DROP", "0.25"))


with_k1 = os.environ.get("K1", "0") == "1"          # K1=1: the patch-layer pass followed by K1's forward pass (two launches)


def launch(i):
    L.check(lib.mpo_patch_coattn_fwd_bagpass(L.ptr(xs[i & 1]), L.ptr(wb), L.ptr(bias), L.ptr(cu), window, L.ptr(qk2) if with_k1 else None, L.ptr(h_out),
                                             L.ptr(part_ml), L.ptr(part_ctx), n_q, patches, drop, 1, 0, plan, stream.cuda_stream), "bagpass")


for i in range(3):
    launch(i)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for i, (s, e) in enumerate(evs):
    s.record()
    for j in range(4):
        launch(4 * i + j)
    e.record()
torch.cuda.synchronize()
us = sorted(s.elapsed_time(e) * 250 for s, e in evs)
print(f"drop={drop} k1={int(with_k1)}: min {us[0]:.1f} us, median {us[len(us) // 2]:.1f} us "
      f"({window * patches * 1280 * 2 / us[len(us) // 2] / 1e6:.2f} TB/s algorithmic)")
