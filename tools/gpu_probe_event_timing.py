"""How much of bench.py's event-timed roofline figure is the event pair itself?  The patch-layer kernel (32 x 15 000 x 1024 bf16)
timed three ways, each launch after ~1 ms of other work on the stream (a stand-in for the window step: a few big copies):
  A  plain events around one eager launch (bench.py's method)
  B  events recorded as EXTERNAL nodes of a captured graph [filler, e0, kernel, e1] (no host launch between the records)
  C  plain events around TWO launches (the second one starts hot), per-launch = half"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from multimodal_path_omic_amd import _lib as L
from multimodal_path_omic_amd.ops import BagBatch, make_cu

dev = torch.device("cuda:0")
window, patches, E = 32, 15000, 256
lib = L.lib()
lengths = [patches] * window
cu = make_cu(lengths, dev)
xs = [torch.randn(window * patches, 1024, device=dev).to(torch.bfloat16) for _ in range(2)]
batch = BagBatch(xs[0], cu, lengths)                    # (kept alive: the plan points at its device-side work list)
plan = batch.plan()
w = torch.randn(E, 1024, device=dev) / 32
wb = torch.empty(E, 1024, device=dev, dtype=torch.bfloat16)
stream = torch.cuda.current_stream(dev)
L.check(lib.mpo_pack_patch_weight(L.ptr(w), L.ptr(wb), E, 1024, stream.cuda_stream), "pack")
bias = torch.randn(E, device=dev) * 0.1
h_out = torch.empty(window * patches, E, device=dev, dtype=torch.bfloat16)
fill_a = torch.empty(512 * 1024 * 1024 // 4, device=dev)
fill_b = torch.empty_like(fill_a)


def launch(i, s):
    L.check(lib.mpo_patch_coattn_fwd_bagpass(L.ptr(xs[i & 1]), L.ptr(wb), L.ptr(bias), L.ptr(cu), window, None, L.ptr(h_out),
                                             None, None, 6, patches, 0.25, 1, 0, plan, s.cuda_stream), "bagpass")


def filler():
    for _ in range(4):
        fill_b.copy_(fill_a)


for i in range(3):
    filler(); launch(i, stream)
torch.cuda.synchronize()
# A
ts = []
for i in range(20):
    filler()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(i, stream); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"A eager event pair        : avg {sum(ts) / len(ts):.1f} us  min {min(ts):.1f}")
# E: the event pair around (almost) nothing -- a 64-thread counter kernel -- in the same position
cnt = torch.zeros(4, dtype=torch.int64, device=dev)
ts = []
for i in range(20):
    filler()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); cnt.add_(1); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"E pair around a tiny kernel: avg {sum(ts) / len(ts):.1f} us  min {min(ts):.1f}")
# C
ts = []
for i in range(20):
    filler()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(i, stream); launch(i + 1, stream); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 2)
print(f"C two launches per pair   : avg {sum(ts) / len(ts):.1f} us  min {min(ts):.1f}")
# D: four launches per pair
ts = []
for i in range(20):
    filler()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(4):
        launch(i + j, stream)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 4)
print(f"D four launches per pair  : avg {sum(ts) / len(ts):.1f} us  min {min(ts):.1f}")
# B
try:
    side = torch.cuda.Stream(dev)
    graphs = []
    for k in range(2):
        e0 = torch.cuda.Event(enable_timing=True, external=True)
        e1 = torch.cuda.Event(enable_timing=True, external=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            filler()
            e0.record(side)
            launch(k, side)
            e1.record(side)
        graphs.append((g, e0, e1))
    ts = []
    for i in range(20):
        g, e0, e1 = graphs[i & 1]
        g.replay()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"B events inside a graph   : avg {sum(ts) / len(ts):.1f} us  min {min(ts):.1f}")
except Exception as ex:                                   # noqa: BLE001
    print("B not available:", repr(ex)[:300])
