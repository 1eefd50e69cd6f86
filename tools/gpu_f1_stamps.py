"""In-kernel phase times of the patch-layer kernel (variant build -DMPO_PF_STAMPS: s_memtime laps summed per stage kind in
wave 0 (leading stream) and wave 4 (lagging stream) of every workgroup)."""
import ctypes, os, sys
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
batch = BagBatch(xs[0], cu, lengths)
plan = batch.plan()
w = torch.randn(E, 1024, device=dev) / 32
wb = torch.empty(E, 1024, device=dev, dtype=torch.bfloat16)
stream = torch.cuda.current_stream(dev)
L.check(lib.mpo_pack_patch_weight(L.ptr(w), L.ptr(wb), E, 1024, stream.cuda_stream), "pack")
bias = torch.randn(E, device=dev) * 0.1
h_out = torch.empty(window * patches, E, device=dev, dtype=torch.bfloat16)
for i in range(6):
    L.check(lib.mpo_patch_coattn_fwd_bagpass(L.ptr(xs[i & 1]), L.ptr(wb), L.ptr(bias), L.ptr(cu), window, None, L.ptr(h_out),
                                             None, None, 6, patches, 0.25, 1, 0, plan, stream.cuda_stream), "bagpass")
torch.cuda.synchronize()
raw = ctypes.CDLL(L.lib()._name)
host = (ctypes.c_float * (1024 * 16))()
assert raw.mpo_debug_patch_fc_stamps(host) == 0
pm = torch.tensor(list(host)).view(1024, 16)[:256]
names = ["main stage work", "epilogue stage work", "first / last stage work", "idle stage work", "wait own pieces (vmcnt)", "barrier"]
for wv, off in (("leading stream, wave 0", 0), ("lagging stream, wave 4", 8)):
    tot = pm[:, off:off + 6].sum(1)
    print(f"{wv}: total k-cycles per workgroup: mean {tot.mean():.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
    for i, n in enumerate(names):
        c = pm[:, off + i]
        print(f"   {n:28s} {c.mean():8.1f} k-cycles ({100 * c.mean() / tot.mean():5.1f} %)")
