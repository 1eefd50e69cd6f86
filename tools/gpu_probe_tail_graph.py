"""Probe: per-kernel cost of the token tail's forward when captured ALONE into a HIP graph (no bag kernels around it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.nn as nn
from multimodal_path_omic_amd import ops
from multimodal_path_omic_amd.blocks import AttentionNetGated
from multimodal_path_omic_amd.transformer import make_set_transformer
dev = torch.device("cuda:0")
torch.manual_seed(0)
b, l, d = 32, 6, 256
enc = [make_set_transformer(d, dropout=0.25).to(dev).train() for _ in range(2)]
heads = [AttentionNetGated(n_classes=1, input_dim=d, hidden_dim=d).to(dev).train() for _ in range(2)]
rhos = [nn.Sequential(nn.Linear(d, d), nn.ReLU(), nn.Dropout(0.25)).to(dev).train() for _ in range(2)]
x = torch.randn(2, b, l, d, device=dev)
ops.set_rng_epoch(torch.zeros(1, dtype=torch.int64, device=dev))
def fwd():
    with torch.no_grad():
        tok = ops.encoder_stacked(x, [list(e.layers) for e in enc], True)
        return ops.gated_pool_stacked(tok, heads, rhos, True, interleave=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fwd(); fwd()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd()
for _ in range(5): g.replay()
torch.cuda.synchronize(); t = time.perf_counter()
n = 50
for _ in range(n): g.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(f"encoder (2 layers, 2 branches) + pooling forward alone: {dt * 1e6:.1f} us per replay = 19 kernels -> {dt * 1e6 / 19:.2f} us per kernel")
