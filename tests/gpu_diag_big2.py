"""Diagnostic: K1 at E=512, d_bag error per row / column against the oracle."""
import sys, torch
sys.path[:0] = [".", "tests", "tests/golden"]
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.blocks import CoAttention
from oracle import mpo_oracle as O
dev = torch.device("cuda:0")
E, M = 512, 1200
shapes = {"co_attention.in_proj_weight": (3 * E, E), "co_attention.in_proj_bias": (3 * E,),
          "co_attention.out_proj.weight": (E, E), "co_attention.out_proj.bias": (E,)}
sd = syn.fill_state_dict(shapes, 77)
mod = CoAttention(E, 1)
mod.load_state_dict({k[len("co_attention."):]: v for k, v in sd.items()})
mod.to(dev)
g = syn.rng(5)
q = syn.normal(g, (6, E)); bag = torch.relu(syn.normal(g, (M, E))); pout = syn.normal(g, (6, E))
for dtype in (torch.float32, torch.bfloat16):
    b_in = bag.to(dtype)
    qo = q.clone().requires_grad_(True); bo = b_in.float().clone().requires_grad_(True)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_o, _ = O.mcat_coattention(qo, bo, p, need_weights=False)
    (out_o * pout).sum().backward()
    qd = q.to(dev).requires_grad_(True); bd = b_in.to(dev).requires_grad_(True)
    out, _ = mod(query=qd, key=bd, value=bd, need_weights=False)
    (out * pout.to(dev)).sum().backward()
    err = (bd.grad.float().cpu() - bo.grad).abs()
    scale = bo.grad.abs().max()
    rows = (err.max(1).values / scale)
    cols = (err.max(0).values / scale)
    bad_r = torch.nonzero(rows > 5e-3 if dtype == torch.float32 else rows > 3e-2).flatten()
    bad_c = torch.nonzero(cols > 5e-3 if dtype == torch.float32 else cols > 3e-2).flatten()
    print(dtype, "out err", float((out.cpu() - out_o).abs().max() / out_o.abs().max()), "dbag max rel", float(rows.max()),
          "bad rows", bad_r[:20].tolist(), len(bad_r), "bad cols", bad_c[:20].tolist(), len(bad_c), flush=True)
