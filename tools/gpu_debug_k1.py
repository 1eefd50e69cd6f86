"""Diagnostic (not a test): print error metrics of K1 fwd/bwd for several cases in one go."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import torch
import cases as C
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.blocks import CoAttention
from multimodal_path_omic_amd.ops import linear
from oracle import mpo_oracle as O

dev = torch.device("cuda:0")
def relerr(a, b):
    a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

g = syn.rng(5)
x = syn.normal(g, (6, 256)).to(dev); w = syn.normal(g, (256, 256), 0.1).to(dev); b = syn.normal(g, (256,)).to(dev)
y = linear(x, w, b)
print("linear fwd relerr", relerr(y, x.double() @ w.double().t() + b.double()))

for case in ["m256", "m777_ragged", "m2000_peaky", "m15000"]:
    for dtype in (torch.float32, torch.bfloat16):
        m, gain, seed = C.COATTN_CASES[case]
        sd = syn.fill_state_dict(C.MCAT_COATTN_SHAPES, seed, gain)
        mod = CoAttention(C.E, 1); mod.load_state_dict({k[len("co_attention."):]: v for k, v in sd.items()}); mod.to(dev)
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        q, bag, p_out, p_a = C.coattn_inputs(m, seed + 1)
        bag_in = bag.to(dtype)
        qo = q.clone().requires_grad_(True); bo = bag_in.float().clone().requires_grad_(True)
        out_o, a_o = O.mcat_coattention(qo, bo, p, need_weights=True)
        named = [("query", qo), ("bag", bo)] + list(p.items())
        go = torch.autograd.grad((out_o * p_out).sum() + (a_o * p_a).sum(), [t for _, t in named], allow_unused=True)
        qd = q.to(dev).requires_grad_(True); bd = bag_in.to(dev).requires_grad_(True)
        try:
            out1, a1 = mod(query=qd, key=bd, value=bd, need_weights=True)
            torch.cuda.synchronize()
            rel_a = ((a1.detach().cpu() - a_o.detach()).abs() / a_o.detach().clamp_min(1e-30)).max().item()
            print(f"{case} {dtype}: out {relerr(out1, out_o):.2e}  A rel {rel_a:.2e}  rowsum {a1.sum(1).cpu().tolist()[:2]}")
            params = dict(mod.named_parameters())
            tensors = [qd, bd] + [params[k[len('co_attention.'):]] for k in p]
            gs = torch.autograd.grad((out1 * p_out.to(dev)).sum() + (a1 * p_a.to(dev)).sum(), tensors)
            torch.cuda.synchronize()
            for (n, _), gr, ref in zip(named, gs, go):
                if ref is None: ref = torch.zeros_like(gr).cpu()
                print(f"    d{n}: {relerr(gr, ref):.2e}  (|ref| {float(ref.abs().max()):.2e}, |got| {float(gr.abs().max()):.2e})")
        except Exception as e:
            print(case, dtype, "EXC", repr(e)[:300])
