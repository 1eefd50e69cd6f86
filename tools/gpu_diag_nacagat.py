"""Diagnostic (not a test): per-output / per-gradient relative error of K2 against the oracle for every fixture."""
import sys
import torch
sys.path[:0] = [".", "tests", "tests/golden"]
import cases as C
from test_gpu_coattn_nacagat import make_module, oracle_grads, relerr
from oracle import mpo_oracle as O

dev = torch.device("cuda:0")
for case, (m, gain, seed) in C.NACAGAT_CASES.items():
    for dtype in (torch.float32, torch.bfloat16):
        mod, p = make_module(seed, gain, dev)
        mod.eval()
        q, bag, p_out, p_a = C.coattn_inputs(m, seed + 1)
        bag_in = bag.to(dtype)
        qo = q.clone().requires_grad_(True)
        bo = bag_in.float().clone().requires_grad_(True)
        out_o, a_o = O.pregating_contextual_attention(qo, bo, p)
        named = [("query", qo), ("bag", bo)] + list(p.items())
        g_o = oracle_grads((out_o * p_out).sum() + (a_o * p_a).sum(), named)
        qd = q.to(dev).requires_grad_(True)
        bd = bag_in.to(dev).requires_grad_(True)
        out, a = mod(query=qd, key=bd, value=bd)
        rel_a = ((a.detach().cpu() - a_o.detach()).abs() / a_o.detach().clamp_min(1e-30)).max().item()
        params = dict(mod.named_parameters())
        tensors = [qd, bd] + [params[k[len("co_attention."):]] for k in p]
        gs = torch.autograd.grad((out * p_out.to(dev)).sum() + (a * p_a.to(dev)).sum(), tensors)
        line = f"{case:12s} {str(dtype)[6:]:9s} out {relerr(out, out_o):.2e} A {rel_a:.2e} |"
        for (n, _), gr in zip(named, gs):
            line += f" {n.replace('co_attention.', '')[:18]} {relerr(gr, g_o[n]):.1e}"
        print(line, flush=True)
