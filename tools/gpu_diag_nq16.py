"""Diagnostic: NaCAGaT co-attention module with 15 / 16 queries vs the oracle."""
import sys, torch
sys.path[:0] = [".", "tests", "tests/golden"]
import cases as C
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.blocks import PreGatingContextualAttention
from multimodal_path_omic_amd.ops import BagBatch
from oracle import mpo_oracle as O
dev = torch.device("cuda:0")
for n_q in (15, 16):
    for lengths in ([300, 500], [1, 2, 31, 32, 33, 65]):
        sd = syn.fill_state_dict(C.NACAGAT_COATTN_SHAPES, 5)
        mod = PreGatingContextualAttention(embed_dim=C.E, num_heads=1)
        mod.load_state_dict({k[len("co_attention."):]: v for k, v in sd.items()})
        mod.to(dev).eval()
        g = syn.rng(9)
        qs = [syn.normal(g, (n_q, C.E)) for _ in lengths]
        bags = [torch.relu(syn.normal(g, (m, C.E))) for m in lengths]
        pouts = [syn.normal(g, (n_q, C.E)) for _ in lengths]
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        outs_o = []
        for q, b, po in zip(qs, bags, pouts):
            out_o, a_o = O.pregating_contextual_attention(q, b, p)
            outs_o.append(out_o)
            (out_o * po).sum().backward()
        batch = BagBatch.from_list([b.to(dev) for b in bags])
        qd = torch.stack(qs).to(dev)
        out, maps = mod.forward_window(qd, batch)
        (out * torch.stack(pouts).to(dev)).sum().backward()
        e_out = max(float((out[i].cpu() - outs_o[i]).abs().max() / outs_o[i].abs().max()) for i in range(len(lengths)))
        errs = []
        for n, prm in mod.named_parameters():
            ref = p["co_attention." + n].grad
            errs.append((float((prm.grad.cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)), n))
        errs.sort(reverse=True)
        print(n_q, lengths, "out", f"{e_out:.1e}", [(f"{e:.1e}", n) for e, n in errs[:3]], flush=True)
