"""Diagnostic: per-parameter gradient errors of MCAT 'big' against the oracle."""
import sys, torch
sys.path[:0] = [".", "tests", "tests/golden"]
import cases as C
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.harness import ces_loss
from multimodal_path_omic_amd.models import MultimodalCoAttentionTransformer
from oracle import mpo_oracle as O
dev = torch.device("cuda:0")
for size in ("medium", "big"):
    for dtype in (torch.float32,):
        omic_sizes, m, seed = [64, 100, 256, 31, 8, 300], 1200, 6160
        model = MultimodalCoAttentionTransformer(omic_sizes=omic_sizes, model_size=size, bag_dtype=dtype)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = syn.fill_state_dict(shapes, seed)
        model.load_state_dict(sd, strict=True)
        model.to(dev).eval()
        wsi, omics, _, _ = C.model_inputs(m, omic_sizes, seed + 1)
        hz, sv, y, att = model(wsi=wsi.to(dev), omics=[o.to(dev) for o in omics], inference=True)
        label, censor = torch.tensor([3]), torch.tensor([0.0])
        ces_loss(hz, sv, label.to(dev), censor.to(dev)).backward()
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        hz_o, sv_o, _, att_o = O.mcat_forward(p, wsi, omics, inference=True)
        O.ces_loss(hz_o, sv_o, label, censor).backward()
        a, a_o = att["coattn"].cpu(), att_o["coattn"].detach()
        print(size, "hz", float((hz.cpu() - hz_o).abs().max()), "map rel", ((a - a_o).abs() / a_o.clamp_min(1e-30)).max().item(), "map max", a_o.max().item())
        worst = []
        for n, prm in model.named_parameters():
            ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
            scale = max(float(ref.abs().max()), 1e-4)
            worst.append((float((prm.grad.cpu() - ref).abs().max()) / scale, n, scale))
        worst.sort(reverse=True)
        for e, n, sc in worst[:8]:
            print(f"   {n:45s} err {e:.2e} scale {sc:.2e}")
