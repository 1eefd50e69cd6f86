"""Diagnostic: whole MCAT 'big' fp32 -- gradient arriving at H_bag vs the oracle's, per row."""
import sys, torch
sys.path[:0] = [".", "tests", "tests/golden"]
import cases as C
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.harness import ces_loss
from multimodal_path_omic_amd.models import MultimodalCoAttentionTransformer
from oracle import mpo_oracle as O
dev = torch.device("cuda:0")
omic_sizes, m, seed = [64, 100, 256, 31, 8, 300], 1200, 6160
model = MultimodalCoAttentionTransformer(omic_sizes=omic_sizes, model_size="big", bag_dtype=torch.float32)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
sd = syn.fill_state_dict(shapes, seed)
model.load_state_dict(sd, strict=True)
model.to(dev).eval()
wsi, omics, _, _ = C.model_inputs(m, omic_sizes, seed + 1)
keep = {}
orig = model._patch_fc
def patched(bags):
    out = orig(bags)
    out.data.retain_grad(); keep["h"] = out.data
    return out
model._patch_fc = patched
hz, sv, y, att = model(wsi=wsi.to(dev), omics=[o.to(dev) for o in omics], inference=True)
label, censor = torch.tensor([3]), torch.tensor([0.0])
ces_loss(hz, sv, label.to(dev), censor.to(dev)).backward()
p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
h_o = O.patch_fc(wsi, p); h_o.retain_grad()
g_o = O.omic_fc(omics, p)
h_co, a_co = O.mcat_coattention(g_o, h_o, p, need_weights=True)
hz_o, sv_o, _, _ = O._tail(h_co, g_o, a_co, p)
O.ces_loss(hz_o, sv_o, label, censor).backward()
gh, gh_o = keep["h"].grad.cpu(), h_o.grad
print("h fwd err", float((keep["h"].detach().cpu() - h_o.detach()).abs().max()))
err = (gh - gh_o).abs(); scale = gh_o.abs().max()
rows = err.max(1).values / scale
print("dH max rel", float(rows.max()), "rows > 1e-3:", torch.nonzero(rows > 1e-3).flatten()[:30].tolist(), int((rows > 1e-3).sum()))
cols = err.max(0).values / scale
print("cols > 1e-3:", torch.nonzero(cols > 1e-3).flatten()[:30].tolist(), int((cols > 1e-3).sum()))
for n in ("co_attention.in_proj_weight", "co_attention.out_proj.weight"):
    prm = dict(model.named_parameters())[n]
    print(n, float((prm.grad.cpu() - p[n].grad).abs().max() / p[n].grad.abs().max()))
