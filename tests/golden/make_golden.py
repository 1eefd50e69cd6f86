#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (read-only mount at
/root/reference, torch CPU fp32) on the seeded cases of cases.py.

Run in the authoring container only:   python tests/golden/make_golden.py
The reference cannot travel to the GPU box; only these outputs do.  Import recipe
(SURVEY.md section 8(c)): model files are imported by bare name, and
models/utils.py:1 imports h5py at top level for helpers the path never calls, so an
empty module object named 'h5py' is registered first (no h5py function is provided
or used).  Weights are pushed into the reference modules with load_state_dict from
multimodal-path-omic_amd/synthetic.py, so fixtures hold seeds + outputs only.

Gradients are taken in eval mode (all dropout off; SURVEY.md section 7 hard part 5)
through a fixed linear probe of the outputs, and stored strided-subsampled.
"""
import os
import sys
import types
import warnings

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.nn as nn

warnings.filterwarnings("ignore")
torch.manual_seed(0)
torch.set_num_threads(8)

REF = "/root/reference"
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path[:0] = [REF, f"{REF}/models/mcat", f"{REF}/models/nacagat"]
from models.blocks import (AttentionNetGated, ContextualAttentionGate,   # noqa: E402
                           PreGatingContextualAttention)
from models.fusion import ConcatFusion                                    # noqa: E402
from models.loss import CrossEntropySurvivalLoss, CrossEntropySurvivalAttnRegLoss  # noqa: E402
from mcat import MultimodalCoAttentionTransformer                          # noqa: E402
from nacagat import NarrowContextualAttentionGateTransformer               # noqa: E402
sys.path.insert(0, f"{REF}/models/ge_nacagat")
from ge_nacagat import GeneExprNarrowContextualAttentionGateTransformer    # noqa: E402

import cases as C                                                          # noqa: E402
from multimodal_path_omic_amd import synthetic as syn                      # noqa: E402

sub = syn.subsample


def load(module, shapes, seed, gain=1.0, strip=""):
    sd = syn.fill_state_dict(shapes, seed, gain)
    module.load_state_dict({k[len(strip):]: v for k, v in sd.items()}, strict=True)
    return sd


def grads_of(loss, named):
    names = [n for n, _ in named]
    gs = torch.autograd.grad(loss, [t for _, t in named], allow_unused=True, retain_graph=True)
    return {n: (torch.zeros_like(t) if g is None else g) for n, g, (_, t) in zip(names, gs, named)}


def save(name, d):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in d.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB  ({len(d)} arrays)")


# ----------------------------------------------------------------------------- H3
def gen_coattn_mcat():
    out = {}
    for case, (m, gain, seed) in C.COATTN_CASES.items():
        mod = nn.MultiheadAttention(embed_dim=C.E, num_heads=1).eval()
        load(mod, C.MCAT_COATTN_SHAPES, seed, gain, strip="co_attention.")
        q, bag, p_out, p_a = C.coattn_inputs(m, seed + 1)
        q.requires_grad_(True)
        bag.requires_grad_(True)
        # training-style call: need_weights=False (models/mcat/mcat.py:97 with inference=False)
        o0, a0 = mod(query=q, key=bag, value=bag, need_weights=False)
        assert a0 is None
        named = [("query", q), ("bag", bag)] + [("co_attention." + n, p) for n, p in mod.named_parameters()]
        g0 = grads_of((o0 * p_out).sum(), named)
        # inference-style call with the map, and a gradient pushed into the map as well
        o1, a1 = mod(query=q, key=bag, value=bag, need_weights=True)
        g1 = grads_of((o1 * p_out).sum() + (a1 * p_a).sum(), named)
        out[f"{case}/out"] = o1
        out[f"{case}/out_noweights"] = o0
        out[f"{case}/A_sub"] = sub(a1)
        out[f"{case}/A_rowmax"] = a1.max(dim=1).values
        out[f"{case}/A_rowsum"] = a1.sum(dim=1)
        for n, g in g0.items():
            out[f"{case}/grad0/{n}"] = sub(g)
        for n, g in g1.items():
            out[f"{case}/grad1/{n}"] = sub(g)
    save("coattn_mcat", out)


# ----------------------------------------------------------------------------- H4 (+H5 inside)
def gen_coattn_nacagat():
    out = {}
    for case, (m, gain, seed) in C.NACAGAT_CASES.items():
        mod = PreGatingContextualAttention(embed_dim=C.E, num_heads=1).eval()
        load(mod, C.NACAGAT_COATTN_SHAPES, seed, gain, strip="co_attention.")
        q, bag, p_out, p_a = C.coattn_inputs(m, seed + 1)
        q.requires_grad_(True)
        bag.requires_grad_(True)
        o, a = mod(query=q, key=bag, value=bag)
        named = [("query", q), ("bag", bag)] + [("co_attention." + n, p) for n, p in mod.named_parameters()]
        g0 = grads_of((o * p_out).sum(), named)
        # 'cesar' loss back-propagates into the map (models/loss.py:97, models/nacagat/main.py:49-50)
        g1 = grads_of((o * p_out).sum() + (a * p_a).sum(), named)
        out[f"{case}/out"] = o
        out[f"{case}/A_sub"] = sub(a)
        out[f"{case}/A_rowmax"] = a.max(dim=1).values
        out[f"{case}/A_rowsum"] = a.sum(dim=1)
        for n, g in g0.items():
            out[f"{case}/grad0/{n}"] = sub(g)
        for n, g in g1.items():
            out[f"{case}/grad1/{n}"] = sub(g)
    save("coattn_nacagat", out)


# ----------------------------------------------------------------------------- H5
def gen_cag():
    mod = ContextualAttentionGate(dim=C.E, hidden_dim=C.E).eval()
    load(mod, C.CAG_SHAPES, 500, strip="co_attention.CAG.")
    q, qh, probe = C.cag_inputs()
    q.requires_grad_(True)
    qh.requires_grad_(True)
    c = mod(q, qh)
    named = [("Q", q), ("Q_hat", qh)] + [("co_attention.CAG." + n, p) for n, p in mod.named_parameters()]
    out = {"C": c}
    for n, g in grads_of((c * probe).sum(), named).items():
        out["grad/" + n] = g if g.numel() <= 4096 else sub(g)
    save("cag", out)


# ----------------------------------------------------------------------------- H6
def gen_encoder():
    layer = nn.TransformerEncoderLayer(d_model=C.E, nhead=8, dim_feedforward=512, dropout=0.25, activation="relu")
    mod = nn.TransformerEncoder(layer, num_layers=2).eval()
    load(mod, C.encoder_shapes("path_transformer"), 600, strip="path_transformer.")
    x, probe = C.encoder_inputs()
    x.requires_grad_(True)
    y = mod(x)
    named = [("x", x)] + [("path_transformer." + n, p) for n, p in mod.named_parameters()]
    out = {"y": y}
    for n, g in grads_of((y * probe).sum(), named).items():
        out["grad/" + n] = sub(g)
    save("encoder", out)


# ----------------------------------------------------------------------------- H7
def gen_pool():
    out = {}
    for case, (l, seed) in C.POOL_CASES.items():
        head = AttentionNetGated(n_classes=1, input_dim=C.E, hidden_dim=C.E).eval()
        rho = nn.Sequential(nn.Linear(C.E, C.E), nn.ReLU(), nn.Dropout(0.25)).eval()
        sd = syn.fill_state_dict(C.pool_shapes("path_attention_head", "path_rho"), seed)
        head.load_state_dict({k[len("path_attention_head."):]: v for k, v in sd.items() if k.startswith("path_attention_head.")})
        rho.load_state_dict({k[len("path_rho."):]: v for k, v in sd.items() if k.startswith("path_rho.")})
        x, probe_h, probe_a = C.pool_inputs(l, seed + 1)
        x.requires_grad_(True)
        # pooling idiom of models/mcat/mcat.py:105-109
        a, hx = head(x)
        a = torch.transpose(a, 1, 0)
        h = rho(torch.mm(torch.softmax(a, dim=1), hx)).squeeze()
        named = ([("x", x)] + [("path_attention_head." + n, p) for n, p in head.named_parameters()]
                 + [("path_rho." + n, p) for n, p in rho.named_parameters()])
        out[f"{case}/A"] = a
        out[f"{case}/h"] = h
        for n, g in grads_of((h * probe_h).sum() + (a * probe_a).sum(), named).items():
            out[f"{case}/grad/{n}"] = sub(g)
    save("pool", out)


# ----------------------------------------------------------------------------- H8
def gen_fusion():
    fus = ConcatFusion(dims=[C.E, C.E], hidden_size=C.E, output_size=C.E).eval()
    cls = nn.Linear(C.E, 4)
    sd = syn.fill_state_dict(C.FUSION_SHAPES, 700)
    fus.load_state_dict({k[len("fusion_layer."):]: v for k, v in sd.items() if k.startswith("fusion_layer.")})
    cls.load_state_dict({k[len("classifier."):]: v for k, v in sd.items() if k.startswith("classifier.")})
    hp, ho, probe = C.fusion_inputs()
    hp.requires_grad_(True)
    ho.requires_grad_(True)
    h = fus(hp, ho)
    logits = cls(h).unsqueeze(0)                              # models/mcat/mcat.py:126-138
    hazards = torch.sigmoid(logits)
    survs = torch.cumprod(1 - hazards, dim=1)
    y = torch.softmax(logits, dim=1)
    named = ([("h_path", hp), ("h_omic", ho)] + [("fusion_layer." + n, p) for n, p in fus.named_parameters()]
             + [("classifier." + n, p) for n, p in cls.named_parameters()])
    out = {"h": h, "hazards": hazards, "survs": survs, "Y": y}
    loss = (hazards * probe).sum() + (survs * probe.flip(1)).sum() + (y * probe * 0.5).sum()
    for n, g in grads_of(loss, named).items():
        out["grad/" + n] = sub(g)
    save("fusion", out)


def gen_fusion_next():
    """Row f4: the reference's BilinearFusion and GatedConcatFusion (eval mode) on the vectors of fusion_inputs.
    GatedConcatFusion keeps its gates in an unregistered list (models/fusion.py:25-27): their weights are pushed in
    directly; the fixture names them gates.<i>.0.* like the build's registered ModuleList."""
    from models.fusion import BilinearFusion, GatedConcatFusion
    hp, ho, _ = C.fusion_inputs()
    out = {}
    # bilinear, constructed as models/mcat/mcat.py:73-74
    fus = BilinearFusion(dim1=C.E, dim2=C.E, output_size=C.E).eval()
    assert {k: tuple(v.shape) for k, v in fus.state_dict().items()} == C.BILINEAR_SHAPES
    sd = syn.fill_state_dict(C.BILINEAR_SHAPES, 710, gain=3.0)
    fus.load_state_dict(sd, strict=True)
    a, b = hp.clone().requires_grad_(True), ho.clone().requires_grad_(True)
    y = fus(a, b)
    probe = syn.normal(syn.rng(711), tuple(y.shape))
    named = [("h_path", a), ("h_omic", b)] + list(fus.named_parameters())
    out["bilinear/out"] = y
    for n, g in grads_of((y * probe).sum(), named).items():
        out["bilinear/grad/" + n] = sub(g)
    # gated concat, constructed as models/mcat/mcat.py:75-77
    fus = GatedConcatFusion(dims=[C.E, C.E], hidden_size=C.E, output_size=C.E).eval()
    sd = syn.fill_state_dict(C.GATED_CONCAT_SHAPES, 720)
    fus.load_state_dict({k: v for k, v in sd.items() if not k.startswith("gates.")}, strict=True)
    for i, gate in enumerate(fus.gates):
        gate[0].weight.data.copy_(sd[f"gates.{i}.0.weight"])
        gate[0].bias.data.copy_(sd[f"gates.{i}.0.bias"])
    a, b = hp.clone().requires_grad_(True), ho.clone().requires_grad_(True)
    y = fus(a, b)
    named = ([("h_path", a), ("h_omic", b)] + list(fus.named_parameters())
             + [(f"gates.{i}.0.{n}", p) for i, gate in enumerate(fus.gates) for n, p in gate[0].named_parameters()])
    out["gated_concat/out"] = y
    for n, g in grads_of((y * probe).sum(), named).items():
        out["gated_concat/grad/" + n] = sub(g)
    save("fusion_next", out)


# ----------------------------------------------------------------------------- H1
def build_model(kind, omic_sizes, seed):
    cls = MultimodalCoAttentionTransformer if kind == "mcat" else NarrowContextualAttentionGateTransformer
    model = cls(omic_sizes=omic_sizes, model_size="medium", fusion="concat").eval()
    shapes = C.model_shapes(omic_sizes, kind == "nacagat")
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert ref_shapes == shapes, "cases.model_shapes drifted from the reference state_dict"
    assert list(ref_shapes) == list(shapes), "state_dict order drifted"
    model.load_state_dict(syn.fill_state_dict(shapes, seed), strict=True)
    return model


def gen_models():
    out = {}
    ces = CrossEntropySurvivalLoss()
    for case, (kind, m, omic_sizes, seed) in C.MODEL_CASES.items():
        model = build_model(kind, omic_sizes, seed)
        wsi, omics, label, censor = C.model_inputs(m, omic_sizes, seed + 1)
        kw = dict(inference=True) if kind == "mcat" else {}
        hz, sv, y, att = model(wsi=wsi, omics=omics, **kw)
        # DataLoader convention (1,M,1024) / [(1,d_i)] gives the same result
        hz_b, sv_b, y_b, att_b = model(wsi=wsi.unsqueeze(0), omics=[o.unsqueeze(0) for o in omics], **kw)
        assert torch.equal(hz, hz_b) and torch.equal(att["path"], att_b["path"])
        if kind == "mcat":
            hz_t, _, _, att_t = model(wsi=wsi, omics=omics)     # training-style: no map
            assert att_t["coattn"] is None and torch.allclose(hz, hz_t, atol=1e-6)
        loss = ces(hz, sv, label, c=censor)
        named = list(model.named_parameters())
        gs = grads_of(loss, named)
        out[f"{case}/hazards"], out[f"{case}/survs"], out[f"{case}/Y"] = hz, sv, y
        out[f"{case}/A_path"], out[f"{case}/A_omic"] = att["path"], att["omic"]
        out[f"{case}/A_coattn_sub"] = sub(att["coattn"])
        out[f"{case}/A_coattn_rowmax"] = att["coattn"].max(dim=1).values
        out[f"{case}/loss"] = loss
        for n, g in gs.items():
            out[f"{case}/grad/{n}"] = sub(g, 256)
    save("models", out)


# ----------------------------------------------------------------------------- f3
def gen_ge_models():
    """models/ge_nacagat/ge_nacagat.py:43-72 in eval mode + the loss of models/ge_nacagat/main.py:33 (CrossEntropyLoss on
    the soft-maxed Y) and its parameter gradients."""
    out = {}
    ce = nn.CrossEntropyLoss()
    for case, (m, seed) in C.GE_MODEL_CASES.items():
        model = GeneExprNarrowContextualAttentionGateTransformer(model_size="medium").eval()
        shapes = C.ge_model_shapes()
        ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        assert ref_shapes == shapes and list(ref_shapes) == list(shapes), "cases.ge_model_shapes drifted from the reference"
        model.load_state_dict(syn.fill_state_dict(shapes, seed), strict=True)
        wsi, target = C.ge_model_inputs(m, seed + 1)
        y, att = model(wsi=wsi)
        assert y.shape == (3,) and att["attn"].shape == (m, m) and att["path"].shape == (1, m)
        loss = ce(y.unsqueeze(0), target)
        gs = grads_of(loss, list(model.named_parameters()))
        out[f"{case}/Y"], out[f"{case}/A_path"], out[f"{case}/loss"] = y, att["path"], loss
        out[f"{case}/A_attn_sub"] = sub(att["attn"])
        out[f"{case}/A_attn_rowmax"] = att["attn"].max(dim=1).values
        out[f"{case}/A_attn_diag"] = att["attn"].diagonal()
        for n, g in gs.items():
            out[f"{case}/grad/{n}"] = sub(g, 256)
    save("ge_models", out)


# ----------------------------------------------------------------------------- H9
def gen_loss():
    ces = CrossEntropySurvivalLoss()
    cesar = CrossEntropySurvivalAttnRegLoss()
    out = {}
    g = syn.rng(801)
    hz = torch.sigmoid(syn.normal(g, (8, 1, 4)))
    for i in range(8):
        sv = torch.cumprod(1 - hz[i], dim=1)
        y = torch.tensor([i % 4])
        c = torch.tensor([float(i // 4)])
        out[f"ces/{i}"] = ces(hz[i], sv, y, c)
        att = syn.normal(g, (6, 50))
        l, al = cesar(hz[i], sv, y, c, att)
        out[f"cesar/{i}"] = torch.stack([l, al])
    out["hazards"] = hz
    save("loss", out)


# ----------------------------------------------------------------------------- cohort
def gen_cohort():
    """The reference train()/validate() loop (models/mcat/main.py:19-155) restated around the
    REFERENCE model and loss, with every dropout off (model.eval()), fixed slide order and a
    fixed 80/20 split: per-slide risks, mean loss per epoch, on a seeded synthetic cohort."""
    cfg = C.COHORT
    slides = syn.make_cohort(cfg["n_slides"], cfg["m_lo"], cfg["m_hi"], cfg["omic_sizes"], cfg["seed"])
    n_train = int(cfg["train_frac"] * len(slides))
    out = {}
    for kind in ("mcat", "nacagat"):
        model = build_model(kind, cfg["omic_sizes"], cfg["weight_seed"])
        model.eval()
        opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        ces = CrossEntropySurvivalLoss()
        for epoch in range(cfg["epochs"]):
            risks, losses = [], []
            for i, s in enumerate(slides[:n_train]):
                hz, sv, _, _ = model(wsi=s["wsi"].unsqueeze(0), omics=[o.unsqueeze(0) for o in s["omics"]])
                loss = ces(hz, sv, torch.tensor([s["survival_class"]]), c=torch.tensor([float(s["censorship"])]))
                losses.append(loss.item())
                risks.append(-torch.sum(sv, dim=1).item())
                (loss / cfg["grad_acc_step"]).backward()
                if (i + 1) % cfg["grad_acc_step"] == 0:
                    opt.step()
                    opt.zero_grad()
            out[f"{kind}/train_risk/{epoch}"] = np.array(risks)
            out[f"{kind}/train_loss/{epoch}"] = np.array(losses)
            vr = []
            with torch.no_grad():
                for s in slides[n_train:]:
                    _, sv, _, _ = model(wsi=s["wsi"].unsqueeze(0), omics=[o.unsqueeze(0) for o in s["omics"]])
                    vr.append(-torch.sum(sv, dim=1).item())
            out[f"{kind}/val_risk/{epoch}"] = np.array(vr)
    save("cohort", out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["coattn_mcat", "coattn_nacagat", "cag", "encoder", "pool", "fusion",
                             "models", "loss", "cohort", "fusion_next", "ge_models"]
    for w in which:
        globals()["gen_" + w]()
