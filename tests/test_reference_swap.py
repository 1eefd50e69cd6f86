"""INTEGRATION.md section 3 against the REAL reference classes: every drop-in module strict-loads the state_dict of
the reference slot it replaces and can be assigned into the reference model, whose full state_dict (names AND
order) is unchanged by the swap -- i.e. checkpoints written by the reference's main.py (models/mcat/main.py:95-100)
keep loading.  CPU only (constructors and state_dicts; running the swapped model needs the GPU and is covered by the
golden-vector tests).  Skipped where /root/reference does not exist (the GPU box).  Runs in a child interpreter: the
reference is imported by bare module names (`models`, `mcat`, `nacagat`) that must not leak into this session."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, types, warnings
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
REF, ROOT = sys.argv[1], sys.argv[2]
# import recipe of SURVEY.md 8(c): models/utils.py:1 imports h5py for helpers the path never calls
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path[:0] = [ROOT, REF, REF + "/models/mcat", REF + "/models/nacagat"]
import torch
import mcat, nacagat
from multimodal_path_omic_amd.blocks import (AttentionNetGated, CoAttention, ContextualAttentionGate,
                                             PreGatingContextualAttention)
from multimodal_path_omic_amd.fusion import BilinearFusion, ConcatFusion, GatedConcatFusion
from multimodal_path_omic_amd.transformer import make_set_transformer
from multimodal_path_omic_amd import models as ours

sizes = [100, 200, 300, 400, 500, 600]            # the reference's own test widths (models/mcat/mcat.py:152)
d = 256

def swap(ref, slot, new):
    new.load_state_dict(getattr(ref, slot).state_dict(), strict=True)
    setattr(ref, slot, new)

for kind in ("mcat", "nacagat"):
    for fusion in ("concat", "bilinear", "gated_concat"):
        torch.manual_seed(0)
        if kind == "mcat":
            ref = mcat.MultimodalCoAttentionTransformer(omic_sizes=sizes, fusion=fusion)
        else:
            ref = nacagat.NarrowContextualAttentionGateTransformer(omic_sizes=sizes, fusion=fusion)
        before = {k: v.clone() for k, v in ref.state_dict().items()}
        swap(ref, "co_attention", CoAttention(d, 1) if kind == "mcat" else PreGatingContextualAttention(d, 1))
        swap(ref, "path_transformer", make_set_transformer(d, dropout=0.25))
        swap(ref, "omic_transformer", make_set_transformer(d, dropout=0.25))
        swap(ref, "path_attention_head", AttentionNetGated(n_classes=1, input_dim=d, hidden_dim=d))
        swap(ref, "omic_attention_head", AttentionNetGated(n_classes=1, input_dim=d, hidden_dim=d))
        if fusion == "concat":
            swap(ref, "fusion_layer", ConcatFusion(dims=[d, d], hidden_size=d, output_size=d))
        elif fusion == "bilinear":
            swap(ref, "fusion_layer", BilinearFusion(dim1=d, dim2=d, output_size=d))
        else:
            # the reference keeps this layer's gates in a plain list (models/fusion.py:25-27): they are absent from its
            # state_dict, so a strict load of the drop-in (which registers them) is not possible by construction
            new = GatedConcatFusion(dims=[d, d], hidden_size=d, output_size=d)
            missing, unexpected = new.load_state_dict(ref.fusion_layer.state_dict(), strict=False)
            assert not unexpected and all(k.startswith("gates.") for k in missing), (missing, unexpected)
            ref.fusion_layer = new
        if kind == "nacagat":                      # the CAG slot inside the (already swapped) co-attention
            cag = ContextualAttentionGate(dim=d, hidden_dim=d)
            cag.load_state_dict(ref.co_attention.CAG.state_dict(), strict=True)
            ref.co_attention.CAG = cag
        after = ref.state_dict()
        keys_after = [k for k in after if not k.startswith("fusion_layer.gates.")]
        assert keys_after == list(before), (kind, fusion, [k for k in keys_after if k not in before][:5])
        for k in before:
            assert torch.equal(before[k], after[k]), k
        # and the whole-model drop-in takes the reference checkpoint as it is
        cls = ours.MultimodalCoAttentionTransformer if kind == "mcat" else ours.NarrowContextualAttentionGateTransformer
        whole = cls(omic_sizes=sizes, fusion=fusion)
        res = whole.load_state_dict(before, strict=(fusion != "gated_concat"))
        assert not res.unexpected_keys and all(k.startswith("fusion_layer.gates.") for k in res.missing_keys)
        print("ok", kind, fusion, len(before))

# row f3: the gene-expression model (models/ge_nacagat/ge_nacagat.py) -- its slots and the whole model
sys.path.insert(0, REF + "/models/ge_nacagat")
import ge_nacagat
for size, dd in (("small", 128), ("medium", 256), ("big", 512)):
    torch.manual_seed(0)
    ref = ge_nacagat.GeneExprNarrowContextualAttentionGateTransformer(model_size=size)
    before = {k: v.clone() for k, v in ref.state_dict().items()}
    swap(ref, "path_transformer", make_set_transformer(dd, dropout=0.25))
    swap(ref, "path_attention_head", AttentionNetGated(n_classes=1, input_dim=dd, hidden_dim=dd))
    after = ref.state_dict()
    assert list(after) == list(before) and all(torch.equal(before[k], after[k]) for k in before)
    whole = ours.GeneExprNarrowContextualAttentionGateTransformer(model_size=size)
    whole.load_state_dict(before, strict=True)
    assert list(whole.state_dict()) == list(before)
    print("ok", "ge_nacagat", size, len(before))
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is not on this machine")
@pytest.mark.timeout(300)
def test_every_slot_swaps_into_the_reference_models():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT, REF, ROOT], capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok ") == 9, r.stdout
