"""Diagnostic: K1 forward/backward bag-pass time vs workgroup target (env MPO_COATTN_TARGET_WGS is read once
per process, so this script is run once per setting)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd import _lib as L
from multimodal_path_omic_amd.ops import make_cu
dev = torch.device("cuda:0")
torch.manual_seed(0)
E, n_q = 256, 6
lib = L.lib()
def run(B, M, dt, reps=20):
    cu = make_cu([M] * B, dev)
    esz = 2 if dt == torch.bfloat16 else 4
    bags = [torch.relu(torch.randn(B * M, E, device=dev)).to(dt) for _ in range(2)]
    dbag = torch.empty_like(bags[0])
    qk2 = torch.randn(B * n_q, E, device=dev) * 0.05
    dctx = torch.randn(B * n_q, E, device=dev) * 0.05
    splits = lib.mpo_coattn_target_workgroups() // B + 1
    part_ml = torch.empty(B * splits * 32, device=dev); part_ctx = torch.empty(B * splits * n_q * E, device=dev)
    lse2 = torch.full((B * n_q,), 14.0, device=dev); delta = torch.zeros(B * n_q, device=dev)
    st = torch.cuda.current_stream()
    code = L.bag_dtype_code(bags[0])
    def fwd(i): L.check(lib.mpo_coattn_fwd_bagpass(L.ptr(bags[i & 1]), code, L.ptr(cu), B, E, L.ptr(qk2), L.ptr(part_ml), L.ptr(part_ctx), None, n_q, M, None, st.cuda_stream), "f")
    def bwd(i): L.check(lib.mpo_coattn_bwd_bagpass(L.ptr(bags[i & 1]), code, L.ptr(cu), B, E, L.ptr(qk2), L.ptr(lse2), L.ptr(dctx), L.ptr(delta), None, L.ptr(dbag), L.ptr(part_ctx), n_q, M, None, st.cuda_stream), "b")
    out = []
    for fn, mult in ((fwd, 1), (bwd, 2)):
        for i in range(3): fn(i)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for i, (s, e) in enumerate(ev):
            s.record(st); fn(i); e.record(st)
        torch.cuda.synchronize()
        us = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)
        med = us[len(us) // 2]
        out.append((med, mult * B * M * E * esz / med / 1e6))
    print(f"target={os.environ.get('MPO_COATTN_TARGET_WGS','1024')} B={B} M={M} {str(dt)[6:]} splits={splits}: "
          f"fwd {out[0][0]:.1f} us {out[0][1]:.2f} TB/s | bwd {out[1][0]:.1f} us {out[1][1]:.2f} TB/s")
run(32, 15000, torch.bfloat16)
run(1, 100000, torch.float32)
run(1, 15000, torch.bfloat16)
