"""What the probability dropout costs the 8-head / head-32 bag self-attention kernels (row f3's encoder layers): forward and
forward + backward at M rows with p = 0 and p = 0.25 (the counter-hash mask is regenerated in all three kernels).
    python tools/gpu_time_sa_dropout.py [rows]"""
import os
import sys

import torch

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
from multimodal_path_omic_amd import ops  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
dev = torch.device("cuda:0")
heads, d = 8, 256
qkv = torch.randn(1, m, 3 * d, device=dev)
probe = torch.randn(1, m, d, device=dev)
for p in (0.0, 0.25):
    for mode in ("forward", "forward+backward"):
        x = qkv.clone().requires_grad_(mode != "forward")
        ts = []
        for it in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out, _ = ops.BagSelfAttentionFn.apply(x, heads, p, False)
            if mode != "forward":
                (out * probe).sum().backward()
                x.grad = None
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print(f"p={p:4.2f} {mode:17s}: {sorted(ts[1:])[len(ts[1:]) // 2]:7.3f} ms", flush=True)
