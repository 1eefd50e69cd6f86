#!/usr/bin/env python3
"""Row f3 timing: the gene-expression model (models.GeneExprNarrowContextualAttentionGateTransformer) at M rows, one slide
per step as in the reference's training loop (models/ge_nacagat/main.py:25-52), forward + loss + backward.
    python tools/gpu_time_ge.py [M] [steps] [eval|train]
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_path_omic_amd import synthetic as syn                      # noqa: E402
from multimodal_path_omic_amd.models import GeneExprNarrowContextualAttentionGateTransformer  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mode = sys.argv[3] if len(sys.argv) > 3 else "train"
dev = torch.device("cuda:0")
model = GeneExprNarrowContextualAttentionGateTransformer(bag_dtype=torch.bfloat16).to(dev)
model.train(mode == "train")
wsi = syn.make_bag(m, 1).to(dev).to(torch.bfloat16)
target = torch.tensor([1], device=dev)


def step():
    y, att = model(wsi=wsi)
    torch.nn.functional.cross_entropy(y.unsqueeze(0), target).backward()


def fwd_only():
    with torch.no_grad():
        model(wsi=wsi)


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
for _ in range(2):
    fwd_only()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    fwd_only()
torch.cuda.synchronize()
df = (time.perf_counter() - t0) / steps
print(f"ge_nacagat medium, M={m}, {mode}: fwd+bwd {dt * 1e3:.2f} ms/slide ({1 / dt:.1f} slides/s), forward alone {df * 1e3:.2f} ms", flush=True)
