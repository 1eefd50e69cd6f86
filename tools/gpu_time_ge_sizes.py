"""Step time of the gene-expression model (row f3) per model size at one bag length; the 'big' size runs the one-head
attention at head dimension 512 on the fp32 kernels (dK / dV in two column passes): functional, this says what it costs.
    python tools/gpu_time_ge_sizes.py [patches]"""
import os
import sys
import time

import torch

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
from multimodal_path_omic_amd import synthetic as syn  # noqa: E402
from multimodal_path_omic_amd.models import GeneExprNarrowContextualAttentionGateTransformer  # noqa: E402

patches = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
dev = torch.device("cuda:0")
for size in ("small", "medium", "big"):
    torch.manual_seed(0)
    model = GeneExprNarrowContextualAttentionGateTransformer(model_size=size, bag_dtype=torch.bfloat16).to(dev).train()
    wsi = syn.make_bag(patches, 77).to(dev).to(torch.bfloat16)
    target = torch.tensor([1], device=dev)

    def step():
        model.zero_grad(set_to_none=True)
        y, _ = model(wsi=wsi)
        torch.nn.functional.cross_entropy(y.unsqueeze(0), target).backward()
        return y
    for _ in range(2):
        y = step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(3):
        y = step()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    finite = all(torch.isfinite(p.grad).all().item() for p in model.parameters())
    print(f"{size:6s} {patches} rows: {ms:8.2f} ms per step, Y {[round(v, 4) for v in y.tolist()]}, gradients finite {finite}", flush=True)
    del model, wsi
