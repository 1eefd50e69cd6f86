"""Workload for rocprofv3: K1 fwd+bwd on a window of 32 x 15k bf16 bags, and 1 x 100k fp32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from multimodal_path_omic_amd.blocks import CoAttention
from multimodal_path_omic_amd.ops import BagBatch
dev = torch.device("cuda:0")
torch.manual_seed(0)
mod = CoAttention(256, 1).to(dev)
for (B, M, dt) in [(32, 15000, torch.bfloat16), (1, 100000, torch.float32), (1, 15000, torch.bfloat16)]:
    bags = [torch.relu(torch.randn(M, 256, device=dev)).to(dt) for _ in range(B)]
    batch = BagBatch.from_list(bags); del bags
    data = batch.data.requires_grad_(True)
    q = torch.randn(B, 6, 256, device=dev, requires_grad=True)
    b2 = batch.with_data(data)
    for _ in range(10):
        out, _ = mod.forward_window(q, b2, False)
        out.sum().backward()
    torch.cuda.synchronize()
