"""Diagnostic: PCIe-inclusive window rate -- WindowFeeder shipping 32 x 15000 x 1024 windows from host memory (fp32
source slides as the reference stores them, bf16 on the link) with and without the MCAT window step consuming them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_path_omic_amd.ingest import ArrayStore, WindowFeeder
dev = torch.device("cuda:0")
n, m, window = 128, 15000, 32
base = [torch.randn(m, 1024) for _ in range(8)]
for src_dtype in (torch.float32, torch.bfloat16):
    slides = [base[i % 8].to(src_dtype) for i in range(n)]             # 128 slides (8 distinct buffers)
    feeder = WindowFeeder(ArrayStore(slides), list(range(n)), window, dev, omics_of=lambda ids: [], labels_of=lambda ids: torch.zeros(len(ids)),
                          cens_of=lambda ids: torch.zeros(len(ids)), depth=2, workers=16)
    torch.cuda.synchronize(); t = time.perf_counter()
    rows = 0
    for bags, *_ in feeder:
        rows += bags.total_rows
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"source {src_dtype}: {n / dt:8.1f} slides/s ingest-only  ({rows * 2048 / dt / 1e9:.1f} GB/s on the link, bf16)", flush=True)
    feeder.close()
