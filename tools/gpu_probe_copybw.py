"""Probe: what a read-once + write-once pass reaches on this part -- the ceiling K1's backward (reads H_bag, writes dH: 491.5 MB per
32 x 15 000-row window) and K2's patch-side gradient are judged against.  torch copies of a 245.76 MB bf16 tensor into another,
two source / destination pairs alternated (working set 983 MB >> the 256 MB Infinity Cache), HIP events."""
import torch

dev = torch.device("cuda:0")
rows = 32 * 15000
src = [torch.randn(rows, 256, device=dev).to(torch.bfloat16) for _ in range(2)]
dst = [torch.empty_like(s) for s in src]
for i in range(6):
    dst[i & 1].copy_(src[i & 1])
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for i, (a, b) in enumerate(evs):
    a.record()
    for j in range(4):
        dst[(i + j) & 1].copy_(src[(i + j) & 1])
    b.record()
torch.cuda.synchronize()
us = sorted(a.elapsed_time(b) * 1e3 / 4 for a, b in evs)
nbytes = 2 * src[0].numel() * 2
print(f"copy of {nbytes / 2e6:.1f} MB (read) + the same written: median {us[len(us) // 2]:.1f} us = {nbytes / us[len(us) // 2] / 1e6:.2f} TB/s "
      f"(read + write bytes), best {us[0]:.1f} us = {nbytes / us[0] / 1e6:.2f} TB/s")
