"""Does a collective-sized kernel on a second stream run DURING the patch layer's weight-gradient kernel?  (VERDICT r03 item
3a)  The data-parallel step starts the all-reduce of flat[head:] and then replays the dW_H graph: the
exchange hides only if the RCCL kernel gets CUs while patch_wgrad_kernel -- persistent, one workgroup per CU, all of a CU's LDS
and registers -- is running.  One card, no RCCL: a stand-in kernel (tools/probes/occupy.hip: 16 / 32 persistent workgroups
sweeping a 17 MB buffer, ~100-200 us alone) on a side stream, the weight gradient of a 32 x 15 000 window on the main stream,
with 256 (one per CU) and 224 workgroups.  Prints HIP-event times: A alone, B alone, A || B; under rocprofv3 --kernel-trace
the csv shows the overlap directly (tools/gpu_trace_two_streams.py)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from multimodal_path_omic_amd import ops

dev = torch.device("cuda:0")
occ = ctypes.CDLL(os.path.join(ROOT, "tools", "_bin", "liboccupy.so"))
occ.occupy_launch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
rows = 32 * 15000
g = (torch.randn(rows, 256, device=dev) * 0.01).to(torch.bfloat16)
x = torch.randn(rows, 1024, device=dev).to(torch.bfloat16)
out = torch.empty(256, 1024, device=dev)
bucket = torch.zeros(4_200_000, device=dev)          # ~ the 4.16 M-element gradient bucket
main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)


def wgrad():
    ops.patch_weight_grad(g, x, out)


def occupy(wgs, iters):
    rc = occ.occupy_launch(bucket.data_ptr(), bucket.numel(), iters, wgs, side.cuda_stream)
    assert rc == 0, rc


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(main)
    for _ in range(n):
        fn()
    e.record(main)
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for occ_wgs, iters in ((16, 1), (32, 2)):
    def b_alone():
        side.wait_stream(main)
        occupy(occ_wgs, iters)
        main.wait_stream(side)
    tb = timed(b_alone)
    for wgs in (256, 224):
        ops.wgrad_workgroups = None if wgs == 256 else wgs
        ta = timed(wgrad)

        def both():
            side.wait_stream(main)             # fork: the collective starts where the step starts it
            occupy(occ_wgs, iters)
            wgrad()
            main.wait_stream(side)             # join before the optimiser
        tab = timed(both)
        print(f"stand-in {occ_wgs} workgroups ({tb:6.1f} us alone) | dW_H on {wgs} workgroups {ta:6.1f} us alone | together {tab:6.1f} us "
              f"(hidden: {ta + tb - tab:6.1f} us of {tb:6.1f})")
ops.wgrad_workgroups = None
