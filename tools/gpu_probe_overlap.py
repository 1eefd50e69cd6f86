"""Probe: can the token tail of one half-window overlap the bag kernels of the other?

Two INDEPENDENT captured steps (model replica + gradient bucket + optimiser each, 16 slides of 15 000 patches) are replayed
  serial      both on one stream, one after the other,
  concurrent  on two streams at once -- the hardware is free to co-schedule half A's tail launches with half B's bag kernels,
against the product's single 32-slide step.  Repeated with the bag kernels' work plans cut to fewer workgroups than CUs
(ops.plan_workgroups), which leaves whole CUs to the other stream's launches (a persistent bag workgroup takes a CU's LDS
and register file, so nothing co-resides with it).
    python tools/gpu_probe_overlap.py [slides_per_half] [patches] [targets, e.g. none,224]"""
import os
import sys
import time

import torch

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import bench as B  # noqa: E402
from multimodal_path_omic_amd import ops  # noqa: E402
from multimodal_path_omic_amd.dp import FlatAdam, FlatGradBucket  # noqa: E402
from multimodal_path_omic_amd.harness import GraphedWindowStep  # noqa: E402

half = int(sys.argv[1]) if len(sys.argv) > 1 else 16
patches = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
targets = [None if t == "none" else int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [None, 240, 224, 192, 128]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def make_step(n_slides, seed):
    model = B.build_model("mcat", dev, torch.bfloat16, 0)
    bucket = FlatGradBucket(list(model.parameters()))
    opt = FlatAdam(bucket, lr=2e-4, weight_decay=1e-5)
    w = B.make_windows(1, n_slides, patches, dev, torch.bfloat16, seed=seed)[0]
    return GraphedWindowStep(model, bucket, w, n_slides, opt=opt)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps * 1e3


s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def concurrent(a, b):
    def fn():
        with torch.cuda.stream(s1):
            a.graph.replay()
        with torch.cuda.stream(s2):
            b.graph.replay()
    return fn


def serial(a, b):
    def fn():
        with torch.cuda.stream(s1):
            a.graph.replay()
            b.graph.replay()
    return fn


for target in targets:
    ops.plan_workgroups = target
    full = make_step(2 * half, 1)
    t_full = timed(lambda: full.graph.replay())
    del full
    a, b = make_step(half, 2), make_step(half, 3)
    t_one = timed(lambda: a.graph.replay())
    t_ser = timed(serial(a, b))
    t_con = timed(concurrent(a, b))
    print(f"plan workgroups {str(target):>4s}: one {2 * half}-slide step {t_full:.3f} ms | one {half}-slide step {t_one:.3f} ms | "
          f"two {half}-slide steps serial {t_ser:.3f} ms, on two streams {t_con:.3f} ms", flush=True)
    del a, b
    torch.cuda.empty_cache()
ops.plan_workgroups = None
