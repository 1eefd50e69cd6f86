"""Probe (not a test): what one node of a dependent launch chain costs inside a captured HIP graph (NOTES r02-tail-floor).

    python tools/gpu_probe_chain.py [suite ...]        suites: basic copy bump sizes deps   (default: all)

  basic  trivial add / small GEMM / FFN pair / torch layer_norm chains
  copy   identical trivial kernels: alone, after a 1 GiB copy, 12 distinct kernels, GEMM/LN/add mix, 64 MB copy between nodes
  bump   this library's own trivial kernel (counters_bump) alone and alternating with add / GEMM
  sizes  elementwise and layer_norm chains at the tail's tensor sizes (49k .. 1.5M floats)
  deps   does an elementwise kernel cost more right after the tiled GEMM that wrote its input?
"""
import sys
import time

import torch

sys.path[:0] = ["."]
from multimodal_path_omic_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def run(name, fn, n=96, reps=10):
    with torch.no_grad():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn(n)
            fn(n)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = fn(n)                                            # noqa: F841  (outputs stay alive with the graph)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name:52s}: {dt * 1e6:9.1f} us total, {dt / n * 1e6:6.2f} us per chain element", flush=True)


def chain(step, x0):
    def fn(n):
        y = x0
        for i in range(n):
            y = step(y, i)
        return y
    return fn


def suite_basic():
    x = torch.randn(192, 256, device=dev)
    w = torch.randn(256, 256, device=dev) / 16
    w2, w3 = torch.randn(512, 256, device=dev) / 16, torch.randn(256, 512, device=dev) / 16
    b, b2 = torch.zeros(256, device=dev), torch.zeros(512, device=dev)
    run("trivial add", chain(lambda y, i: y + 1.0, x))
    run("gemm 192x256x256", chain(lambda y, i: ops.linear(y, w, b), x))
    run("ffn 256->512->256 (two kernels per element)", chain(lambda y, i: ops.linear(ops.linear(y, w2, b2), w3, b), x), n=48)
    run("torch layer_norm 192x256", chain(lambda y, i: torch.nn.functional.layer_norm(y, (256,)), x))


def suite_copy():
    x = torch.randn(192, 256, device=dev)
    big = torch.randn(256 * 1024 * 1024, device=dev)                # 1 GiB
    big2 = torch.empty_like(big)
    w, b = torch.randn(256, 256, device=dev) / 16, torch.zeros(256, device=dev)
    fns = [lambda y: y + 1.0, lambda y: y * 1.001, lambda y: y - 0.5, torch.neg, torch.abs, torch.relu, torch.sigmoid,
           torch.tanh, torch.sin, torch.cos, lambda y: torch.clamp(y, -1, 1), lambda y: torch.sqrt(torch.abs(y))]

    def after_copy(n):
        big2.copy_(big)
        return chain(lambda y, i: y + 1.0, x)(n)

    def copies_between(n):
        y = x
        for _ in range(n):
            big2[:16 * 1024 * 1024].copy_(big[:16 * 1024 * 1024])
            y = y + 1.0
        return y
    run("same trivial kernel", chain(lambda y, i: y + 1.0, x))
    run("1 GiB copy, then the same trivial kernel", after_copy)
    run("12 distinct trivial kernels", chain(lambda y, i: fns[i % len(fns)](y), x))
    run("gemm / layer_norm / add mix", chain(lambda y, i: (ops.linear(y, w, b) if i % 3 == 0 else
                                                           torch.nn.functional.layer_norm(y, (256,)) if i % 3 == 1 else y + 1.0), x))
    run("trivial kernel after a 64 MB copy each", copies_between)


def suite_bump():
    e = torch.zeros(1, dtype=torch.int64, device=dev)
    t = torch.zeros(1, dtype=torch.int32, device=dev)
    x = torch.randn(192, 256, device=dev)
    w, b = torch.randn(256, 256, device=dev) / 16, torch.zeros(256, device=dev)

    def bump(y, i):
        ops.bump_step_counters(e, t)
        return y
    run("counters_bump", chain(bump, x))
    run("counters_bump / add alternating", chain(lambda y, i: bump(y, i) if i % 2 == 0 else y + 1.0, x))
    run("counters_bump / gemm alternating", chain(lambda y, i: bump(y, i) if i % 2 == 0 else ops.linear(y, w, b), x))
    run("gemm / gemm(relu): same kernel", chain(lambda y, i: ops.linear(y, w, b, "relu") if i % 2 else ops.linear(y, w, b), x))


def suite_sizes():
    for numel in (49152, 98304, 393216, 1572864):
        a = torch.randn(numel, device=dev)
        m = torch.randn(numel, device=dev) * 0.01 + 1.0
        run(f"torch mul chain, {numel} floats", chain(lambda y, i, m=m: y * m, a))
        run(f"torch layer_norm chain, {numel // 256} x 256",
            chain(lambda y, i: torch.nn.functional.layer_norm(y, (256,)), a.view(-1, 256)))


def suite_deps():
    x = torch.randn(384, 256, device=dev)
    w, b = torch.randn(256, 256, device=dev) / 16, torch.zeros(256, device=dev)
    m = torch.randn(384, 256, device=dev) * 0.01 + 1.0

    def indep(n):
        y, z = x, x
        for _ in range(n // 2):
            y = ops.linear(y, w, b)
            z = z * m
        return y, z
    run("gemm 384x256x256 chain", chain(lambda y, i: ops.linear(y, w, b), x))
    run("torch mul 98304 chain", chain(lambda y, i: y * m, x))
    run("gemm -> mul (dependent) alternating", chain(lambda y, i: ops.linear(y, w, b) if i % 2 == 0 else y * m, x))
    run("gemm, mul (independent data) alternating", indep)
    run("gemm -> torch layer_norm (dependent) alternating",
        chain(lambda y, i: ops.linear(y, w, b) if i % 2 == 0 else torch.nn.functional.layer_norm(y, (256,)), x))


SUITES = {"basic": suite_basic, "copy": suite_copy, "bump": suite_bump, "sizes": suite_sizes, "deps": suite_deps}
if __name__ == "__main__":
    for name in (sys.argv[1:] or list(SUITES)):
        print(f"--- {name}", flush=True)
        SUITES[name]()
