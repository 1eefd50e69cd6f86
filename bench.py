#!/usr/bin/env python3
"""Headline benchmark: whole-model slides/sec (forward + backward + optimiser step) on synthetic
15k-patch bags (BASELINE.json configs[1]: MCAT, bf16-stored 15000 x 1024 patch bag + 6 x 256 omic
tokens), one process per GPU, RCCL gradient all-reduce once per optimiser step.

A "step" = one gradient-accumulation window of --window slides per rank pushed through the model as
one ragged batch (loss = ces, models/loss.py:5-28), gradients all-reduced, Adam step.  Prints ONE
JSON line (rank 0) with the contract fields plus `roofline` (the model's long-bag cross-attention
kernel timed by HIP events on its own stream against algorithmic bytes), `cpu_baseline` (the CPU
oracle timed on this box's host cores on a bounded sample, N=1 only) and `extra`: the same
measurement for the other BASELINE configs (cfg 3 NaCAGaT 15k, cfg 4 ragged 2k-30k windows, cfg 5
100k-patch fp32 bags), each with its own roofline.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(torch.distributed.run, 127.0.0.1) BEFORE this process touches the GPU, relays their output and exits
with their status; under the driver's torch.distributed.run launch every rank runs main() directly.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BASE_SEED = 1234               # synthetic inputs from seed 1234 + config index / rank
SETTLE_STEPS = 40            # default of --settle: untimed replays after graph capture and before the W warmup steps (setup; see run_config)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="mcat", choices=["mcat", "nacagat"])
    ap.add_argument("--model-size", default="medium", choices=["small", "medium", "big"],
                    help="models/mcat/mcat.py:16-21; the headline (and the default) is medium = embed 256")
    ap.add_argument("--window", type=int, default=32, help="slides per rank per optimiser step (grad_acc_step)")
    ap.add_argument("--patches", type=int, default=15000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--n-windows", type=int, default=2, help="distinct resident windows cycled (working set >> 256 MB L3)")
    ap.add_argument("--ragged", action="store_true",
                    help="BASELINE cfg 4: bag lengths drawn uniformly from [2000, 30000] (fixed multiset, length-aware "
                         "assignment of each window's slides to ranks) instead of --patches for every slide")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --window is the GLOBAL accumulation window (the reference's grad_acc_step = 32, "
                         "models/mcat/main.py:69-74), each of the N ranks holds window / N slides of it, so the optimiser sees "
                         "the same window -- the same trajectory -- at every N.  Default (weak): --window slides PER RANK, the "
                         "optimiser's window grows with N.")
    ap.add_argument("--wgrad-workgroups", type=int, default=None,
                    help="workgroups of the patch layer's weight-gradient kernel (default: 224 of the 256 CUs at N > 1 -- the "
                         "gradient all-reduce runs beside it and needs CUs of its own --, one per CU at N = 1)")
    ap.add_argument("--plan-workgroups", type=int, default=None, help="upper bound on the workgroups of the bag passes' work plan (A/B knob)")
    ap.add_argument("--settle", type=int, default=SETTLE_STEPS,
                    help="untimed replays of the captured step BEFORE the --warmup steps (clock / residency settling; reported as "
                         "config.setup_replays_before_warmup; 0 = the bare W-warmup / K-timed contract)")
    ap.add_argument("--no-extras", action="store_true", help="headline configuration only")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a captured HIP graph")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a, argv):
    """--gpus N > 1 without a torch.distributed environment: start N fresh rank processes.  Nothing in this process
    has touched the GPU yet (torch is not even imported), and it is not replaced: it waits for the children."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


# ------------------------------------------------------------------------------------------------ workload
def build_model(kind, dev, bag_dtype, rank, size="medium"):
    """Weights: module default init under torch.manual_seed(0) on EVERY rank (replicas start identical; FlatAdam's flat
    parameter buffer is additionally broadcast from rank 0).  Afterwards the generator is re-seeded per rank: the Philox
    seed of every dropout stream is torch.initial_seed() (ops._reserve), so ranks draw different masks."""
    import torch
    from multimodal_path_omic_amd.models import (MultimodalCoAttentionTransformer,
                                                 NarrowContextualAttentionGateTransformer)
    torch.manual_seed(0)
    cls = MultimodalCoAttentionTransformer if kind == "mcat" else NarrowContextualAttentionGateTransformer
    model = cls(omic_sizes=[256] * 6, model_size=size, bag_dtype=bag_dtype).to(dev).train()
    torch.manual_seed(BASE_SEED + 1000 * (rank + 1))
    return model


def make_windows(n_windows, window, patches, dev, bag_dtype, seed, ragged=False, rank=0, world=1):
    """Synthetic N(0,1) patch features generated on the device (seeded), labels/censorship cycling
    (i mod 4, i mod 2) as SURVEY 8(d) prescribes.  ragged: every window is a fixed multiset of world*window bag
    lengths in [2000, 30000] (the same at every GPU count for a given global window), dealt to the ranks by
    dp.assign_slides (longest-first bin packing on patch count); this rank keeps its share."""
    import torch
    from multimodal_path_omic_amd.dp import assign_slides
    from multimodal_path_omic_amd.ops import BagBatch, make_cu
    from multimodal_path_omic_amd.synthetic import slide_lengths
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = []
    for w in range(n_windows):
        if ragged:
            all_len = slide_lengths(world * window, 2000, 30000, 4321 + w)
            lengths = [all_len[i] for i in assign_slides(all_len, world)[rank]]
        else:
            lengths = [patches] * window
        window_n = len(lengths)
        data = torch.randn(sum(lengths), 1024, device=dev, dtype=torch.float32, generator=g).to(bag_dtype)
        bags = BagBatch(data, make_cu(lengths, dev), lengths)
        omics = [torch.randn(window_n, 256, device=dev, generator=g) for _ in range(6)]
        idx = torch.arange(window_n, device=dev) + w * window
        out.append((bags, omics, idx % 4, (idx % 2).float()))
    return out


def _time_launches(dev, launch, reps, burst=4, between=None):
    """HIP-event timing on the launching stream.
    between is None: one event pair brackets a burst of back-to-back launches (alternating resident inputs): a pair around
      a single launch also times that launch's dispatch latency (~3 us, which rocprofv3's kernel duration does not
      contain); inside a burst the next dispatch overlaps the running kernel.
    between given: every event pair is preceded by between(i) -- one replay of the captured window step -- so the kernel
      is timed in the state the chip has when it runs inside the workload, and brackets TWO launches (on the two resident
      inputs).  A pair around ONE launch right behind the step reads 10-20 us high with a wide spread (r04,
      tools/gpu_probe_event_timing.py: 288 us average / 270 minimum for the patch-layer kernel, against 276 / 273 per launch
      with two launches per pair, 274 / 270 with four, and 257-262 us for the same kernel in rocprofv3's timeline of the
      step); long bursts of the r03 kernel had read ~15 % slow instead (lower clock state), hence two.
    Returns (sorted per-launch microseconds, launches per event pair)."""
    import torch
    stream = torch.cuda.current_stream(dev)
    for i in range(3):
        launch(i)
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    if between is not None:
        burst = 2
    for i, (s, e) in enumerate(evs):
        if between is not None:
            between(i)
        s.record(stream)
        for j in range(burst):
            launch(i * burst + j)
        e.record(stream)
    torch.cuda.synchronize(dev)
    return sorted(s.elapsed_time(e) * 1e3 / burst for s, e in evs), burst


def _stored_profile(kernel_key, avg_us):
    """PMC-derived fields come from a STORED profile of the same kernel inside the same workload (profiles/r04_traffic.json:
    separate rocprofv3 --pmc passes over the eager window step, FETCH_SIZE x 2 per the gfx950 correction), not from this run:
    labelled as such.  Returns (hbm bytes per launch, matrix-pipe busy share, source)."""
    path = os.path.join(ROOT, "profiles", "r04_traffic.json")
    if not os.path.exists(path):
        return None, None, None
    with open(path) as f:
        prof = json.load(f).get(kernel_key)
    if not prof:
        return None, None, None
    busy = prof.get("SQ_VALU_MFMA_BUSY_CYCLES")
    # PMC busy cycles (summed over SIMDs) against this run's measured launch time, priced at the 2.4 GHz peak clock
    util = round(busy / (4 * 256 * avg_us * 1e-6 * 2.4e9), 4) if busy else None
    return prof.get("hbm_bytes_per_launch"), util, "profiles/r04_traffic.json <- " + prof.get("pmc_file", "")


def roofline_leg(dev, window, patches, bag_dtype, kind="mcat", reps=20, between=None):
    """Time the model's dominant long-bag forward kernel alone over a window of bags against its ALGORITHMIC bytes
    (DESIGN.md section 3):
      MCAT, bf16 window (the headline): the patch-layer pass (rows H2 / f1, patch_fc_fwd_kernel): reads the raw patch matrix
        once, writes H_bag once: M * (1024 + 256) * 2 bytes per slide; the co-attention itself (K1's forward pass over
        H_bag, M * 256 * 2 bytes per slide) is timed the same way and reported under "cross_attention";
      MCAT, fp32 window: K1's coattn_fwd_partial over H_bag, M * 256 * 4 bytes per slide;
      NaCAGaT: K2's bag_rowdot_gated over the key bag, fp32 whatever the bag dtype: M * 256 * 4 bytes per slide."""
    import torch
    from multimodal_path_omic_amd import _lib as L
    from multimodal_path_omic_amd.ops import BagBatch, make_cu
    E, n_q = 256, 6
    lib = L.lib()
    lengths = [patches] * window
    cu = make_cu(lengths, dev)
    stream = torch.cuda.current_stream(dev)
    parts = lib.mpo_coattn_target_workgroups() + window
    part_ml = torch.empty(parts * 32, device=dev)
    part_ctx = torch.empty(parts * n_q * E, device=dev)
    qk2 = torch.randn(window * n_q, E, device=dev) * 0.05
    fused = kind == "mcat" and bag_dtype == torch.bfloat16
    if fused:
        xs = [torch.randn(window * patches, 1024, device=dev).to(torch.bfloat16) for _ in range(2)]
        batch = BagBatch(xs[0], cu, lengths)
        plan = batch.plan()
        w = torch.randn(E, 1024, device=dev) / 32
        wb = torch.empty(E, 1024, device=dev, dtype=torch.bfloat16)
        L.check(lib.mpo_pack_patch_weight(L.ptr(w), L.ptr(wb), E, 1024, stream.cuda_stream), "mpo_pack_patch_weight")
        bias = torch.randn(E, device=dev) * 0.1
        h_out = torch.empty(window * patches, E, device=dev, dtype=torch.bfloat16)

        def launch(i):
            L.check(lib.mpo_patch_coattn_fwd_bagpass(L.ptr(xs[i & 1]), L.ptr(wb), L.ptr(bias), L.ptr(cu), window, None,
                                                     L.ptr(h_out), None, None, n_q, patches, 0.25, 1, 0,
                                                     plan, stream.cuda_stream), "mpo_patch_coattn_fwd_bagpass")
        alg_bytes = window * patches * (1024 + E) * 2
        name = "patch_fc_fwd_kernel<1024->256, bf16>"
        applies = window == 32 and patches == 15000
    else:
        k2 = kind == "nacagat"
        store = torch.float32 if k2 else bag_dtype
        esz = 4 if store == torch.float32 else 2
        bags = [torch.relu(torch.randn(window * patches, E, device=dev)).to(store) for _ in range(2)]
        batch = BagBatch(bags[0], cu, lengths)
        plan = batch.plan()
        tq = torch.tanh(torch.randn(window * n_q, E, device=dev))
        maps = torch.empty(2, n_q * window * patches, device=dev) if k2 else None

        def launch(i):
            if k2:
                L.check(lib.mpo_nacagat_fwd_bagpass(L.ptr(bags[i & 1]), L.ptr(cu), window, E, L.ptr(qk2), L.ptr(tq), L.ptr(maps[0]),
                                                    L.ptr(maps[1]), n_q, patches, plan, stream.cuda_stream), "mpo_nacagat_fwd_bagpass")
            else:
                L.check(lib.mpo_coattn_fwd_bagpass(L.ptr(bags[i & 1]), L.bag_dtype_code(bags[0]), L.ptr(cu), window, E,
                                                   L.ptr(qk2), L.ptr(part_ml), L.ptr(part_ctx), None, n_q, patches, plan,
                                                   stream.cuda_stream), "mpo_coattn_fwd_bagpass")
        alg_bytes = window * patches * E * esz
        name = "bag_rowdot_gated_exact_kernel<256> (f32 key bag)" if k2 else "coattn_fwd_partial_kernel<256,%s>" % ("bf16" if esz == 2 else "f32")
        applies = (window == 32 and patches == 15000) or (window == 8 and patches == 100000 and esz == 4 and not k2)
    extra = {"timing": "each event pair brackets two launches (the two resident inputs) and is preceded by one replay of the window step"} if between is not None else {}
    us, burst = _time_launches(dev, launch, reps, between=between)
    avg_us = sum(us) / len(us)
    achieved = alg_bytes / (avg_us * 1e-6) / 1e9
    traffic, mfma_util, source = _stored_profile(name, avg_us) if applies else (None, None, None)
    if fused:
        # the cross-attention proper (north_star: "the 15k-patch cross-attention kernel"): K1's forward bag pass over the bf16 H_bag
        hb = [torch.relu(torch.randn(window * patches, E, device=dev)).to(torch.bfloat16) for _ in range(2)]

        def launch_k1(i):
            L.check(lib.mpo_coattn_fwd_bagpass(L.ptr(hb[i & 1]), L.bag_dtype_code(hb[0]), L.ptr(cu), window, E, L.ptr(qk2),
                                               L.ptr(part_ml), L.ptr(part_ctx), None, n_q, patches, plan, stream.cuda_stream),
                    "mpo_coattn_fwd_bagpass")
        us1, burst1 = _time_launches(dev, launch_k1, reps, between=between)
        a1 = sum(us1) / len(us1)
        b1 = window * patches * E * 2
        t1, u1, s1 = _stored_profile("coattn_fwd_partial_kernel<256,bf16>", a1) if applies else (None, None, None)
        extra["cross_attention"] = {"kernel": "coattn_fwd_partial_kernel<256,bf16>", "bound": "hbm", "algorithmic_bytes_per_launch": b1,
                                    "traffic": t1, "mfma_util": u1, "traffic_source": s1,
                                    "avg_launch_us": round(a1, 2), "min_launch_us": round(us1[0], 2),
                                    "achieved": round(b1 / (a1 * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(b1 / (a1 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "launches_timed": reps * burst1}
    return {**extra, "bound": "hbm", "kernel": name,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "mfma_util": mfma_util,
            "traffic_source": source,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": round(avg_us, 2),
            "min_launch_us": round(us[0], 2), "launches_timed": reps * burst, "launches_per_event_pair": burst}


def cpu_baseline_leg(kind, patches, budget_s=15.0):
    """The CPU oracle (kind 'port': PyTorch-CPU restatement pinned to the reference by golden
    vectors) on this box's host cores: whole-model forward + ces + backward, fp32, one slide at a
    time like the reference loop."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases as C
    from multimodal_path_omic_amd import synthetic as syn
    from oracle import mpo_oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    sd = syn.fill_state_dict(C.model_shapes([256] * 6, kind == "nacagat"), 1)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    wsi = syn.make_bag(patches, 1235)
    omics = syn.make_omics([256] * 6, 1236)
    fwd = O.mcat_forward if kind == "mcat" else O.nacagat_forward

    def one():
        hz, sv, _, _ = fwd(p, wsi, omics)
        O.ces_loss(hz, sv, torch.tensor([1]), torch.tensor([0.0])).backward()
    one()
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 200:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "slides/s", "cores": threads, "kind": "port",
            "sample": f"{n} slides of {patches}x1024 fp32, {kind} medium, fwd+ces+bwd, one slide per call"}


def _all_ranks_ok(ok: bool, dev, world) -> bool:
    """Collective agreement on a per-rank outcome: True only if EVERY rank reports ok (MIN all-reduce; identity at N = 1)."""
    if world == 1:
        return ok
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def _ge_cpu_leg_inprocess(patches=15000):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import cases as C
    from multimodal_path_omic_amd import synthetic as syn
    from oracle import mpo_oracle as O
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    sd = syn.fill_state_dict(C.ge_model_shapes(), 2)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    wsi = syn.make_bag(patches, 78)
    t0 = time.perf_counter()
    y, _ = O.ge_nacagat_forward(p, wsi)
    O.ge_ce_loss(y, torch.tensor([1])).backward()
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "slides/s", "cores": threads, "kind": "port",
            "sample": f"1 slide of {patches}x1024 fp32, GE-NaCAGaT medium, fwd+CE+bwd ({dt:.1f} s)"}


def ge_cpu_baseline_leg(patches=15000, timeout_s=180, need_gib=64):
    """Row f3's CPU baseline: the oracle's gene-expression model (kind 'port') on ONE bag of the benchmarked length -- forward,
    cross-entropy, backward, fp32, all host threads.  One slide is the bounded sample: the reference algorithm keeps the
    8-head M x M probabilities of both encoder layers for the backward (~45 GB at M = 15 000) and takes tens of seconds.
    Because of that footprint it runs in a CHILD process, after a check of the host's free memory and under a time limit:
    a kill of the child (out of memory, limit) costs this leg, never the bench line."""
    import subprocess
    try:
        with open("/proc/meminfo") as f:
            avail = next(int(l.split()[1]) for l in f if l.startswith("MemAvailable:")) / 2 ** 20
    except Exception:
        avail = None
    if avail is not None and avail < need_gib:
        return {"error": f"skipped: {avail:.0f} GiB of host memory available, the oracle needs ~45 GiB (limit {need_gib})"}
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--_ge_cpu_leg", str(patches)], capture_output=True, text=True,
                           timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": f"child process exceeded {timeout_s} s"}
    if r.returncode != 0:
        return {"error": f"child process exit code {r.returncode}: {r.stderr[-160:]}"}
    return json.loads(r.stdout.strip().splitlines()[-1])


def run_config(a, dev, rank, world, steps, warmup, with_roofline=True):
    """Build the model and its resident windows, time `steps` window steps (barrier + synchronize on both sides, max over
    ranks) and return the contract dict (rank 0; None elsewhere)."""
    import torch
    import torch.distributed as dist
    from multimodal_path_omic_amd.dp import FlatAdam, FlatGradBucket
    from multimodal_path_omic_amd.harness import GraphedWindowStep, train_window
    bag_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    model = build_model(a.model, dev, bag_dtype, rank, a.model_size)
    bucket = FlatGradBucket(list(model.parameters()))
    opt = FlatAdam(bucket, lr=2e-4, weight_decay=1e-5)            # adam, lr 2e-4, wd 1e-5: config.yaml:57-63
    if world > 1:
        dist.broadcast(opt.flat_p, src=0)                         # replicas start from rank 0's weights
    windows = make_windows(a.n_windows, a.window, a.patches, dev, bag_dtype, seed=BASE_SEED + rank, ragged=a.ragged,
                           rank=rank, world=world)

    def eager_step(i):
        bags, omics, labels, cens = windows[i % len(windows)]
        bucket.begin()
        train_window(model, bags, omics, labels, cens, a.window)
        bucket.finish()
        bucket.all_reduce_mean()
        opt.step()

    graphed, graph_note = None, "eager"
    if not a.no_graph:
        # forward + backward (+ Adam when there is no all-reduce to wait for) captured per resident window
        try:
            graphed, pool = [], None
            for w in windows:
                gs = GraphedWindowStep(model, bucket, w, a.window, opt=opt if world == 1 else None, pool=pool,
                                       split_patch_grad=world > 1)
                pool = gs.pool()
                graphed.append(gs)
            graph_note = ("hipgraph(fwd+bwd+adam)" if world == 1 else
                          "hipgraph(fwd+bwd) | allreduce(rest) overlapped with hipgraph(dW_H) | allreduce(dW_H) | adam")
        except Exception as e:                                    # same kernels either way; say so in the output
            graphed, graph_note = None, f"eager (graph capture failed: {type(e).__name__}: {str(e)[:120]})"
            torch.cuda.synchronize(dev)
        # The graphed and the eager step issue DIFFERENT collective sequences (two split all-reduces around the dW_H graph
        # against one): the choice must be the same on every rank, or the ranks hang in mismatched collectives.
        if not _all_ranks_ok(graphed is not None, dev, world) and graphed is not None:
            graphed, graph_note = None, "eager (graph capture failed on another rank)"
    def step(i):
        if graphed is None:
            return eager_step(i)
        gs = graphed[i % len(graphed)]
        gs()
        if world > 1:
            # the one exchange step of the path, split so that most of it hides behind the patch layer's weight-gradient
            # GEMM: everything but that gradient is ready after the main graph
            head = gs.head_numel()
            rest = bucket.all_reduce_mean_async(lo=head)
            gs.replay_tail()
            first = bucket.all_reduce_mean_async(lo=0, hi=head)
            for h in (rest, first):
                if h is not None:
                    h.wait()
            opt.step()

    # Setup, not measurement: a few dozen replays so that the first timed steps do not pay for the card leaving its idle
    # state or for the first touch of the second resident window (the timed region of the driver's K = 20 is 22 ms long: one
    # 5 ms hiccup is a quarter of it -- seen once in round 3, 1.37 ms per step with every kernel at its usual time under
    # rocprofv3 minutes later).  Then the contract: W untimed warmup steps, exactly K timed ones.
    bare_ms = None
    if world == 1 and a.settle > 0:
        # the bare contract first, for the record (config.ms_per_step_without_setup_replays): W warmup steps and K timed ones
        # straight after graph capture, beside the headline figure below
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        torch.cuda.synchronize(dev)
        bare_ms = (time.perf_counter() - t0) / steps * 1e3
    for i in range(a.settle):
        step(i)
    torch.cuda.synchronize(dev)
    for i in range(warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = None
    if rank == 0:
        slides = world * a.window * steps            # (ragged: ranks hold different counts; the global window is world * window)
        out = {
            "metric": "slides/sec (fwd+bwd) at 15k-patch bags", "value": round(slides / dt, 2), "unit": "slides/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if getattr(a, "strong", False) else "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{a.model.upper()} {a.model_size} whole model fwd+bwd+Adam, "
                                   f"{'2k-30k' if a.ragged else a.patches}x1024 {a.dtype} patch bag "
                                   f"+ 6x256 omic tokens per slide, ces loss", "slides_per_rank_per_step": a.window,
                       "global_slides_per_step": world * a.window,
                       "patches_per_slide": "uniform[2000,30000] (fixed multiset)" if a.ragged else a.patches,
                       "parallelism": f"dp{world}", "resident_windows": a.n_windows, "launch": graph_note,
                       "setup_replays_before_warmup": a.settle,
                       "ms_per_step_without_setup_replays": None if bare_ms is None else round(bare_ms, 3)},
        }
    if rank == 0 and with_roofline:
        # the roofline kernel is timed inside the workload it belongs to: a replay of the captured step before every
        # timed launch (single process; with several ranks the step contains collectives: back-to-back launches there)
        between = (lambda i: step(i)) if (world == 1 and graphed is not None) else None
        out["roofline"] = roofline_leg(dev, a.window, a.patches if not a.ragged else 16000, bag_dtype, a.model, between=between)
    # release this configuration's graphs and windows before the next one is built
    del graphed, windows, opt, bucket, model
    gc.collect()
    torch.cuda.empty_cache()
    return out


def ge_extra(dev, patches=15000, steps=5):
    """Row f3 as an extra: the gene-expression model (models/ge_nacagat/ge_nacagat.py) on one 15 000 x 1024 bf16 bag per step, as
    the reference's loop feeds it (models/ge_nacagat/main.py:25-52): forward, CrossEntropyLoss on Y, backward -- training
    mode, every dropout on.  Roofline: the 8-head attention forward over the M rows (fp32 MFMA bound)."""
    import torch
    from multimodal_path_omic_amd import ops, synthetic as syn
    from multimodal_path_omic_amd.models import GeneExprNarrowContextualAttentionGateTransformer
    torch.manual_seed(0)
    model = GeneExprNarrowContextualAttentionGateTransformer(bag_dtype=torch.bfloat16).to(dev).train()
    wsi = syn.make_bag(patches, 77).to(dev).to(torch.bfloat16)
    target = torch.tensor([1], device=dev)

    def step():
        model.zero_grad(set_to_none=True)
        y, _ = model(wsi=wsi)
        torch.nn.functional.cross_entropy(y.unsqueeze(0), target).backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    heads, d = 8, 256
    qkv = torch.randn(1, patches, 3 * d, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
    with torch.no_grad():
        for a_, b_ in ev:
            a_.record()
            ops.BagSelfAttentionFn.apply(qkv, heads, 0.25, False)
            b_.record()
    torch.cuda.synchronize(dev)
    us = [a_.elapsed_time(b_) * 1e3 for a_, b_ in ev[1:]]
    flops = 2 * 2.0 * patches * patches * d                       # scores + context, all heads (algorithmic)
    avg = sum(us) / len(us)
    del model, wsi, qkv
    # the kernel runs each product as three bf16 MFMA terms (hi*hi + lo*hi + hi*lo): issued matrix flops = 3 x algorithmic,
    # priced against the dense bf16 peak; the rest of its time is the soft-max / dropout / operand-split VALU work
    return {"metric": "slides/sec (fwd+bwd) at 15k-patch bags", "value": round(1e3 / ms, 2), "unit": "slides/s", "n_gpus": 1,
            "steps": steps, "ms_per_step": round(ms, 2), "dtype": "bf16x3 attention (hi+lo operands, fp32 accumulate) / bf16 bag",
            "data": "synthetic",
            "config": {"workload": f"GE-NACAGAT medium fwd+CE+bwd, one {patches}x1024 bf16 bag per step, M x M map returned",
                       "launch": "eager"},
            "roofline": {"bound": "mfma", "kernel": "bag_sa_b3_fwd_kernel<32> (8 heads, dropout 0.25, three-term bf16 MFMA)",
                         "achieved": round(3 * flops / avg / 1e6, 2), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(3 * flops / avg / 1e6 / 2500.0, 4), "traffic": None, "flops_per_launch": 3 * flops,
                         "algorithmic_flops_per_launch": flops, "algorithmic_tflops": round(flops / avg / 1e6, 2),
                         "avg_launch_us": round(avg, 1), "min_launch_us": round(min(us), 1), "launches_timed": len(us)}}


def extras(a, dev, rank, world):
    """The other BASELINE configs as `extra` entries of the same line: cfg 4 (ragged windows; at every N, it is the
    data-parallel config), and at N = 1 also cfg 3 (NaCAGaT) and cfg 5 (100k-patch fp32 bags, window 8)."""
    plan = [("cfg4_ragged_2k_30k", dict(model=a.model, ragged=True, patches=15000, dtype="bf16", n_windows=2,
                                        window=a.window if getattr(a, "strong", False) else 32), False)]
    if world == 1:
        plan += [("cfg3_nacagat_15k", dict(model="nacagat", ragged=False, patches=15000, dtype="bf16", window=32, n_windows=2), True),
                 ("cfg5_mcat_100k_fp32", dict(model="mcat", ragged=False, patches=100000, dtype="f32", window=8, n_windows=2), True)]
    out = {}
    for name, over, roof in plan:
        b = argparse.Namespace(**{**vars(a), **over})
        err = None
        try:
            r = run_config(b, dev, rank, world, steps=max(5, min(a.steps, 10)), warmup=2, with_roofline=roof)
        except Exception as e:                                    # an extra must never take the headline line down
            err, r = f"{type(e).__name__}: {str(e)[:200]}", None
        if world > 1 and err is not None:
            # a rank that left run_config() early has skipped collectives its peers are still waiting in: there is no
            # sequence to rejoin.  Fail the job (the launcher tears the other ranks down) rather than hang until the timeout.
            print(f"[bench] rank {rank}: extra '{name}' failed ({err}); aborting the multi-rank run", file=sys.stderr, flush=True)
            os._exit(3)
        if rank == 0:
            out[name] = r if err is None else {"error": err}
    if world == 1:
        try:
            out["f3_ge_nacagat_15k"] = ge_extra(dev)
        except Exception as e:
            out["f3_ge_nacagat_15k"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    return out


def main():
    argv = sys.argv[1:]
    if argv[:1] == ["--_ge_cpu_leg"]:                             # the child of ge_cpu_baseline_leg(): CPU only, no GPU touched
        print(json.dumps(_ge_cpu_leg_inprocess(int(argv[1]))), flush=True)
        return
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, argv))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL.  MPO_DIST_BACKEND=gloo exists to rehearse the N > 1 code path with several ranks sharing
        # one card (RCCL refuses two ranks on the same device); the driver's runs never set it.
        dist.init_process_group(os.environ.get("MPO_DIST_BACKEND", "nccl"))
    if a.gpus != world and rank == 0:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE is {world}: reporting n_gpus = {world}", file=sys.stderr)
    dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)

    from multimodal_path_omic_amd import ops as _ops
    _ops.wgrad_workgroups = a.wgrad_workgroups if a.wgrad_workgroups is not None else (224 if world > 1 else None)
    _ops.plan_workgroups = a.plan_workgroups
    if a.strong:
        if a.window % world:
            sys.exit(f"--strong: the global window of {a.window} slides does not divide over {world} ranks")
        a.global_window = a.window
        a.window //= world                                        # per rank; the loss scale 1 / window x the mean over ranks = 1 / global window
    out = run_config(a, dev, rank, world, a.steps, a.warmup)
    if not a.no_extras and not a.ragged and a.model == "mcat" and a.patches == 15000 and a.dtype == "bf16":
        ex = extras(a, dev, rank, world)
        if rank == 0:
            out["extra"] = ex
    # the CPU leg comes LAST: ~20 s of 64 busy host threads right before a GPU leg left that leg's first seconds slow
    # (measured: the ragged extra at 1.92 ms per step after it, 1.47 ms before it)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_leg(a.model, a.patches)
        f3 = out.get("extra", {}).get("f3_ge_nacagat_15k")
        if isinstance(f3, dict) and "error" not in f3:
            try:
                f3["cpu_baseline"] = ge_cpu_baseline_leg()
            except Exception as e:                                # (host memory: the oracle materialises the M x M maps)
                f3["cpu_baseline"] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
