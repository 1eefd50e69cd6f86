"""CPU-only checks: the C-ABI library loads and exports every symbol include/mpo_hip.h declares,
host logic (ragged batches, slide assignment, C-index, ces loss) and the product's refusal to run
without a GPU.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from multimodal_path_omic_amd import _lib as L
from multimodal_path_omic_amd import harness, ops
from multimodal_path_omic_amd.dp import assign_slides
from oracle import mpo_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mpo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpo_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(L.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(L.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(handle, name), f"{name} declared in include/mpo_hip.h but not exported"
    for name in L.exported_symbols():
        assert name in decl, f"{name} bound in _lib.py but not declared in the public header"
    assert L.lib().mpo_abi_version() == 14
    # size queries are pure host functions
    assert L.lib().mpo_coattn_saved_floats(2, 6, 256) == 4 * 12 * 256 + 12
    assert L.lib().mpo_coattn_splits(32, 15000) == 8 and L.lib().mpo_coattn_splits(1, 100) == 1   # one workgroup per CU


def test_ops_refuse_cpu_tensors():
    x = torch.zeros(4, 8)
    with pytest.raises(RuntimeError, match="GPU tensors only"):
        ops.linear(x, torch.zeros(3, 8), torch.zeros(3))


def test_bag_batch_and_cu():
    with pytest.raises(ValueError):
        ops.make_cu([3, 0], "cpu")
    bags = [torch.zeros(m, 4) for m in (3, 1, 5)]
    b = ops.BagBatch.from_list(bags)
    assert b.cu.tolist() == [0, 3, 4, 9] and b.max_rows == 5 and b.total_rows == 9 and b.n_slides == 3
    flat = torch.arange(2 * 9, dtype=torch.float32)
    maps = b.split_map(flat, 2)
    assert [tuple(m.shape) for m in maps] == [(2, 3), (2, 1), (2, 5)]
    assert maps[1].tolist() == [[6.0], [7.0]]


def test_assign_slides_balances_and_is_deterministic():
    g = np.random.Generator(np.random.PCG64(3))
    lengths = [int(x) for x in g.integers(2000, 30001, size=32)]
    for world in (1, 2, 4, 8):
        parts = assign_slides(lengths, world)
        assert sorted(i for p in parts for i in p) == list(range(32))
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lengths)
        assert parts == assign_slides(lengths, world)


def test_ces_loss_matches_oracle_and_reference_constants():
    hz = torch.tensor([[0.51, 0.52, 0.49, 0.48]])
    s = torch.tensor([[0.5, 0.4, 0.2, 0.1]])
    # the reference's own known answers, models/loss.py:104-123
    assert harness.ces_loss(hz, s, torch.tensor([0]), torch.tensor([0.0])).item() == pytest.approx(0.6782951951026917, abs=1e-7)
    assert harness.ces_loss(hz, s, torch.tensor([0]), torch.tensor([1.0])).item() == pytest.approx(0.1732867956161499, abs=1e-7)
    g = torch.Generator().manual_seed(0)
    hzb = torch.sigmoid(torch.randn(5, 4, generator=g))
    svb = torch.cumprod(1 - hzb, 1)
    y = torch.tensor([0, 1, 2, 3, 1])
    c = torch.tensor([0., 1., 0., 1., 1.])
    per = harness.ces_loss(hzb, svb, y, c, reduction="none")
    for i in range(5):
        assert per[i].item() == pytest.approx(O.ces_loss(hzb[i:i + 1], svb[i:i + 1], y[i:i + 1], c[i:i + 1]).item(), abs=1e-6)


def test_c_index_vectorised_equals_oracle_loop():
    g = np.random.Generator(np.random.PCG64(9))
    for _ in range(20):
        n = int(g.integers(3, 40))
        event = g.random(n) < 0.7
        if not event.any():
            event[0] = True
        time = np.round(g.random(n) * 10, 1)                    # ties in time on purpose
        risk = np.round(g.standard_normal(n), 1)                # ties in risk on purpose
        try:
            ref = O.concordance_index_censored(event, time, risk)
        except ValueError:
            with pytest.raises(ValueError):
                harness.concordance_index_censored(event, time, risk)
            continue
        assert harness.concordance_index_censored(event, time, risk) == pytest.approx(ref, abs=1e-12)


def test_model_sizes_of_the_reference_construct():
    """small / medium / big (models/mcat/mcat.py:16-21, models/nacagat/nacagat.py:13-18, models/ge_nacagat/ge_nacagat.py:12-17)
    construct for all three models."""
    from multimodal_path_omic_amd.models import (GeneExprNarrowContextualAttentionGateTransformer,
                                                 MultimodalCoAttentionTransformer,
                                                 NarrowContextualAttentionGateTransformer)
    for size, d in (("small", 128), ("medium", 256), ("big", 512)):
        for cls in (MultimodalCoAttentionTransformer, NarrowContextualAttentionGateTransformer):
            m = cls(omic_sizes=[8] * 6, model_size=size)
            assert tuple(m.co_attention.in_proj_weight.shape) == (3 * d, d)
        ge = GeneExprNarrowContextualAttentionGateTransformer(model_size=size)
        assert tuple(ge.self_attention.in_proj_weight.shape) == (3 * d, d) and tuple(ge.H[0].weight.shape) == (d, 1024)


def test_bench_gpus_n_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` outside a torchrun environment: N rank processes through torch.distributed.run on
    127.0.0.1, this process neither touches the GPU nor is replaced (the box refuses an exec from a process that did)."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class _Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _Done()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert bench.self_launch(bench.parse(argv), argv) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-len(argv):] == argv and cmd[-len(argv) - 1].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_empty_slides_are_refused():
    import torch
    from multimodal_path_omic_amd.ops import BagBatch
    x = torch.zeros(5, 8)
    with pytest.raises(ValueError, match="at least one patch"):
        BagBatch.from_lengths(x, [5, 0])
    with pytest.raises(ValueError):
        BagBatch.from_lengths(x, [3, 3])
