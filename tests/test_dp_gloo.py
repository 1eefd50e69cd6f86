"""Multi-process data-parallel path on CPU (gloo, world_size 2): the flat gradient bucket and its single
all-reduce per optimiser step reproduce single-process accumulation over the union of the slides.
(The HIP kernels need a GPU; the collective logic does not.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodal_path_omic_amd.dp import FlatGradBucket, assign_slides


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))


def _slides():
    g = torch.Generator().manual_seed(1)
    lengths = [5, 40, 17, 3, 29, 11, 8, 23]
    return lengths, [torch.randn(m, 8, generator=g) for m in lengths]


def _loss(model, bag):
    return model(bag).mean() ** 2


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model()
    bucket = FlatGradBucket(list(model.parameters()))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    lengths, slides = _slides()
    mine = assign_slides(lengths, world)[rank]
    bucket.begin()
    for i in mine:
        (_loss(model, slides[i]) / len(mine)).backward()          # per-rank 1/grad_acc_step scaling
    bucket.finish()
    # the split exchange of the window step (dp.FlatGradBucket.all_reduce_mean_async): tail slice first, head slice second,
    # must equal one all_reduce_mean over the whole bucket
    head = bucket.flat.numel() // 3
    handles = [bucket.all_reduce_mean_async(lo=head), bucket.all_reduce_mean_async(lo=0, hi=head)]
    for h in handles:
        h.wait()
    opt.step()
    if rank == 0:
        torch.save([p.detach().clone() for p in model.parameters()], out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_allreduce_equals_single_process(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "params.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    # single process: the same 8 slides, 4 per (virtual) rank, mean over ranks
    model = _model()
    lengths, slides = _slides()
    parts = assign_slides(lengths, world)
    for part in parts:
        for i in part:
            (_loss(model, slides[i]) / len(part) / world).backward()
    torch.optim.SGD(model.parameters(), lr=0.1).step()
    for a, b in zip(got, model.parameters()):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)


def _worker_n(rank, world, port, out, strong):
    """One optimiser step of the data-parallel path at world size N on the cfg-4 multiset (32 slides per rank, or 32 in all
    with `strong`), dealt by assign_slides; the toy model stands in for the kernels, the exchange is the product's."""
    from multimodal_path_omic_amd.synthetic import slide_lengths
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _model()
    bucket = FlatGradBucket(list(model.parameters()))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    n_global = 32 if strong else 32 * world
    lengths = slide_lengths(n_global, 2000, 30000, 4321)
    mine = assign_slides(lengths, world)[rank]
    g = torch.Generator().manual_seed(7)
    feats = torch.randn(n_global, 8, generator=g)                 # one feature row per slide (its length only weighs the split)
    bucket.begin()
    for i in mine:
        # every slide carries 1 / (global window): ranks may hold different COUNTS (the split balances patches, not slides),
        # so the per-rank scale is world / n_global and the exchange takes the mean over ranks
        (_loss(model, feats[i:i + 1]) * (world / n_global)).backward()
    bucket.finish()
    bucket.all_reduce_mean()
    opt.step()
    if rank == 0:
        torch.save([p.detach().clone() for p in model.parameters()], out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world,strong", [(4, False), (8, False), (4, True), (8, True)],
                         ids=["world4_weak", "world8_weak", "world4_strong", "world8_strong"])
def test_exchange_at_world_4_and_8_equals_single_process(tmp_path, world, strong):
    """the data-parallel step at N = 4 and 8 on the fixed 2k-30k multiset of BASELINE config 4 -- the
    length-aware split leaves the slowest rank within 5 % of the mean load, and bucket + one all-reduce reproduce the
    single-process update over the global window; `strong` keeps the global window at 32 slides (the reference's
    grad_acc_step, models/mcat/main.py:69-74: the optimiser's trajectory does not change with N), weak grows it with N."""
    from multimodal_path_omic_amd.synthetic import slide_lengths
    n_global = 32 if strong else 32 * world
    lengths = slide_lengths(n_global, 2000, 30000, 4321)
    parts = assign_slides(lengths, world)
    assert sorted(i for p in parts for i in p) == list(range(n_global))
    loads = [sum(lengths[i] for i in p) for p in parts]
    mean = sum(loads) / world
    imbalance = max(loads) / mean - 1.0
    print(f"[dp] world {world} {'strong' if strong else 'weak'}: {n_global} slides, load max/mean - 1 = {imbalance:.3%}")
    assert imbalance <= 0.05, imbalance                           # (measured: 0.04 % .. 1.3 %)
    port, out = _free_port(), str(tmp_path / "p.pt")
    mp.spawn(_worker_n, args=(world, port, out, strong), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    model = _model()
    g = torch.Generator().manual_seed(7)
    feats = torch.randn(n_global, 8, generator=g)
    for i in range(n_global):
        (_loss(model, feats[i:i + 1]) / n_global).backward()
    torch.optim.SGD(model.parameters(), lr=0.1).step()
    for a, b in zip(got, model.parameters()):
        torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-6)


def test_bucket_views_alias_param_grads():
    model = _model()
    bucket = FlatGradBucket(list(model.parameters()))
    bucket.begin()
    _loss(model, torch.ones(3, 8)).backward()
    bucket.finish()                                               # stock-torch gradients are copied into the slices
    assert bucket.flat.abs().sum() > 0
    for p, off in zip(model.parameters(), bucket.offsets):       # slices in parameter order, each 256-byte aligned
        assert p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * off and off % 64 == 0
    ref = bucket.flat.clone()
    _loss(model, torch.ones(3, 8)).backward()                     # a second backward accumulates in place
    torch.testing.assert_close(bucket.flat, 2 * ref)
    bucket.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in model.parameters())


# ---------------------------------------------------------------------------------------------------------------
# The REAL model's parameter layout through the bucket, the flat optimiser buffers and the split exchange of the
# data-parallel step (bench.py / harness.GraphedWindowStep(split_patch_grad=True)): flat[head:] is reduced while the
# patch layer's weight gradient flat[:head] is still being computed, then flat[:head].  CPU tensors + gloo: the bucket
# logic needs no kernel (the HIP kernels are covered on one GPU by tests/test_gpu_graph.py, and the whole step at
# world size 2 by tests/test_gpu_dp.py).
def _real_model():
    from multimodal_path_omic_amd.models import MultimodalCoAttentionTransformer
    torch.manual_seed(0)
    return MultimodalCoAttentionTransformer(omic_sizes=[32, 48, 64], model_size="small")


def _synthetic_grads(n, rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(n, generator=g)


def _real_worker(rank, world, port, out):
    from multimodal_path_omic_amd.dp import FlatAdam
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _real_model()
    bucket = FlatGradBucket(list(model.parameters()))
    opt = FlatAdam(bucket, lr=1e-3)                       # construction is pure torch: parameters re-pointed at flat_p
    dist.broadcast(opt.flat_p, src=0)
    head = bucket.head_numel(model.H[0].weight)
    n = bucket.flat.numel()
    bucket.begin()
    bucket.finish()                                       # nothing ran: every slice zero-filled, .grad re-attached
    assert float(bucket.flat.abs().sum()) == 0.0
    bucket.flat.copy_(_synthetic_grads(n, rank))
    bucket.flat[:head].zero_()                            # "the main graph has run": everything but dW_H is there
    rest = bucket.all_reduce_mean_async(lo=head)
    bucket.flat[:head].copy_(_synthetic_grads(n, rank)[:head])     # "the tail graph": dW_H lands while `rest` is in flight
    first = bucket.all_reduce_mean_async(lo=0, hi=head)
    for h in (rest, first):
        h.wait()
    if rank == 0:
        torch.save({"flat": bucket.flat.clone(), "head": head,
                    "grad_H": model.H[0].weight.grad.clone(), "grad_cls": model.classifier.bias.grad.clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_split_exchange_on_real_parameter_layout(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "real.pt")
    mp.spawn(_real_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    model = _real_model()
    bucket = FlatGradBucket(list(model.parameters()))            # (same layout as in the workers: slices 256-byte aligned)
    n = bucket.flat.numel()
    assert n == got["flat"].numel() >= sum(p.numel() for p in model.parameters())
    want = sum(_synthetic_grads(n, r) for r in range(world)) / world
    torch.testing.assert_close(got["flat"], want, rtol=1e-6, atol=1e-7)
    assert got["head"] == model.H[0].weight.numel() == 1024 * 128
    # parameter .grad tensors ARE the bucket slices (first and last parameter of the model)
    torch.testing.assert_close(got["grad_H"].flatten(), want[:got["head"]])
    last = bucket.offsets[-1]
    torch.testing.assert_close(got["grad_cls"].flatten(), want[last:last + model.classifier.bias.numel()])


def test_flat_adam_repoints_parameters_and_keeps_state_dict():
    from multimodal_path_omic_amd.dp import FlatAdam
    model = _real_model()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    bucket = FlatGradBucket(list(model.parameters()))
    opt = FlatAdam(bucket, lr=1e-3)
    after = model.state_dict()
    assert list(before) == list(after)
    for k in before:
        torch.testing.assert_close(before[k], after[k], rtol=0, atol=0)
    for p, off in zip(model.parameters(), bucket.offsets):    # every parameter is a view of the flat buffer, in bucket order
        assert p.data_ptr() == opt.flat_p.data_ptr() + 4 * off and off % 64 == 0
    assert opt.flat_p.numel() == bucket.flat.numel() >= sum(p.numel() for p in model.parameters())
    opt.flat_p.add_(1.0)                                  # an update of the flat buffer IS an update of the model
    torch.testing.assert_close(model.classifier.bias.detach(), before["classifier.bias"] + 1.0)
    with pytest.raises(RuntimeError):                     # the update kernel is HIP: no CPU fallback
        opt.step()


def test_bucket_slice_is_handed_out_once_per_window():
    """ops.grad_out: the kernels OVERWRITE their gradient outputs, so a parameter's bucket slice may be handed to one
    producer per window only; a second producer gets a private tensor (autograd then adds the two)."""
    from multimodal_path_omic_amd import ops
    model = _model()
    bucket = FlatGradBucket(list(model.parameters()))
    p = next(model.parameters())
    bucket.begin()
    a = ops.grad_out(p)
    b = ops.grad_out(p)
    assert a.data_ptr() == p._mpo_grad_view.data_ptr()
    assert b.data_ptr() != a.data_ptr()
    bucket.begin()                                        # next window: the slice is available again
    assert ops.grad_out(p).data_ptr() == p._mpo_grad_view.data_ptr()
