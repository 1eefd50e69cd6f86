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


def test_bucket_views_alias_param_grads():
    model = _model()
    bucket = FlatGradBucket(list(model.parameters()))
    bucket.begin()
    _loss(model, torch.ones(3, 8)).backward()
    bucket.finish()                                               # stock-torch gradients are copied into the slices
    assert bucket.flat.abs().sum() > 0
    off = 0
    for p in model.parameters():
        assert p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * off
        off += p.numel()
    ref = bucket.flat.clone()
    _loss(model, torch.ones(3, 8)).backward()                     # a second backward accumulates in place
    torch.testing.assert_close(bucket.flat, 2 * ref)
    bucket.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in model.parameters())
