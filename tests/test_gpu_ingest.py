"""Row f2: the window feeder delivers exactly the slides it was given (order, dtype conversion, ragged lengths, ring
reuse over more windows than slots), from memory and from the reference's per-slide .pt layout, and a model trains
off it."""
import os

import pytest
import torch

from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.ingest import ArrayStore, PtDirStore, WindowFeeder

pytestmark = pytest.mark.gpu


def _cohort(n, seed):
    g = syn.rng(seed)
    lengths = [int(x) for x in torch.randint(40, 400, (n,), generator=torch.Generator().manual_seed(seed))]
    slides = [syn.normal(g, (m, 1024)) for m in lengths]
    omics = [[syn.normal(g, (w,)) for w in (16, 24, 8)] for _ in range(n)]
    labels = torch.arange(n) % 4
    cens = (torch.arange(n) % 2).float()
    return slides, omics, labels, cens


@pytest.mark.parametrize("bag_dtype", [torch.bfloat16, torch.float32])
def test_feeder_delivers_windows_in_order(dev, bag_dtype, tmp_path):
    slides, omics, labels, cens = _cohort(23, 3)
    order = [int(i) for i in torch.randperm(23, generator=torch.Generator().manual_seed(1))]
    for i, t in enumerate(slides):                                   # reference layout: <dir>/<slide_id>.pt
        torch.save(t, os.path.join(tmp_path, f"slide{i}.pt"))
    stores = [ArrayStore(slides), PtDirStore(str(tmp_path), [f"slide{i}.svs" for i in range(23)])]
    for store in stores:
        feeder = WindowFeeder(store, order, window=4, device=dev, bag_dtype=bag_dtype, depth=2, workers=4,
                              omics_of=lambda ids: [torch.stack([omics[i][k] for i in ids]) for k in range(3)],
                              labels_of=lambda ids: labels[ids], cens_of=lambda ids: cens[ids])
        seen = []
        for bags, om, lab, cen, ids in feeder:                       # 6 windows through a ring of 3 slots
            assert bags.data.dtype == bag_dtype and bags.lengths == [slides[i].shape[0] for i in ids]
            ref = torch.cat([slides[i] for i in ids]).to(bag_dtype)
            assert torch.equal(bags.data.cpu(), ref)
            assert torch.equal(bags.cu.cpu(), torch.tensor([0] + list(torch.tensor(bags.lengths).cumsum(0)), dtype=torch.int32))
            assert torch.equal(om[1].cpu(), torch.stack([omics[i][1] for i in ids]))
            assert torch.equal(lab.cpu(), labels[ids]) and torch.equal(cen.cpu(), cens[ids])
            seen += ids
        assert seen == order
        feeder.close()


def test_feeder_reports_a_bad_slide(dev):
    slides, omics, labels, cens = _cohort(6, 4)
    store = ArrayStore(slides)
    store.slides[3] = store.slides[3][:, :1000]                      # wrong feature width
    feeder = WindowFeeder(store, list(range(6)), window=2, device=dev, omics_of=lambda ids: [], labels_of=lambda ids: labels[ids],
                          cens_of=lambda ids: cens[ids])
    with pytest.raises(ValueError, match="slide 3"):
        for _ in feeder:
            pass
    feeder.close()


def test_feeder_keeps_slabs_intact_when_the_gpu_lags(dev):
    """The consumer's stream is stalled by a long kernel per window, so the packer runs a whole ring ahead of the GPU:
    a pinned slab must not be refilled before its previous H2D copy has left it (the copy itself is queued behind the
    stalled consumer).  Every window must still arrive bit-exact."""
    slides, omics, labels, cens = _cohort(24, 3)
    order = list(range(24))
    feeder = WindowFeeder(ArrayStore(slides), order, window=2, device=dev, bag_dtype=torch.bfloat16, depth=2, workers=4,
                          omics_of=lambda ids: [], labels_of=lambda ids: labels[ids], cens_of=lambda ids: cens[ids])
    sums, refs = [], []
    for bags, _, _, _, ids in feeder:                                # 12 windows through a ring of 3 slabs
        torch.cuda._sleep(40_000_000)                                # ~20 ms of GPU time in front of every use
        sums.append(bags.data.clone())                               # consumed on the (late) compute stream
        refs.append(torch.cat([slides[i] for i in ids]).to(torch.bfloat16))
    torch.cuda.synchronize(dev)
    for got, ref in zip(sums, refs):
        assert torch.equal(got.cpu(), ref)
    feeder.close()
