"""HIP-graph replay of the window step: equals the eager step with dropout off, and draws fresh dropout masks
on every replay with dropout on (device-resident RNG epoch), with Adam's bias correction following the
device-resident step count."""
import pytest
import torch

import cases as C
from multimodal_path_omic_amd import harness, ops
from multimodal_path_omic_amd import synthetic as syn
from multimodal_path_omic_amd.dp import FlatAdam, FlatGradBucket
from multimodal_path_omic_amd.models import MultimodalCoAttentionTransformer

pytestmark = pytest.mark.gpu


def setup(dev, train_mode, bag_dtype=torch.bfloat16):
    sizes = [64] * 6
    model = MultimodalCoAttentionTransformer(omic_sizes=sizes, bag_dtype=bag_dtype)
    model.load_state_dict(syn.fill_state_dict(C.model_shapes(sizes, False), 55))
    model.to(dev)
    model.train(train_mode)
    slides = syn.make_cohort(6, 200, 700, sizes, 56)
    window = harness.make_window(slides, dev, bag_dtype)
    bucket = FlatGradBucket(list(model.parameters()))
    opt = FlatAdam(bucket, lr=1e-3, weight_decay=1e-5)
    return model, bucket, opt, window


def test_graph_replay_equals_eager_steps(dev):
    ops.set_rng_epoch(None)
    model_e, bucket_e, opt_e, window_e = setup(dev, False)
    losses_e = []
    for _ in range(4):
        bucket_e.begin()
        loss, _ = harness.train_window(model_e, *window_e, 6)
        bucket_e.finish()
        opt_e.step()
        losses_e.append(loss.clone())
    ops.set_rng_epoch(None)
    model_g, bucket_g, opt_g, window_g = setup(dev, False)
    step = harness.GraphedWindowStep(model_g, bucket_g, window_g, 6, opt=opt_g, warmup=0)   # capture = step 1
    losses_g = [step.loss.clone()]
    # the capture pass does not execute kernels; replay 4 times = steps 1..4
    losses_g = []
    for _ in range(4):
        loss, _ = step()
        losses_g.append(loss.clone())
    for a, b in zip(losses_e, losses_g):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-4)
    assert float(losses_g[-1].mean()) < float(losses_g[0].mean())        # it trains
    torch.testing.assert_close(opt_e.flat_p, opt_g.flat_p, rtol=5e-3, atol=5e-4)
    assert int(opt_g.t_dev) == 4
    ops.set_rng_epoch(None)


def test_graph_replays_draw_fresh_dropout_masks(dev):
    ops.set_rng_epoch(None)
    model, bucket, _, window = setup(dev, True)
    step = harness.GraphedWindowStep(model, bucket, window, 6, opt=None, warmup=1)
    l1 = step()[0].clone()
    l2 = step()[0].clone()
    l3 = step()[0].clone()
    assert not torch.equal(l1, l2) and not torch.equal(l2, l3)           # same weights, different masks
    assert torch.isfinite(l1).all() and torch.isfinite(bucket.flat).all()
    ops.set_rng_epoch(None)


@pytest.mark.parametrize("kind", ["mcat", "nacagat"])
def test_split_step_fills_the_same_bucket(dev, kind):
    """Data-parallel steps capture the patch layer's weight gradient into a second graph (run while the all-reduce of
    the other gradients is in flight): main + tail must leave exactly the gradients of the one-graph step, and the main
    graph alone must not touch the head slice."""
    from multimodal_path_omic_amd.models import NarrowContextualAttentionGateTransformer
    ops.set_rng_epoch(None)
    sizes = [64] * 6
    cls = MultimodalCoAttentionTransformer if kind == "mcat" else NarrowContextualAttentionGateTransformer

    def build():
        model = cls(omic_sizes=sizes, bag_dtype=torch.bfloat16)
        model.load_state_dict(syn.fill_state_dict(C.model_shapes(sizes, kind == "nacagat"), 55))
        model.to(dev).eval()
        window = harness.make_window(syn.make_cohort(6, 200, 700, sizes, 56), dev, torch.bfloat16)
        return model, FlatGradBucket(list(model.parameters())), window
    model_a, bucket_a, window_a = build()
    one = harness.GraphedWindowStep(model_a, bucket_a, window_a, 6, opt=None, warmup=1)
    one()
    ref = bucket_a.flat.clone()
    model_b, bucket_b, window_b = build()
    two = harness.GraphedWindowStep(model_b, bucket_b, window_b, 6, opt=None, warmup=1, split_patch_grad=True)
    head = two.head_numel()
    assert head == model_b.H[0].weight.numel()
    bucket_b.flat[:head].fill_(123.0)
    two()                                                           # main graph only
    assert torch.equal(bucket_b.flat[:head], torch.full_like(bucket_b.flat[:head], 123.0))
    torch.testing.assert_close(bucket_b.flat[head:], ref[head:], rtol=0, atol=0)
    two.replay_tail()
    torch.testing.assert_close(bucket_b.flat, ref, rtol=0, atol=0)
    assert not ops._deferred_patch
    ops.set_rng_epoch(None)
