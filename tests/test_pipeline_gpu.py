"""End-to-end: device pyramid vs oracle collate, full registration of synthetic pairs."""
import numpy as np
import pytest
import torch

from buffer_amd import synth
from util import assert_neighbors_equal_mod_ties as nbr_eq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    return synth.make_pair(11, n_raw=40_000, size=(1.0, 1.0, 0.9), n_boxes=3)


def test_device_pyramid_equals_oracle_collate(tiny, oracle, dev):
    """A0-A3 composed on device == oracle collate (same row order: both ascending voxel key)."""
    from buffer_amd import pyramid
    from buffer_amd.config import THREEDMATCH as cfg
    from oracle import torch_ref as T
    limits = T.calibrate_limits([tiny])
    got_limits = pyramid.calibrate_limits([tiny], cfg, dev)
    assert np.array_equal(limits, got_limits)
    want = T.collate(tiny, limits)
    pts, lens, *_ = pyramid.stack_sample(tiny, dev)
    got = pyramid.build_pyramid(pts, lens, limits, cfg)
    for l in range(3):
        assert np.array_equal(got['lengths'][l], want['stack_lengths'][l].numpy())
        assert np.array_equal(got['points'][l].cpu().numpy().view(np.uint32), want['points'][l].numpy().view(np.uint32))
        assert np.array_equal(got['neighbors'][l].cpu().numpy(), want['neighbors'][l].numpy())
        if l < 2:
            assert np.array_equal(got['pools'][l].cpu().numpy(), want['pools'][l].numpy())
            assert np.array_equal(got['upsamples'][l].cpu().numpy(), want['upsamples'][l].numpy())


def test_register_tiny_pair(tiny, dev):
    from buffer_amd.pipeline import BufferPipeline
    pipe = BufferPipeline(device=dev)
    pipe.calibrate([tiny])
    inp = pipe.upload(tiny)
    pose, d = pipe.register(inp, seed=0, detail=True)
    pose2 = pipe.register(inp, seed=0)
    assert torch.equal(pose, pose2), "registration must be deterministic for a fixed seed"
    gt = tiny['relt_pose']
    T = pose.cpu().numpy().astype(np.float64)
    rte = np.linalg.norm(T[:3, 3] - gt[:3, 3])
    rre = np.degrees(np.arccos(np.clip((np.trace(T[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1)))
    assert rte < 0.05 and rre < 2.0, (rte, rre)


def test_register_3dmatch_shape_pair(dev):
    from buffer_amd.pipeline import BufferPipeline
    s = synth.make_pair(1)
    pipe = BufferPipeline(device=dev)
    pipe.calibrate([s])
    pose = pipe.register(pipe.upload(s), seed=0)
    gt = s['relt_pose']
    T = pose.cpu().numpy().astype(np.float64)
    rte = np.linalg.norm(T[:3, 3] - gt[:3, 3])
    rre = np.degrees(np.arccos(np.clip((np.trace(T[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1)))
    assert rte < 0.3 and rre < 15.0, (rte, rre)       # the DGR success criterion of ThreeDMatch/test.py:264-270


def test_batched_registration_equals_single(tiny, dev):
    """register_batch (stacked launches) reproduces register() pair by pair."""
    from buffer_amd.pipeline import BufferPipeline
    from dataclasses import replace
    from buffer_amd.config import THREEDMATCH
    other = synth.make_pair(12, n_raw=40_000, size=(1.0, 1.0, 0.9), n_boxes=3)
    pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=300), dev)
    pipe.calibrate([tiny])
    inps = [pipe.upload(tiny), pipe.upload(other), pipe.upload(tiny)]
    single = [pipe.register(x, seed=i) for i, x in enumerate(inps)]
    batch = pipe.register_batch(inps, seeds=[0, 1, 2])
    for a, b in zip(single, batch):
        assert torch.equal(a, b)            # same arithmetic per pair, whatever the stacking (was atol 2e-4 in round 1)


def test_batch_with_a_keypoint_free_pair_keeps_pinned_permutations(tiny, dev):
    """register_batch's rare path (a cloud without any score above the threshold -> identity for that pair, the others
    redone one by one) must hand the caller's permutations on, so that the healthy pairs equal register() exactly."""
    from buffer_amd.pipeline import BufferPipeline
    from dataclasses import replace
    from buffer_amd.config import THREEDMATCH
    other = synth.make_pair(12, n_raw=40_000, size=(1.0, 1.0, 0.9), n_boxes=3)
    cfg = replace(THREEDMATCH, num_keypts=200)
    probe = BufferPipeline(cfg, dev)
    probe.calibrate([tiny])
    tops = []
    for s in (tiny, other):
        _, d = probe.register(probe.upload(s), seed=0, detail=True)
        n_src = int(probe.upload(s)['lengths'][0])
        sc = d['score'][:, 0]
        tops.append(min(float(sc[:n_src].max()), float(sc[n_src:].max())))
    lo, hi = sorted(tops)
    if not lo < hi:
        pytest.skip('the two pairs peak at the same score')
    starved = 0 if tops[0] == lo else 1                      # the pair whose weaker cloud stays below the threshold
    pipe = BufferPipeline(replace(cfg, keypts_th=0.5 * (lo + hi)), dev, limits=probe.limits)
    samples = [tiny, other]
    inps = [pipe.upload(s) for s in samples]
    rng = np.random.default_rng(5)
    perms = [[torch.from_numpy(rng.permutation(len(s[k]))).to(dev) for k in ('src_fds_pts', 'tgt_fds_pts')] for s in samples]
    batch = pipe.register_batch(inps, seeds=[3, 4], perms=perms)
    assert torch.equal(batch[starved], torch.eye(4, device=dev))
    healthy = 1 - starved
    want = pipe.register(inps[healthy], seed=[3, 4][healthy], perms=perms[healthy])
    assert torch.equal(batch[healthy], want)


def test_pipelined_batches_equal_batch_by_batch(tiny, dev):
    """register_batches (keypoint stage of batch i+1 on a side stream beside the descriptor stage of batch i) returns
    exactly what register_batch returns batch by batch"""
    from buffer_amd.pipeline import BufferPipeline
    from dataclasses import replace
    from buffer_amd.config import THREEDMATCH
    other = synth.make_pair(12, n_raw=40_000, size=(1.0, 1.0, 0.9), n_boxes=3)
    pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=300), dev)
    pipe.calibrate([tiny])
    a, b = pipe.upload(tiny), pipe.upload(other)
    batches = [[a, b], [b], [b, a, a], [a]]
    seeds = [[0, 1], [2], [3, 4, 5], [6]]
    want = [pipe.register_batch(x, seeds=s) for x, s in zip(batches, seeds)]
    for _ in range(2):
        got = pipe.register_batches(batches, seeds=seeds)
        assert [len(g) for g in got] == [2, 1, 3, 1]
        for g, w in zip(got, want):
            for p, q in zip(g, w):
                assert torch.equal(p, q)
    assert pipe.register_batches([]) == []
