"""A4-A16 on the GPU: HIP stages vs the golden vectors produced by the reference and vs the torch
fp32 oracle.  Tolerances: 1e-4 relative on descriptors / poses (BASELINE.json north_star)."""
import os
from dataclasses import replace

import numpy as np
import pytest
import torch

from oracle import torch_ref as T  # noqa: E402  (test-side torch restatement)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.fixture(scope="module")
def W():
    from buffer_amd.weights import load_weights
    return load_weights("3dmatch")


def _on(W, dev):
    """the snapshot as torch tensors on `dev`, for the test-side torch restatement (oracle/torch_ref.py)"""
    return {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in W.items()}


def _pyr_from_golden(g, dev):
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    return dict(points=[t(g[f'points_{l}'], torch.float32) for l in range(3)],
                neighbors=[t(g[f'neighbors_{l}'], torch.int32) for l in range(3)],
                pools=[t(g[f'pools_{l}'], torch.int32) for l in range(2)],
                upsamples=[t(g[f'upsamples_{l}'], torch.int32) for l in range(2)])


def test_point_learner_vs_reference(W, dev):
    """EFCNN + DetNet on the reference's own pyramid tables (fixture F1) -> fixture F3."""
    from buffer_amd.point_learner import PointLearner
    g, f = load("pyramid_tiny.npz"), load("point_learner_tiny.npz")
    pyr = _pyr_from_golden(g, dev)
    pl = PointLearner(W, dev)
    axis, eps, bottle, skips, blocks = pl.efcnn(pyr, torch.from_numpy(f['features']).to(dev))
    score = pl.detnet(pyr, bottle, skips)
    from util import assert_close
    # tolerances = 2 x the worst element measured on this build (round 4; assert_close prints the share used): the fixture is
    # the reference's own fp32 run, one summation order among many (test_point_learner_error_is_the_fp32_conditioning_...)
    for i in range(5):
        assert_close(blocks[i].cpu().numpy(), f[f'block{i}'], 2.6e-4, 2.6e-5, f'VN block {i}')
    assert_close(bottle.cpu().numpy(), f['bottle'], 2.6e-4, 2.6e-5, 'bottleneck')
    assert_close(axis.cpu().numpy(), f['axis'], 3.2e-5, 6.4e-6, 'axis')
    assert_close(eps.cpu().numpy(), f['eps'], 6.4e-5, 1.3e-5, 'eps')
    assert_close(score.cpu().numpy(), f['score'], 5.4e-4, 1.1e-4, 'score')
    # the keypoint decision (score > 0.1) must agree wherever the score is not on the threshold
    s_ref, s_got = f['score'][:, 0], score.cpu().numpy()[:, 0]
    clear = np.abs(s_ref - 0.1) > 1e-3
    assert np.array_equal((s_got > 0.1)[clear], (s_ref > 0.1)[clear])


def test_patch_embedder_vs_reference(W, dev):
    """select_patches + fused voxelise + Cylindrical_Net -> fixture F4."""
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    f = load("desc_tiny.npz")
    pe = PatchEmbedder(W, dev, THREEDMATCH)
    t = lambda a: torch.from_numpy(a).to(dev)
    out = pe(t(f['raw']), t(f['kpts']), t(f['kaxis']), t(f['perm']), want_patches=True)
    np.testing.assert_allclose(out['patches'].cpu().numpy(), f['patches'], rtol=0, atol=3e-6)
    np.testing.assert_allclose(out['R'].cpu().numpy(), f['R'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out['rand_axis'].cpu().numpy(), f['rand_axis'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out['desc'].cpu().numpy(), f['desc'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out['equi'].cpu().numpy(), f['equi'], rtol=1e-4, atol=2e-5)


def test_patch_embedder_split_arithmetic_vs_reference(W, dev):
    """the opt-in split-f16 descriptor CNN (cnn_arith='split', csrc/convnet_h3.hip) against the same fixture F4 at the same tolerance,
    and against the fp32 kernel: descriptors within 2e-6 of each other."""
    from dataclasses import replace
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    f = load("desc_tiny.npz")
    pe = PatchEmbedder(W, dev, replace(THREEDMATCH, cnn_arith='split'))
    assert pe.fused.entry == "buf_cylindrical_net_split"
    t = lambda a: torch.from_numpy(a).to(dev)
    out = pe(t(f['raw']), t(f['kpts']), t(f['kaxis']), t(f['perm']))
    np.testing.assert_allclose(out['desc'].cpu().numpy(), f['desc'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out['equi'].cpu().numpy(), f['equi'], rtol=1e-4, atol=2e-5)
    ref = PatchEmbedder(W, dev, THREEDMATCH)(t(f['raw']), t(f['kpts']), t(f['kaxis']), t(f['perm']))
    assert (out['desc'] - ref['desc']).abs().max().item() < 2e-6
    pe.fused.check_range()
    # the head fused behind the last layer (what embed_patches runs) == the two launches in sequence, bit for bit
    x = out['x']
    d2, e2 = pe.head(pe.fused(x))
    d1, e1 = pe.fused.with_head(x, pe.fused_head)
    assert torch.equal(d1, d2) and torch.equal(e1, e2)
    assert torch.equal(d1, out['desc']) and torch.equal(e1, out['equi'])
    d0, e0 = pe.fused.with_head(x[:0], pe.fused_head)
    assert d0.shape == (0, 32) and e0.shape == (0, 32, 7, 20)
    with pytest.raises(ValueError):
        PatchEmbedder(W, dev, replace(THREEDMATCH, cnn_arith='bf16'))


def test_split_kernel_flags_activations_outside_the_f16_range(W, dev):
    """fail loudly: an activation >= 65504 sets the status word and check_range() raises; other widths run (64 -> 128 -> 32 ...)
    and agree with library convolutions; unsupported widths are rejected on the host."""
    from buffer_amd import _lib, ops
    rng = np.random.default_rng(2)
    widths = [(32, 64), (64, 128), (128, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 32)]
    layers = [((rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32), (rng.standard_normal(co) * 0.1).astype(np.float32), i < 7)
              for i, (ci, co) in enumerate(widths)]
    net = ops.CylindricalNetSplit(layers, dev)
    x = torch.from_numpy(rng.standard_normal((9, 32, 140)).astype(np.float32)).to(dev)
    h = x.double().reshape(-1, 32, 7, 20)
    for w, b, relu in layers:
        h = torch.cat([h[..., -1:], h, h[..., :1]], -1)
        h = torch.nn.functional.pad(h, (0, 0, 1, 1))
        h = torch.nn.functional.conv2d(h, torch.from_numpy(w).double().to(dev), torch.from_numpy(b).double().to(dev))
        h = torch.relu(h) if relu else h
    y = net(x)
    assert (y.double() - h).abs().max().item() < 1e-5 * h.abs().max().item()
    net.check_range()
    net(x * 1e6)
    with pytest.raises(FloatingPointError):
        net.check_range()
    net.check_range()                                                          # the flag was cleared
    bad = list(layers)
    bad[0] = ((rng.standard_normal((64, 80, 3, 3))).astype(np.float32), layers[0][1], True)      # Cin 80: three k-steps, not built
    bad[1] = ((rng.standard_normal((128, 64, 3, 3))).astype(np.float32), layers[1][1], True)
    with pytest.raises(_lib.BufferHipError):
        ops.CylindricalNetSplit(bad, dev)(torch.zeros(1, 80, 140, device=dev))


def test_split_kernels_are_safe_by_construction(W, dev):
    """cnn_arith='split' never returns a value from behind an f16 overflow (csrc/split_safe.hip): patches / matches whose input or
    hidden activations leave the f16 range -- inputs x 1e6 (the input watch), x 3e3 (hidden activations only), NaN, inf -- are flagged on
    the device and recomputed by the fp32 kernel in the same stream: their rows equal the fp32 kernels' BIT FOR BIT, every other row
    equals the split kernel's own result, the flags say exactly which is which, and nothing raises.  With the fused head and without."""
    from buffer_amd import registration
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    pe32 = PatchEmbedder(W, dev, THREEDMATCH)
    pes = PatchEmbedder(W, dev, replace(THREEDMATCH, cnn_arith='split'))
    assert pes.fused.safe
    g = torch.Generator(device='cpu').manual_seed(9)
    x = torch.rand((41, 16, 420), generator=g).to(dev)
    clean = pes.fused(x)
    assert pes.fused.range_fallbacks() == 0
    dclean, eclean = pes.fused.with_head(x, pes.fused_head)
    bad = x.clone()
    bad[3] *= 1e6                      # input beyond the f16 range
    bad[17] *= 5e4                     # input inside the range, hidden activations possibly beyond
    bad[18, 5, 77] = float('nan')
    bad[40, 0, 0] = float('inf')
    y32 = pe32.fused(bad)
    d32, e32 = pe32.head(y32)
    y = pes.fused(bad)
    rows = sorted(np.nonzero(pes.fused.last_flags.cpu().numpy())[0].tolist())
    assert set(rows) >= {3, 18, 40} and set(rows) <= {3, 17, 18, 40} and pes.fused.range_fallbacks() == len(rows)
    keep = torch.tensor([i for i in range(41) if i not in rows], device=dev)
    rr = torch.tensor(rows, device=dev)
    assert torch.equal(y[keep], clean[keep])
    assert torch.equal(y[rr].view(torch.int32), y32[rr].view(torch.int32))           # bitwise, NaN patterns included
    assert float(y32[3].abs().max()) > 65504.0 and bool(torch.isfinite(y[3]).all())   # a real overflow case, a finite fp32 result
    d, e = pes.fused.with_head(bad, pes.fused_head)
    assert sorted(np.nonzero(pes.fused.last_flags.cpu().numpy())[0].tolist()) == rows
    assert torch.equal(d[keep], dclean[keep]) and torch.equal(e[keep], eclean[keep])
    assert torch.equal(d[rr].view(torch.int32), d32[rr].view(torch.int32)) and torch.equal(e[rr].view(torch.int32), e32[rr].view(torch.int32))
    pes.fused.check_range()                                                           # nothing to raise
    assert int(pes.fused.status.item()) == 0
    # hidden activations only: inputs in [0, 1), layer-0 filters x 3e4 (still f16 numbers) -> the watch of the layer epilogues trips
    from buffer_amd import ops
    big = [(w * np.float32(3e4) if i == 0 else w, b, r) for i, (w, b, r) in enumerate(pe32.layers)]
    ns, n32 = ops.CylindricalNetSplit(big, dev), ops.CylindricalNet(big, dev)
    assert ns.safe
    ys, y3 = ns(x), n32(x)
    hit = torch.nonzero(ns.last_flags[:41]).flatten()
    assert hit.numel() > 0 and torch.equal(ys[hit].view(torch.int32), y3[hit].view(torch.int32))
    assert (ys - y3).abs().max().item() <= 1e-5 * y3.abs().max().item()
    # the cost net: dense and gathered forms
    cv32 = registration.CostVolume(W, dev)
    cvs = registration.CostVolume(W, dev, arith='split')
    assert cvs.fused.safe
    a = torch.nn.functional.normalize(torch.rand((37, 32, 5, 20), generator=g), dim=1).to(dev)
    b = torch.nn.functional.normalize(torch.rand((37, 32, 5, 20), generator=g), dim=1).to(dev)
    ind_clean = cvs(a, b)
    assert cvs.fused.range_fallbacks() == 0
    a2, b2 = a.clone(), b.clone()
    a2[5] *= 1e6
    b2[20] *= 3e4
    a2[36, 1, 2, 3] = float('nan')
    want = cv32(a2, b2)
    got = cvs(a2, b2)
    fl = sorted(np.nonzero(cvs.fused.last_flags.cpu().numpy())[0].tolist())
    assert set(fl) >= {5, 36} and set(fl) <= {5, 20, 36}
    keep = torch.tensor([i for i in range(37) if i not in fl], device=dev)
    ff = torch.tensor(fl, device=dev)
    assert torch.equal(got[keep], ind_clean[keep]) and torch.equal(got[ff].view(torch.int32), want[ff].view(torch.int32))
    equi = torch.nn.functional.normalize(torch.rand((50, 32, 7, 20), generator=g), dim=1).to(dev)
    equi[7] *= 1e6
    s_rows = torch.arange(0, 40, device=dev)
    t_rows = torch.arange(10, 50, device=dev)
    got = cvs.gathered(equi, s_rows, t_rows)
    want = cv32.gathered(equi, s_rows, t_rows)
    fl = np.nonzero(cvs.fused.last_flags.cpu().numpy())[0].tolist()
    assert fl == [7] and torch.equal(got[7].view(torch.int32), want[7].view(torch.int32))
    assert (got - want).abs().max().item() < 2e-4
    cvs.fused.check_range()


def test_split_pipeline_survives_an_overflowing_pair(W, dev):
    """BufferPipeline(cnn_arith='split') on a pair whose descriptor CNN overflows the f16 range in some patches (the point MLP's folded
    BatchNorm scaled so that a third of the voxel features exceed 65504): no exception, the pose is the fp32 pipeline's to round-off,
    and range_fallbacks() reports the re-run patches."""
    from buffer_amd import synth
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.pipeline import BufferPipeline
    sample = synth.make_pair(11, n_raw=40_000, size=(1.0, 1.0, 0.9), n_boxes=3)
    cfg = replace(THREEDMATCH, num_keypts=256)
    p32 = BufferPipeline(cfg, dev)
    ps = BufferPipeline(replace(cfg, cnn_arith='split'), dev, limits=p32.calibrate([sample]))
    inp = ps.upload(sample)
    pose32 = p32.register(inp, seed=0)
    pose_s = ps.register(inp, seed=0)
    assert ps.range_fallbacks() == (0, 0)
    assert (pose32 - pose_s).abs().max().item() < 1e-4
    for pipe in (p32, ps):                                   # the same (absurd) point-MLP scale in both pipelines
        pipe.desc.mlp_s = pipe.desc.mlp_s * np.float32(3e5)
        pipe.desc.mlp_t = pipe.desc.mlp_t * np.float32(3e5)
    pose32b = p32.register(inp, seed=0)
    pose_sb = ps.register(inp, seed=0)                       # rounds 4-5 raised FloatingPointError here
    assert ps.range_fallbacks()[0] > 0
    assert torch.isfinite(pose_sb).all() and (pose32b - pose_sb).abs().max().item() < 1e-4


def test_voxelize_vs_oracle_spt(W, dev):
    """fused A9+A10+point-MLP == oracle SPT [P,420,10,3] -> conv1x1+BN+ReLU -> max (incl. the zero-slot quirks)."""
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    from oracle import torch_ref as T
    f = load("desc_tiny.npz")
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    pe = PatchEmbedder(W, dev, THREEDMATCH)
    t = lambda a: torch.from_numpy(a).to(dev)
    out = pe(t(f['raw']), t(f['kpts']), t(f['kaxis']), t(f['perm']), want_patches=True)
    with torch.no_grad():
        inv = T.spt(torch.from_numpy(f['patches']))
        np.testing.assert_allclose(inv[:8].numpy(), f['spt_first8'], rtol=0, atol=2e-6)
        want = T.point_mlp_max(inv, Wt).numpy()
    np.testing.assert_allclose(out['x'].cpu().numpy(), want, rtol=1e-4, atol=1e-5)


def test_matching_and_pose_vs_reference(W, dev):
    """mutual 1-NN, cost volume, hypotheses, scoring, refinement -> fixture F5."""
    from buffer_amd import ops, registration
    f = load("match_tiny.npz")
    t = lambda a: torch.from_numpy(a).to(dev)
    s_mids, t_mids = registration.mutual_matching(t(f['src_desc']), t(f['tgt_desc']))
    assert np.array_equal(s_mids.cpu().numpy(), f['s_mids']) and np.array_equal(t_mids.cpu().numpy(), f['t_mids'])
    cv = registration.CostVolume(W, dev)
    se, te = t(f['src_equi'])[s_mids][:, :, 1:6].contiguous(), t(f['tgt_equi'])[t_mids][:, :, 1:6].contiguous()
    ind = cv(se, te)
    np.testing.assert_allclose(ind.cpu().numpy(), f['ind'], rtol=1e-4, atol=2e-4)
    ss, tt = t(f['src_kpts'])[s_mids].contiguous(), t(f['tgt_kpts'])[t_mids].contiguous()
    R, tr, num, best, mask = ops.hypotheses_score(t(f['ind']), ss, tt, t(f['src_R'])[s_mids].contiguous(),
                                                  t(f['tgt_R'])[t_mids].contiguous())
    np.testing.assert_allclose(R.cpu().numpy(), f['R_hyp'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(tr.cpu().numpy(), f['t_hyp'], rtol=0, atol=1e-5)
    dnum = np.abs(num.cpu().numpy() - f['inlier_num'])
    print('INLIER_NUM max |diff|', int(dnum.max()), 'hypotheses differing', int((dnum > 0).sum()), 'of', dnum.size)
    assert dnum.max() == 0                         # (rounds 1-3 allowed one borderline residual: it does not fire on the committed fixture)
    assert int(best.item()) == int(f['best'])
    assert np.array_equal(np.nonzero(mask.cpu().numpy())[0], f['inlier_ind'])
    T, info = ops.post_refine(t(f['init_pose']), ss, tt, 0.10, 20)
    np.testing.assert_allclose(T.cpu().numpy(), f['refined_pose'][0], rtol=0, atol=2e-5)


def test_ransac_recovers_planted_pose(dev):
    """A15: deterministic GPU RANSAC finds a planted rigid motion among 40 % outliers and is reproducible."""
    from buffer_amd import ops, synth
    rng = np.random.default_rng(4)
    n = 400
    src = rng.normal(size=(n, 3)).astype(np.float32)
    R = synth.random_rotation(rng)
    tvec = rng.normal(size=3)
    tgt = (src @ R.T + tvec + rng.normal(scale=0.01, size=(n, 3))).astype(np.float32)
    out = rng.permutation(n)[:160]
    tgt[out] = rng.normal(size=(160, 3)).astype(np.float32) * 2
    t = lambda a: torch.from_numpy(a).to(dev)
    corr = torch.arange(n, dtype=torch.int32, device=dev)
    T1, info1 = ops.ransac_kabsch(t(src), t(tgt), corr, nhyp=4096, seed=1, max_dist=0.1, edge_similarity=0.8)
    T2, _ = ops.ransac_kabsch(t(src), t(tgt), corr, nhyp=4096, seed=1, max_dist=0.1, edge_similarity=0.8)
    assert torch.equal(T1, T2)
    T1 = T1.cpu().numpy()
    assert int(info1[0]) >= 200
    assert np.abs(T1[:3, :3] - R).max() < 0.05 and np.abs(T1[:3, 3] - tvec).max() < 0.05
    Tr, _ = ops.post_refine(t(T1), t(src), t(tgt), 0.1, 20)
    Tr = Tr.cpu().numpy()
    assert np.abs(Tr[:3, :3] - R).max() < 5e-3 and np.abs(Tr[:3, 3] - tvec).max() < 5e-3


def test_fused_cylindrical_net_vs_library_convs(W, dev):
    """csrc/convnet_wg.hip (fp32 MFMA in the Winograd domain, padding in the LDS layout) == torch convolutions with explicit padding."""
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    pe = PatchEmbedder(W, dev, THREEDMATCH)
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.rand((37, 16, 420), generator=g).to(dev)
    want = T.cylindrical_net(x.view(-1, 16, 3, 7, 20), _on(W, dev))      # library convolutions (test-side restatement)
    got = pe.fused(x)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() < 2e-5 * max(scale, 1.0)


def test_winograd_and_direct_forms_against_float64(W, dev):
    """csrc/convnet_wg.hip (Winograd F(2x2,3x3), the product's kernel) and tests/native/convnet_direct.hip (direct form) against the stack in
    float64: both stay within 1e-5 of the output scale on the released weights, on non-negative (post-ReLU-like) and on signed
    inputs, and agree with each other; a patch's result does not depend on its position in the batch."""
    from buffer_amd import ops
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    pe = PatchEmbedder(W, dev, THREEDMATCH)
    assert pe.fused.entry == "buf_cylindrical_net_wg"
    from util import DirectCylindricalNet
    direct = DirectCylindricalNet(pe.layers, dev)                       # tests/native: the round-1/2 kernel, test infrastructure now
    split = ops.CylindricalNetSplit(pe.layers, dev)                     # csrc/convnet_h3.hip: fp32-equivalent on the f16 matrix pipe (opt-in)
    g = torch.Generator(device='cpu').manual_seed(5)
    for signed in (False, True):
        x = torch.rand((70, 16, 420), generator=g)
        if signed:
            x = x * 2 - 1
        x = x.to(dev)
        h = x.double().reshape(-1, 48, 7, 20)
        for w, b, relu in pe.layers:                                         # circular azimuth, zero elevation, float64
            h = torch.cat([h[..., -1:], h, h[..., :1]], -1)
            h = torch.nn.functional.pad(h, (0, 0, 1, 1))
            h = torch.nn.functional.conv2d(h, torch.from_numpy(w).double().to(dev), torch.from_numpy(b).double().to(dev))
            h = torch.relu(h) if relu else h
        scale = h.abs().max().item()
        yw, yd, ys = pe.fused(x), direct(x), split(x)
        ew, ed, es = ((v.double() - h).abs().max().item() / scale for v in (yw, yd, ys))
        print(f'signed={signed}: error vs float64 / output scale: winograd fp32 {ew:.2e}, direct fp32 {ed:.2e}, split f16 {es:.2e}')
        assert ew < 1e-5 and ed < 1e-5
        assert (yw - yd).abs().max().item() < 1e-5 * scale
        # the split-f16 form: the SAME bound, and not more than 1.5 x the fp32 product kernel's error (measured: 0.5-0.6 x)
        assert es < 1e-5 and es <= 1.5 * ew
        perm = torch.randperm(70, generator=g).to(dev)
        assert torch.equal(pe.fused(x[perm]), yw[perm])
        assert torch.equal(split(x[perm]), ys[perm])
    split.check_range()                                                      # no activation left the f16 range


@pytest.mark.parametrize("head", ["eps_head", "key_head"])
def test_fused_score_head_is_bit_identical_to_the_layer_by_layer_path(W, dev, head):
    """buf_score_head (7 launches: VNStdFeature + first Conv1d in one kernel, InstanceNorm statistics in two, the normalisation
    folded into the next Conv1d) against the 17-launch path (buf_vn_pointwise x 3, buf_vn_std, buf_row_linear,
    buf_segment_instance_norm): the same bits, for one pair, for a stacked batch with uneven segments, an EMPTY segment and
    300 segments shorter than a block's 256 rows (more than 8 segments inside one block of the fused Conv1d)."""
    from buffer_amd.point_learner import PointLearner
    pl = PointLearner(W, dev)
    h = getattr(pl, head)
    g = torch.Generator(device='cpu').manual_seed(4)
    for n, seg in ((9000, None), (20011, [7000, 0, 9011, 4000]), (6000, [20] * 300), (1, None), (0, None)):
        x = torch.randn((n, 30), generator=g).to(dev)
        h.fused = True
        got = h(x, seg)
        h.fused = False
        want = h(x, seg)
        h.fused = True
        assert got.shape == want.shape == (n, 1)
        assert torch.equal(got, want), (head, n, float((got - want).abs().max()))


def _stack64(x, layers, dev):
    h = x.double().reshape(-1, layers[0][0].shape[1], 7, 20)
    for w, b, relu in layers:                                                # circular azimuth, zero elevation, float64
        h = torch.cat([h[..., -1:], h, h[..., :1]], -1)
        h = torch.nn.functional.pad(h, (0, 0, 1, 1))
        h = torch.nn.functional.conv2d(h, torch.from_numpy(np.ascontiguousarray(w)).double().to(dev), torch.from_numpy(np.ascontiguousarray(b)).double().to(dev))
        h = torch.relu(h) if relu else h
    return h


@pytest.mark.parametrize("case", ["wide_inputs", "sparse_maps", "wide_weights", "tiny_everything"])
def test_cnn_kernels_with_wide_dynamic_range_inside_one_contraction(W, dev, case):
    """both descriptor-CNN kernels against the float64 stack where ONE contraction mixes operands of very
    different magnitude -- inputs spread over 1e-6 .. 1e3 element by element, maps with 90 % exact zeros (post-ReLU sparsity),
    synthetic weights spread over 1e-6 .. 3e-2, and a stack whose activations sit in the f16 SUBNORMAL range (inputs ~1e-6: the
    `hi` plane of the split form is subnormal there and the scaled low part carries the value).  Same bound as the unit-range test:
    error < 1e-5 of the output scale for both, split-f16 not more than 1.5 x the fp32 kernel's error."""
    from buffer_amd import ops
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    pe = PatchEmbedder(W, dev, THREEDMATCH)
    g = torch.Generator(device='cpu').manual_seed(11)
    layers = pe.layers
    x = torch.relu(torch.randn((64, 48, 140), generator=g))
    if case == "wide_inputs":
        x = x * 10.0 ** (torch.rand((64, 48, 140), generator=g) * 9 - 6)          # 1e-6 .. 1e3 inside every 3 x 3 x 48 window
    elif case == "sparse_maps":
        x = x * (torch.rand((64, 48, 140), generator=g) < 0.1) * 10.0 ** (torch.rand((64, 48, 140), generator=g) * 4 - 2)
    elif case == "tiny_everything":
        x = x * 1e-6                                                               # every activation below the f16 normal range
    elif case == "wide_weights":
        rng = np.random.default_rng(3)
        layers = []
        for (w, b, relu) in pe.layers:
            mag = 10.0 ** rng.uniform(-6, np.log10(3e-2), size=w.shape)
            layers.append(((mag * rng.choice([-1.0, 1.0], size=w.shape)).astype(np.float32),
                           (10.0 ** rng.uniform(-6, -2, size=b.shape)).astype(np.float32), relu))
        x = x * 10.0 ** (torch.rand((64, 48, 140), generator=g) * 5 - 2)          # 1e-2 .. 1e3
    x = x.to(dev)
    h = _stack64(x, layers, dev)
    scale = h.abs().max().item()
    wg, split = ops.CylindricalNet(layers, dev), ops.CylindricalNetSplit(layers, dev)
    ew = (wg(x).double() - h).abs().max().item() / scale
    es = (split(x).double() - h).abs().max().item() / scale
    split.check_range()
    print(f'{case}: output scale {scale:.3e}; error vs float64 / scale: fp32 Winograd kernel {ew:.2e}, split-f16 kernel {es:.2e}')
    assert ew < 1e-5 and es < 1e-5
    assert es <= 1.5 * ew + 1e-7


def test_fused_descriptor_head_vs_library(W, dev):
    """k_desc_head (attention pooling + both normalisations in one launch) == the torch restatement
    (patch_embedder.py:81-84) on conv-net outputs, on an all-zero map (eps clamps) and on P = 0."""
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    pe = PatchEmbedder(W, dev, THREEDMATCH)
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.rand((41, 16, 420), generator=g).to(dev)
    y = pe.fused(x)
    y[5] = 0.0
    want_d, want_e = T.desc_head(y, _on(W, dev))
    got_d, got_e = pe.head(y)
    np.testing.assert_allclose(got_d.cpu().numpy(), want_d.cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got_e.cpu().numpy(), want_e.cpu().numpy(), rtol=1e-4, atol=1e-6)
    d0, e0 = pe.head(y[:0])
    assert d0.shape == (0, 32) and e0.shape == (0, 32, 7, 20)


def test_fused_cost_volume_vs_library_convs(W, dev):
    """csrc/costnet.hip (cost volume never materialised, sliding-window layer 0/1 fusion) == torch convolutions."""
    from buffer_amd import registration
    cv = registration.CostVolume(W, dev)
    f = load("match_tiny.npz")
    se = torch.from_numpy(f['src_equi'])[f['s_mids']][:, :, 1:6].contiguous().to(dev)
    te = torch.from_numpy(f['tgt_equi'])[f['t_mids']][:, :, 1:6].contiguous().to(dev)
    Wd = _on(W, dev)
    want = T.cost_volume(se, te, Wd)
    got = cv(se, te)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(got.cpu().numpy(), f['ind'], rtol=1e-4, atol=2e-4)
    g = torch.Generator(device='cpu').manual_seed(1)
    a = torch.nn.functional.normalize(torch.rand((70, 32, 5, 20), generator=g), dim=1).to(dev)
    b = torch.nn.functional.normalize(torch.rand((70, 32, 5, 20), generator=g), dim=1).to(dev)
    np.testing.assert_allclose(cv(a, b).cpu().numpy(), T.cost_volume(a, b, Wd).cpu().numpy(), rtol=1e-4, atol=2e-4)
    with pytest.raises(ValueError):                     # no library fallback: other geometries are rejected
        cv(a[:, :, :4].contiguous(), b[:, :, :4].contiguous())
    with pytest.raises(NotImplementedError):
        registration.CostVolume(W, dev, azi_n=18)


def test_split_cost_volume_vs_reference_and_float64(W, dev):
    """the opt-in split-f16 cost net (cnn_arith='split', csrc/costnet_h3.hip): `ind` against fixture F5 and library convolutions at
    the fp32 kernel's tolerance; against the network in float64 its error is below 1e-4 and not above 1.5 x the fp32 kernel's
    (measured 0.5 x); gather form == dense form bitwise; other geometries are rejected."""
    from buffer_amd import registration
    cv32, cvs = registration.CostVolume(W, dev), registration.CostVolume(W, dev, arith='split')
    f = load("match_tiny.npz")
    se = torch.from_numpy(f['src_equi'])[f['s_mids']][:, :, 1:6].contiguous().to(dev)
    te = torch.from_numpy(f['tgt_equi'])[f['t_mids']][:, :, 1:6].contiguous().to(dev)
    got = cvs(se, te)
    np.testing.assert_allclose(got.cpu().numpy(), f['ind'], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(got.cpu().numpy(), T.cost_volume(se, te, _on(W, dev)).cpu().numpy(), rtol=1e-4, atol=2e-4)
    g = torch.Generator(device='cpu').manual_seed(1)
    a = torch.nn.functional.normalize(torch.rand((150, 32, 5, 20), generator=g), dim=1).to(dev)
    b = torch.nn.functional.normalize(torch.rand((150, 32, 5, 20), generator=g), dim=1).to(dev)
    W64 = {k: torch.from_numpy(np.asarray(v)).double().to(dev) for k, v in W.items() if k.startswith('Inlier')}
    ref = T.cost_volume(a.double(), b.double(), W64)
    e32 = (cv32(a, b).double() - ref).abs().max().item()
    es = (cvs(a, b).double() - ref).abs().max().item()
    print(f'cost net, |ind - ind(float64)| (ind in [0, 20)): fp32 kernel {e32:.2e}, split f16 {es:.2e}')
    assert es < 1e-4 and es <= max(1.5 * e32, 1e-5)
    equi = torch.nn.functional.normalize(torch.rand((300, 32, 7, 20), generator=g), dim=1).to(dev)
    s_rows = torch.randint(0, 300, (157,), generator=g).to(dev)
    t_rows = torch.randint(0, 300, (157,), generator=g).to(dev)
    want = cvs(equi[s_rows][:, :, 1:6].contiguous(), equi[t_rows][:, :, 1:6].contiguous())
    assert torch.equal(cvs.gathered(equi, s_rows, t_rows), want)
    assert cvs.gathered(equi, s_rows[:0], t_rows[:0]).shape == (0,)
    cvs.fused.check_range()
    with pytest.raises(ValueError):
        cvs(a[:, :, :4].contiguous(), b[:, :, :4].contiguous())
    with pytest.raises(ValueError):
        registration.CostVolume(W, dev, arith='bf16')


def _truth64(pyr_npz, feats, Wnp, scale=1.0):
    """the same network evaluated in float64 on the CPU (oracle/torch_ref.py is dtype-agnostic): the reference point for
    separating fp32 conditioning from implementation error"""
    W64 = {k: torch.from_numpy(np.asarray(v)).double() for k, v in Wnp.items()}
    z = torch.zeros((0, 1), dtype=torch.long)
    b = dict(points=[torch.from_numpy(pyr_npz[f'points_{l}']).double() for l in range(3)],
             neighbors=[torch.from_numpy(pyr_npz[f'neighbors_{l}']).long() for l in range(3)],
             pools=[torch.from_numpy(pyr_npz[f'pools_{l}']).long() for l in range(2)] + [z],
             upsamples=[torch.from_numpy(pyr_npz[f'upsamples_{l}']).long() for l in range(2)] + [z],
             features=torch.from_numpy(feats).double(), stack_lengths=[torch.from_numpy(pyr_npz[f'lengths_{l}']) for l in range(3)])
    with torch.no_grad():
        axis, eps, bottle, skips = T.efcnn_forward(b, W64, scale)
        score = T.detnet_forward(b, bottle, skips, W64)
    return dict(axis=axis.numpy(), eps=eps.numpy(), bottle=bottle.numpy(), score=score.numpy())


@pytest.mark.parametrize("which", ["3dmatch", "kitti"])
def test_point_learner_error_is_the_fp32_conditioning_of_the_network(which, dev):
    """Why the element-wise tolerances of the point-learner tests sit above 1e-4: the REFERENCE'S OWN fp32 outputs (fixtures
    F3 / F7) differ from the exact (float64) network by up to 5e-4 element-wise (values near zero after InstanceNorm), i.e.
    the fixture is one fp32 summation order among many.  The bound that holds: measured against the float64 network and
    normalised by the tensor's scale, the HIP path is as accurate as the reference's run (within 1.6x) and below 1e-4
    (below 1.6 times the reference's own error where that already exceeds 3e-5: the KITTI branch at 80 m coordinates)."""
    from buffer_amd.config import KITTI, THREEDMATCH
    from buffer_amd.point_learner import PointLearner
    from buffer_amd.weights import load_weights
    cfg = KITTI if which == 'kitti' else THREEDMATCH
    g = load('kitti_tiny.npz' if which == 'kitti' else 'pyramid_tiny.npz')
    f = g if which == 'kitti' else load('point_learner_tiny.npz')
    Wn = load_weights(which)
    truth = _truth64(g, f['features'], Wn, cfg.scale)
    pl = PointLearner(Wn, dev, cfg.scale)
    axis, eps, bottle, skips, _ = pl.efcnn(_pyr_from_golden(g, dev), torch.from_numpy(f['features']).to(dev))
    score = pl.detnet(_pyr_from_golden(g, dev), bottle, skips)
    got = dict(axis=axis.cpu().numpy(), eps=eps.cpu().numpy(), score=score.cpu().numpy())
    for k in ('axis', 'eps', 'score'):
        scale = np.abs(truth[k]).max()
        e_ref = np.abs(f[k] - truth[k]).max() / scale                # the reference's own fp32 run vs the exact network
        e_hip = np.abs(got[k] - truth[k]).max() / scale
        print(f'{which} {k}: reference fp32 vs fp64 {e_ref:.2e}, HIP vs fp64 {e_hip:.2e}, HIP vs reference '
              f'{np.abs(got[k] - f[k]).max() / scale:.2e} (of the tensor scale {scale:.3g})')
        # (KITTI: 80 m coordinates put the reference's own run at ~2e-4 of the exact axis; 1e-4 is not available to anyone there)
        # measured (round 3, fp64 sums in the VN kernels): 3DMatch HIP 0.4-0.5x the reference's error; KITTI 1.0-1.5x
        assert e_hip <= max(1.6 * e_ref, 2e-5) and e_hip < max(1e-4, 1.6 * e_ref), (k, e_ref, e_hip)
        assert np.abs(got[k] - f[k]).max() <= max(1e-4, 3 * e_ref) * scale


def test_point_learner_honours_test_scale(W, dev):
    """test.scale = voxel_size_0 / voxel_size_1 != 1 (generalization/*/config.py, SURVEY Appendix D: 5 for 3DMatch -> ETH):
    the neighbour offsets are divided by it in every block (models/point_learner.py:332-343).  HIP vs the torch restatement."""
    from buffer_amd.point_learner import PointLearner
    g, f = load("pyramid_tiny.npz"), load("point_learner_tiny.npz")
    for scale in (5.0, 0.5):
        truth = _truth64(g, f['features'], W, scale)
        pl = PointLearner(W, dev, scale)
        axis, eps, bottle, skips, _ = pl.efcnn(_pyr_from_golden(g, dev), torch.from_numpy(f['features']).to(dev))
        score = pl.detnet(_pyr_from_golden(g, dev), bottle, skips)
        for k, v in (('axis', axis), ('eps', eps), ('score', score)):
            sc = np.abs(truth[k]).max()
            assert np.abs(v.cpu().numpy() - truth[k]).max() <= 1e-4 * sc, (scale, k)
        assert np.abs(axis.cpu().numpy() - f['axis']).max() > 1e-3          # and it is not the scale = 1 answer


def test_cost_volume_gather_form_equals_dense_form(W, dev):
    """buf_cost_volume_net_gather (row gather + elevation rows 1..5 read inside the kernel) == gathering and slicing first"""
    from buffer_amd import registration
    cv = registration.CostVolume(W, dev)
    g = torch.Generator(device='cpu').manual_seed(4)
    equi = torch.nn.functional.normalize(torch.rand((300, 32, 7, 20), generator=g), dim=1).to(dev)
    s_rows = torch.randint(0, 300, (157,), generator=g).to(dev)
    t_rows = torch.randint(0, 300, (157,), generator=g).to(dev)
    want = cv(equi[s_rows][:, :, 1:6].contiguous(), equi[t_rows][:, :, 1:6].contiguous())
    got = cv.gathered(equi, s_rows, t_rows)
    assert torch.equal(got, want)
    assert cv.gathered(equi, s_rows[:0], t_rows[:0]).shape == (0,)
