"""BASELINE config #4: KITTI-shape LiDAR pair (ring pattern, 0.05 / 0.30 m voxels, KITTI constants and weights):
device pyramid vs oracle collate, full registration vs the CPU oracle pipeline."""
import os
from dataclasses import replace

import numpy as np
import pytest
import torch

from buffer_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kitti_pair():
    return synth.make_kitti_pair(5)


def test_kitti_pyramid_equals_oracle(kitti_pair, oracle, dev):
    from buffer_amd import pyramid
    from buffer_amd.config import KITTI as cfg
    from oracle import torch_ref as T
    limits = T.calibrate_limits([kitti_pair], cfg.voxel_size_0, cfg.conv_radius)
    assert np.array_equal(limits, pyramid.calibrate_limits([kitti_pair], cfg, dev))
    want = T.collate(kitti_pair, limits, cfg.voxel_size_0, cfg.conv_radius)
    pts, lens, *_ = pyramid.stack_sample(kitti_pair, dev)
    got = pyramid.build_pyramid(pts, lens, limits, cfg)
    for l in range(3):
        assert np.array_equal(got['points'][l].cpu().numpy().view(np.uint32), want['points'][l].numpy().view(np.uint32))
        assert np.array_equal(got['neighbors'][l].cpu().numpy(), want['neighbors'][l].numpy())
        if l < 2:
            assert np.array_equal(got['pools'][l].cpu().numpy(), want['pools'][l].numpy())
            assert np.array_equal(got['upsamples'][l].cpu().numpy(), want['upsamples'][l].numpy())


def test_kitti_registration_matches_cpu_oracle(kitti_pair, dev):
    from buffer_amd.config import KITTI
    from buffer_amd.pipeline import BufferPipeline
    from buffer_amd.weights import load_weights
    from oracle import pipeline_ref
    cfg = replace(KITTI, num_keypts=256)
    pipe = BufferPipeline(cfg, dev)
    limits = pipe.calibrate([kitti_pair])
    rng = np.random.default_rng(0)
    perms = [rng.permutation(len(kitti_pair['src_fds_pts'])), rng.permutation(len(kitti_pair['tgt_fds_pts']))]
    pose, d = pipe.register(pipe.upload(kitti_pair), seed=0, perms=[torch.from_numpy(p).to(dev) for p in perms], detail=True)
    W = {k: torch.from_numpy(v) for k, v in load_weights(cfg.weights).items()}
    want, wd = pipeline_ref.register_pair(kitti_pair, W, limits, cfg, 0, perms)
    for i in range(2):
        assert np.array_equal(d['kpts'][i].cpu().numpy(), wd['kpts'][i].numpy())
    assert np.array_equal(d['s_mids'].cpu().numpy(), wd['s_mids'])
    np.testing.assert_allclose(d['ind'].cpu().numpy(), wd['ind'].numpy(), rtol=1e-4, atol=5e-4)
    # 1e-4 relative on the recovered SE(3): rotation entries to 1e-4, translation to 1e-4 of the scene extent (the scan spans
    # ~80 m; coordinates of that size carry 8e-6 m of fp32 round-off each)
    got = pose.cpu().numpy()
    extent = float(np.abs(kitti_pair['src_fds_pts']).max())
    print('KITTI pose vs oracle: dR', np.abs(got[:3, :3] - want[:3, :3]).max(), 'dt', np.abs(got[:3, 3] - want[:3, 3]).max(), 'extent', extent)
    assert np.abs(got[:3, :3] - want[:3, :3]).max() < 1e-4 and np.abs(got[:3, 3] - want[:3, 3]).max() < 1e-4 * extent


def test_kitti_branch_vs_reference_fixture(dev):
    """HIP point learner and patch embedder with the KITTI constants / weights against fixture F7 (`kitti_tiny.npz`,
    produced by the reference itself: KITTI/config.py, released KITTI snapshot, R = I alignment)."""
    import os
    from buffer_amd.config import KITTI
    from buffer_amd.patch_embedder import PatchEmbedder
    from buffer_amd.point_learner import PointLearner
    from buffer_amd.weights import load_weights
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_tiny.npz"))
    W = load_weights("kitti")
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    pyr = dict(points=[t(f[f'points_{l}']) for l in range(3)], neighbors=[t(f[f'neighbors_{l}'], torch.int32) for l in range(3)],
               pools=[t(f[f'pools_{l}'], torch.int32) for l in range(2)], upsamples=[t(f[f'upsamples_{l}'], torch.int32) for l in range(2)])
    pl = PointLearner(W, dev, scale=KITTI.scale)
    axis, eps, bottle, skips, _ = pl.efcnn(pyr, t(f['features']))
    score = pl.detnet(pyr, bottle, skips)
    a, b = axis.cpu().numpy(), f['axis']
    # 80 m coordinates in fp32, features near zero behind VN-BN: the REFERENCE'S OWN run is 1.8e-4 (axis) / 4.3e-4 (eps) /
    # 3.3e-4 (score) of scale away from the float64 network (test_point_learner_error_is_the_fp32_conditioning_of_the_network),
    # the HIP path 2.0e-4 / 6.3e-4 / 3.4e-4 with its fp64 sums; element-wise against the fixture that is 1.4e-3 (eps) and
    # 3.8e-3 (score) relative at the worst element; the bounds below are at most 2 x what is measured (assert_close prints the share used)
    from util import assert_close
    used = float((np.linalg.norm(a - b, axis=1) / (1e-3 * np.linalg.norm(b, axis=1) + 1e-5)).max())
    print(f'TOL KITTI axis: worst row uses {used:.2f} of 1e-3 |axis| + 1e-5')
    assert np.all(np.linalg.norm(a - b, axis=1) < 1e-3 * np.linalg.norm(b, axis=1) + 1e-5)
    assert np.min((a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))) > 1 - 1e-5
    assert_close(eps.cpu().numpy(), f['eps'], 2.9e-3, 5.8e-5, 'KITTI eps')             # sigmoid / softplus after two
    assert_close(score.cpu().numpy(), f['score'], 5e-3, 1e-4, 'KITTI score')       # InstanceNorms over the pair
    assert np.array_equal(score.cpu().numpy() > KITTI.keypts_th, f['score'] > KITTI.keypts_th)
    pe = PatchEmbedder(W, dev, KITTI)
    out = pe(t(f['raw']), t(f['kpts']), t(f['kaxis']), t(f['perm']), want_patches=True)
    np.testing.assert_allclose(out['patches'].cpu().numpy(), f['patches'], rtol=0, atol=3e-6)
    assert np.array_equal(out['R'].cpu().numpy(), f['R'])                                       # identity
    np.testing.assert_allclose(out['rand_axis'].cpu().numpy(), f['rand_axis'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(out['desc'].cpu().numpy(), f['desc'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out['equi'].cpu().numpy(), f['equi'], rtol=1e-4, atol=2e-5)


def test_cross_dataset_presets(kitti_pair, dev):
    """generalization/*/config.py as presets: target-data constants, source-data weights, test.scale = voxel_size_0 / voxel_size_1.
    3DMatch weights on the KITTI-shape pair (generalization/ThreeD2KITTI: scale 10) registers it."""
    from buffer_amd import config as C
    from buffer_amd.pipeline import BufferPipeline
    assert abs(C.THREEDMATCH_TO_KITTI.scale - 10.0) < 1e-9 and abs(C.THREEDMATCH_TO_ETH.scale - 5.0) < 1e-9
    assert abs(C.KITTI_TO_ETH.scale - 0.5) < 1e-9 and abs(C.KITTI_TO_3DLOMATCH.scale - 0.035 / 0.30) < 1e-9
    assert C.THREEDMATCH_TO_KITTI.weights == '3dmatch' and C.KITTI_TO_3DLOMATCH.weights == 'kitti' and not C.THREEDMATCH_TO_ETH.pose_refine
    cfg = replace(C.THREEDMATCH_TO_KITTI, num_keypts=1500)
    pipe = BufferPipeline(cfg, dev)
    pipe.calibrate([kitti_pair])
    pose = pipe.register(pipe.upload(kitti_pair), seed=0).cpu().numpy().astype(np.float64)
    gt = kitti_pair['relt_pose']
    rte = np.linalg.norm(pose[:3, 3] - gt[:3, 3])
    rre = np.degrees(np.arccos(np.clip((np.trace(pose[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1)))
    print('3DMatch->KITTI on the synthetic scan pair: RTE', rte, 'RRE', rre)
    assert np.isfinite(pose).all() and rte < 0.6 and rre < 5.0, (rte, rre)


KITTI_PARITY_PAIRS = [int(x) for x in os.environ.get('BUF_KITTI_PARITY_PAIRS', '0,1,2,3,4,5,6,7,10').split(',')]


def test_kitti_hip_equals_cpu_oracle_pair_by_pair(dev, oracle):
    """Round 5: the KITTI-shape pairs of `bench.py --workload kitti` (synth.make_kitti_pair(2000 + i), 1500
    keypoints, KITTI constants / weights, no refinement) through the HIP path AND through the CPU oracle in worker processes:
    keypoints identical, mutual matches identical (at most 2 per pair where two descriptors tie to fp32 round-off), pose within
    1e-4 (rotation entries; translation relative to the 80 m extent), and the same success flag under BOTH criteria: the
    reference's own (KITTI/test.py:66-72 as coded: RTE < 0.3 m and RRE < 1 deg) and the bench's (RTE < 0.3 m, RRE < 15 deg).
    A pair the bench counts as failed must fail in the oracle too: the failure is the model's on that scan pair, not a kernel's.
    Pair 10 (seed 2010) is the one `bench.py --workload kitti` reports as not registered (profiles/r04_kitti.json: 90/96 = that pair in
    each of 6 steps): HIP and oracle agree on it to 2e-7 -- both return the same pose 178.8 deg / 10 m from the ground truth (the
    ring scan of that synthetic scene is symmetric under the half turn).  profiles/r06_kitti_parity.json holds ALL 16 pairs of the bench
    (BUF_KITTI_PARITY_PAIRS=0,...,15; the round-5 record lacked pair 7 for no reason but an incomplete list: it agrees like the others)."""
    import json
    import subprocess
    import sys
    import tempfile
    from buffer_amd.config import KITTI
    from buffer_amd.evaluate import dgr_success
    from buffer_amd.pipeline import BufferPipeline
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = replace(KITTI, num_keypts=1500)
    ids = KITTI_PARITY_PAIRS
    pipe = BufferPipeline(cfg, dev)
    limits = pipe.calibrate([synth.make_kitti_pair(1000)])          # bench.py's calibration pair
    nproc = min(8, len(ids))
    threads = max(1, (os.cpu_count() or 8) // nproc)
    tmp = tempfile.mkdtemp(prefix='buf_kitti_')
    env = dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    workers = []
    for w in range(nproc):
        out = os.path.join(tmp, f'oracle_{w}.npz')
        cmd = [sys.executable, os.path.join(root, 'tests', 'oracle_worker.py'), '--dataset', 'kitti', '--pairs', ','.join(map(str, ids[w::nproc])),
               '--keypts', '1500', '--threads', str(threads), '--limits', ','.join(map(str, limits)), '--out', out]
        workers.append((subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), out))
    got = {}
    for i in ids:
        s = synth.make_kitti_pair(2000 + i)
        rng = np.random.default_rng(i)
        perms = [rng.permutation(len(s['src_fds_pts'])), rng.permutation(len(s['tgt_fds_pts']))]
        pose, d = pipe.register(pipe.upload(s), seed=i, perms=[torch.from_numpy(p).to(dev) for p in perms], detail=True)
        got[i] = dict(pose=pose.cpu().numpy().astype(np.float64), kp=[k.cpu().numpy() for k in d['kpts']], smids=d['s_mids'].cpu().numpy(),
                      tmids=d['t_mids'].cpu().numpy(), gt=s['relt_pose'], extent=float(np.abs(s['src_fds_pts']).max()))
        # the batched form the bench runs gives the pose of the pair-by-pair call (keyed permutations there: compared on the flags below)
    want = {}
    for p, out in workers:
        log = p.communicate(timeout=1500)[0].decode()
        assert p.returncode == 0, log[-3000:]
        z = np.load(out)
        want.update({k: z[k] for k in z.files})
    rows = []
    for i in ids:
        g = got[i]
        assert np.array_equal(g['kp'][0], want[f'kp0_{i}']) and np.array_equal(g['kp'][1], want[f'kp1_{i}']), f'KITTI pair {i}: keypoints differ'
        sym = len(set(zip(g['smids'].tolist(), g['tmids'].tolist())) ^ set(zip(want[f'smids_{i}'].tolist(), want[f'tmids_{i}'].tolist())))
        wp = want[f'pose_{i}']
        dR, dt = float(np.abs(g['pose'][:3, :3] - wp[:3, :3]).max()), float(np.abs(g['pose'][:3, 3] - wp[:3, 3]).max())
        ref_g, rte, rre = dgr_success(g['pose'], g['gt'], 0.3, 1.0)       # KITTI/test.py:66-72 as coded
        ref_o, rte_o, rre_o = dgr_success(wp, g['gt'], 0.3, 1.0)
        rows.append(dict(pair=i, seed=2000 + i, matches=int(len(g['smids'])), matches_differing=sym, dR=dR, dt=dt, rte=rte, rre=rre,
                         rte_oracle=rte_o, rre_oracle=rre_o, ok_reference_criterion=ref_g, ok_reference_criterion_oracle=ref_o,
                         ok_bench_criterion=dgr_success(g['pose'], g['gt'], 0.3, 15.0)[0],
                         ok_bench_criterion_oracle=dgr_success(wp, g['gt'], 0.3, 15.0)[0], extent=g['extent']))
    print('KITTI_PARITY ' + json.dumps(dict(limits=limits, pairs=rows)))
    for r in rows:
        assert r['matches_differing'] <= 2, r
        assert r['ok_reference_criterion'] == r['ok_reference_criterion_oracle'] and r['ok_bench_criterion'] == r['ok_bench_criterion_oracle'], r
        if r['matches_differing'] == 0:
            assert r['dR'] < 1e-4 and r['dt'] < 1e-4 * r['extent'], r
    assert sum(r['matches_differing'] == 0 for r in rows) >= len(rows) - 2
