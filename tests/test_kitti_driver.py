"""The KITTI odometry driver (pair selection, velodyne IO, ground truth, success metric) on a synthetic mini sequence
in the reference's directory layout (KITTI/dataset.py:44-118, KITTI/test.py:43-88)."""
import os

import numpy as np
import pytest


def _mini_sequence(root, drive=8, frames=40, step=0.9, seed=0):
    """A straight drive through a synthetic street: odometry file + ring-pattern scans for every frame."""
    from buffer_amd import kitti
    rng = np.random.default_rng(seed)
    boxes = [(rng.uniform(-10, 60), s * rng.uniform(4, 14), rng.uniform(2, 8), rng.uniform(2, 8), rng.uniform(1.5, 6))
             for s in (-1, 1) for _ in range(40)]
    el = np.deg2rad(np.linspace(-24.8, 2.0, 48))
    az = np.linspace(0, 2 * np.pi, 1200, endpoint=False)
    E, A = np.meshgrid(el, az, indexing='ij')
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    os.makedirs(os.path.join(root, 'dataset', 'poses'), exist_ok=True)
    vdir = os.path.join(root, 'dataset', 'sequences', '%02d' % drive, 'velodyne')
    os.makedirs(vdir, exist_ok=True)
    V = kitti.VELO2CAM.T                                             # velodyne -> camera (4x4)
    odo = []
    for f in range(frames):
        yaw = 0.01 * f
        Rw = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
        Tw = np.eye(4)
        Tw[:3, :3], Tw[:3, 3] = Rw, [step * f, 0.02 * f, 0.0]          # velodyne pose in the world
        o = Tw[:3, 3] + np.array([0, 0, 1.73])
        dw = d @ Rw.T
        with np.errstate(divide='ignore', invalid='ignore'):
            t = np.where(dw[:, 2] < 0, -o[2] / dw[:, 2], np.inf)
            for (bx, by, w, l, hh) in boxes:
                lo, hi = np.array([bx - w / 2, by - l / 2, 0.0]), np.array([bx + w / 2, by + l / 2, hh])
                t1, t2 = (lo - o) / dw, (hi - o) / dw
                tn, tf = np.nanmax(np.minimum(t1, t2), 1), np.nanmin(np.maximum(t1, t2), 1)
                t = np.where((tn < tf) & (tn > 0) & (tn < t), tn, t)
        ok = (t > 3) & (t < 70)
        pw = o + dw[ok] * t[ok, None] + rng.normal(scale=0.01, size=(ok.sum(), 3))
        pl = (pw - Tw[:3, 3]) @ Rw                                     # into the velodyne frame (sensor height kept in z)
        scan = np.concatenate([pl, np.zeros((pl.shape[0], 1))], 1).astype(np.float32)
        scan.tofile(os.path.join(vdir, '%06d.bin' % f))
        Tc = V @ Tw @ np.linalg.inv(V)                                 # camera-0 pose, what poses/<dd>.txt stores
        odo.append(Tc[:3].reshape(-1))
    np.savetxt(os.path.join(root, 'dataset', 'poses', '%02d.txt' % drive), np.array(odo))


def test_kitti_pair_selection_and_ground_truth(tmp_path):
    from buffer_amd import kitti
    root = str(tmp_path / 'kitti')
    _mini_sequence(root)
    with pytest.raises(FileNotFoundError):                             # no ICP cache and no explicit permission
        kitti.KittiTestSet(root, drives=(8,)).ground_truth(0)
    ds = kitti.KittiTestSet(root, drives=(8,), allow_odometry_gt=True)
    assert len(ds) == 3 and ds.files[0][:2] == (8, 0)
    for drive, t0, t1 in ds.files:                                     # > 10 m apart, the frame before the first such one
        assert 9.0 < 0.9 * (t1 - t0) <= 10.9
    # ground truth maps scan t0 into scan t1: the velodyne origin of t0 lands 0.9 * (t1 - t0) m behind
    gt = ds.ground_truth(0)
    assert abs(np.linalg.norm(gt[:3, 3]) - 0.9 * (ds.files[0][2] - ds.files[0][1])) < 0.2
    os.makedirs(os.path.join(root, 'icp'), exist_ok=True)
    np.save(os.path.join(root, 'icp', '%d_%d_%d.npy' % ds.files[0]), np.eye(4))
    assert np.array_equal(ds.ground_truth(0), np.eye(4))               # the ICP cache wins when present
    ds.ground_truth(1)
    out = kitti.summarize(ds, np.stack([np.eye(4)] * len(ds)))
    assert out['gt_source'] == {'icp-cache': 1, 'icp-device': 0, 'odometry': len(ds) - 1}


def _pairs_by_full_table(scan_ids, positions):
    """the selection rule stated over the full n x n distance table (KITTI/dataset.py:52-67), for pinning select_pairs"""
    n = len(positions)
    table = np.sqrt(((positions.reshape(1, n, 3) - positions.reshape(n, 1, 3)) ** 2).sum(-1)) > 10
    have, out, t = set(scan_ids), [], min(scan_ids)
    guard = 0
    while t in have and guard < 10 * n:
        guard += 1
        hits = np.where(table[t][t:t + 100])[0]
        if len(hits) == 0:
            t += 1
            continue
        nxt = int(hits[0]) + t - 1
        if nxt in have:
            out.append((t, nxt))
            t = nxt + 1
        else:
            break                                                      # (the reference never terminates here)
    return out


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_kitti_select_pairs_matches_the_full_table_rule(seed):
    """random-walk trajectories with stops (no partner within 100 frames), speed changes and loops"""
    from buffer_amd import kitti
    rng = np.random.default_rng(seed)
    n = 900
    speed = np.abs(rng.normal(0.8, 0.5, n)) * (rng.random(n) > 0.15)
    speed[300:450] = 0.0                                               # a long stop: more than `window` frames without progress
    heading = np.cumsum(rng.normal(0, 0.05, n))
    pos = np.cumsum(np.stack([speed * np.cos(heading), np.zeros(n), speed * np.sin(heading)], 1), 0)
    ids = list(range(n))
    got = kitti.select_pairs(ids, pos)
    assert got == _pairs_by_full_table(ids, pos) and len(got) > 20
    assert all(np.linalg.norm(pos[b] - pos[a]) <= 10 < np.linalg.norm(pos[b + 1] - pos[a]) for a, b in got)


@pytest.mark.gpu
def test_kitti_layout_end_to_end(tmp_path, dev):
    from buffer_amd import kitti
    from buffer_amd.config import KITTI
    from buffer_amd.pipeline import BufferPipeline
    import torch
    root = str(tmp_path / 'kitti')
    _mini_sequence(root)
    ds = kitti.KittiTestSet(root, drives=(8,))                         # no cache: ground truth refined by ICP on the device
    pipe = BufferPipeline(KITTI, dev)
    s = ds.item(0, dev)
    pipe.calibrate([{k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in s.items()}])
    poses = kitti.register_pairs(pipe, ds, range(len(ds)), batch=2).cpu().numpy()
    out = kitti.summarize(ds, poses, rte_thresh=0.6, rre_thresh=2.0)
    # a driver test, not a model-quality test: the synthetic street is far sparser than a real scan
    assert out['pairs'] == 3 and out['recall'] >= 2 / 3 and out['te'] < 0.3 and out['re'] < 1.5, out
    assert out['gt_source']['icp-device'] + out['gt_source']['icp-cache'] == 3 and out['gt_source']['odometry'] == 0
    # the refinement stays close to the (exact, synthetic) odometry transform and is read back from the cache afterwards
    odo = kitti.KittiTestSet(root, drives=(8,), allow_odometry_gt=True)
    os.rename(os.path.join(root, 'icp'), os.path.join(root, 'icp_kept'))
    for i in range(3):
        M = odo.ground_truth(i)
        M2 = np.load(os.path.join(root, 'icp_kept', '%d_%d_%d.npy' % ds.files[i]))
        assert np.abs(M2[:3, 3] - M[:3, 3]).max() < 0.1 and np.abs(M2[:3, :3] - M[:3, :3]).max() < 5e-3
