"""Stand-ins for the non-operator packages the reference's model files import (SURVEY 8b last row):
kornia.geometry.conversions, easydict, nibabel.quaternions, and the slice of open3d the inference path touches."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module", autouse=True)
def _installed():
    import buffer_amd.shims as shims
    shims.install()


def test_easydict_attribute_and_nested_access():
    from easydict import EasyDict as edict
    cfg = edict()
    cfg.data = edict()
    cfg.data.voxel_size_0 = 0.035
    cfg.train = {'all_stage': ['Ref', 'Desc'], 'opt': {'lr': 1e-3}}
    assert cfg['data']['voxel_size_0'] == 0.035 and cfg.train.opt.lr == 1e-3 and cfg.train.all_stage[1] == 'Desc'
    cfg.stage = 'test'
    assert cfg['stage'] == 'test' and 'stage' in cfg and dict(cfg)['stage'] == 'test'
    with pytest.raises(AttributeError):
        cfg.missing
    assert edict({'a': [{'b': 1}]}).a[0].b == 1


def test_kornia_angle_axis_matches_rz_on_the_reachable_angles():
    """models/BUFFER.py:295-299 feeds (0, 0, ind*2pi/20 + 1e-6), ind in [0, 19]; the device kernel uses the exact Rz."""
    import kornia.geometry.conversions as Convert
    ind = torch.linspace(0, 19, 400)
    ang = ind * 2 * np.pi / 20 + 1e-6
    aa = torch.zeros((400, 3))
    aa[:, 2] = ang
    R = Convert.angle_axis_to_rotation_matrix(aa)
    c, s = torch.cos(ang), torch.sin(ang)
    want = torch.zeros((400, 3, 3))
    want[:, 0, 0], want[:, 0, 1], want[:, 1, 0], want[:, 1, 1], want[:, 2, 2] = c, -s, s, c, 1.0
    assert (R - want).abs().max().item() < 4e-6
    # first-order branch for theta^2 <= 1e-6, generic axis against the closed-form Rodrigues rotation
    tiny = torch.tensor([[3e-4, -2e-4, 5e-4]])
    Rt = Convert.angle_axis_to_rotation_matrix(tiny)[0]
    assert torch.equal(torch.diagonal(Rt), torch.ones(3)) and Rt[0, 1].item() == -tiny[0, 2].item()
    g = torch.Generator().manual_seed(0)
    aa = torch.randn((64, 3), generator=g, dtype=torch.float64)
    th = aa.norm(dim=1).view(-1, 1, 1)
    k = aa / aa.norm(dim=1, keepdim=True)
    K = torch.zeros((64, 3, 3), dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    want = torch.eye(3, dtype=torch.float64) + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)
    assert (Convert.angle_axis_to_rotation_matrix(aa) - want).abs().max().item() < 1e-5
    with pytest.raises(ValueError):
        Convert.angle_axis_to_rotation_matrix(torch.zeros(3))


def test_nibabel_mat2quat_and_open3d_containers(tmp_path):
    import nibabel.quaternions as nq
    import open3d as o3d
    a = 0.3
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    np.testing.assert_allclose(nq.mat2quat(R), [np.cos(a / 2), 0, 0, np.sin(a / 2)], atol=1e-12)
    pts = np.random.default_rng(0).random((50, 3)).astype(np.float32)
    pcd = o3d.geometry.PointCloud()
    pcd.points = o3d.utility.Vector3dVector(pts)
    assert np.array(pcd.points).dtype == np.float64 and np.array(pcd.points).shape == (50, 3)
    pcd.paint_uniform_color([1, 0.706, 0])
    assert pcd.has_colors() and not pcd.has_normals()
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, [1, 2, 3]
    moved = np.array(pcd.points) @ R.T + [1, 2, 3]
    pcd.transform(T)
    np.testing.assert_allclose(np.array(pcd.points), moved, atol=1e-12)
    corr = o3d.utility.Vector2iVector(np.array([[0, 1, 2], [0, 1, 2]]).T)
    assert corr.dtype == np.int32 and corr.shape == (3, 2)
    with pytest.raises(RuntimeError):
        o3d.utility.Vector3dVector(np.zeros((4, 2)))
    # orientation towards the sensor; a zero normal becomes the view direction
    pcd.normals = o3d.utility.Vector3dVector(np.tile([[0, 0, 1.0]], (50, 1)))
    pcd._normals[3] = 0
    pcd.orient_normals_towards_camera_location()
    n, p = np.array(pcd.normals), np.array(pcd.points)
    assert ((n * (0 - p)).sum(1) >= 0).all() and abs(np.linalg.norm(n[3]) - 1) < 1e-12
    # PLY reader behind open3d.io.read_point_cloud; a missing file gives an empty cloud like open3d
    from buffer_amd.threedmatch import write_ply
    f = str(tmp_path / 'frag.ply')
    write_ply(f, pts)
    np.testing.assert_array_equal(np.array(o3d.io.read_point_cloud(f).points), pts.astype(np.float64))
    assert not o3d.io.read_point_cloud(str(tmp_path / 'absent.ply')).has_points()


@pytest.mark.gpu
def test_open3d_point_cloud_compute_calls_run_the_device_kernels(dev):
    """the call sequence of ThreeDMatch/dataset.py:91-153 through the stand-in == buffer_amd.preprocess directly"""
    import open3d as o3d
    from buffer_amd import preprocess, synth
    s = synth.make_pair(3, n_raw=60_000, size=(1.2, 1.0, 0.9), n_boxes=3)
    raw = s['src_fds_pts']
    pcd = o3d.geometry.PointCloud()
    pcd.points = o3d.utility.Vector3dVector(raw)
    pcd.paint_uniform_color([1, 0.706, 0])
    fds = o3d.geometry.PointCloud.voxel_down_sample(pcd, voxel_size=0.05)
    sds = o3d.geometry.PointCloud.voxel_down_sample(fds, voxel_size=0.1)
    want_f = preprocess.voxel_down_sample(torch.from_numpy(raw.astype(np.float64)).to(dev), 0.05)
    want_s = preprocess.voxel_down_sample(want_f, 0.1)
    assert np.array_equal(np.array(fds.points), want_f.cpu().numpy()) and np.array_equal(np.array(sds.points), want_s.cpu().numpy())
    kp = o3d.geometry.PointCloud()
    kp.points = o3d.utility.Vector3dVector(np.array(sds.points))
    kp.estimate_normals()
    kp.orient_normals_towards_camera_location()
    want_n = preprocess.estimate_normals(want_s.float(), knn=30, orient=True).cpu().numpy()
    got_n = np.array(kp.normals)
    assert np.abs(np.abs((got_n * want_n).sum(1)) - 1).max() < 1e-5            # same lines
    assert ((got_n * want_n).sum(1) > 0).mean() > 0.999                         # same orientation (ties at 90 degrees aside)


@pytest.mark.gpu
def test_open3d_ransac_call_of_the_reference_forwards_to_the_kernel(dev):
    """models/BUFFER.py:314-326 verbatim call shape -> buf_ransac_kabsch"""
    import open3d as o3d
    from buffer_amd import ops
    reg = o3d.pipelines.registration
    rng = np.random.default_rng(1)
    src = rng.random((400, 3)).astype(np.float32) * 2
    a = 0.4
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    tgt = src @ R.T + np.array([0.3, -0.2, 0.1], np.float32)
    tgt[::3] += rng.normal(scale=0.5, size=tgt[::3].shape).astype(np.float32)       # a third are outliers
    inlier_ind = np.arange(400)
    pcd0, pcd1 = o3d.geometry.PointCloud(), o3d.geometry.PointCloud()
    pcd0.points, pcd1.points = o3d.utility.Vector3dVector(src), o3d.utility.Vector3dVector(tgt)
    corr = o3d.utility.Vector2iVector(np.array([inlier_ind, inlier_ind]).T)
    result = reg.registration_ransac_based_on_correspondence(
        pcd0, pcd1, corr, 0.10, reg.TransformationEstimationPointToPoint(False), 3,
        [reg.CorrespondenceCheckerBasedOnEdgeLength(0.8), reg.CorrespondenceCheckerBasedOnDistance(0.10)],
        reg.RANSACConvergenceCriteria(50000, 0.999))
    t = lambda x: torch.from_numpy(x).to(dev)
    T, _ = ops.ransac_kabsch(t(src), t(tgt), t(np.array(corr)), nhyp=4096, seed=0, max_dist=0.10, edge_similarity=0.8)
    np.testing.assert_array_equal(result.transformation, T.cpu().numpy().astype(np.float64))
    assert np.abs(result.transformation[:3, :3] - R).max() < 0.02 and result.fitness > 0.6 and result.inlier_rmse < 0.1
    assert len(result.correspondence_set) == round(result.fitness * 400)
    empty = reg.registration_ransac_based_on_correspondence(pcd0, pcd1, o3d.utility.Vector2iVector(), 0.1)
    assert np.array_equal(empty.transformation, np.eye(4))


@pytest.mark.gpu
def test_icp_recovers_a_planted_offset_and_refines_kitti_ground_truth(dev):
    """open3d registration_icp(threshold 0.2, identity init, 200 iterations) as KITTI/dataset.py:104-107 calls it"""
    import open3d as o3d
    from buffer_amd import synth
    reg = o3d.pipelines.registration
    s = synth.make_pair(5, n_raw=60_000, size=(1.2, 1.0, 0.9), n_boxes=3)
    tgt = s['src_fds_pts'].astype(np.float64)
    a = np.deg2rad(1.5)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    t = np.array([0.03, -0.02, 0.015])
    rng = np.random.default_rng(0)
    src = (tgt[rng.permutation(len(tgt))[:15000]] - t) @ R                     # R src + t lands on the target surface
    pcd0, pcd1 = o3d.geometry.PointCloud(), o3d.geometry.PointCloud()
    pcd0.points, pcd1.points = o3d.utility.Vector3dVector(src), o3d.utility.Vector3dVector(tgt)
    res = reg.registration_icp(pcd0, pcd1, 0.20, np.eye(4), reg.TransformationEstimationPointToPoint(),
                               reg.ICPConvergenceCriteria(max_iteration=200))
    assert res.fitness > 0.99 and res.inlier_rmse < 1e-3
    assert np.abs(res.transformation[:3, :3] - R).max() < 1e-3 and np.abs(res.transformation[:3, 3] - t).max() < 1e-3
    none = reg.registration_icp(pcd0, o3d.geometry.PointCloud(), 0.2)
    assert none.fitness == 0.0 and np.array_equal(none.transformation, np.eye(4))
