"""N1 pre-processing (open3d voxel_down_sample / estimate_normals / orient_normals): the oracle restatement against
independent numpy formulations (CPU), the HIP kernels against the oracle (GPU).  Parity with open3d itself is
UNPINNED (open3d 0.13.0 is a pip dependency absent from /root/reference and from this image)."""
import numpy as np
import pytest


def _cloud(seed, n, scale=1.0):
    rng = np.random.default_rng(seed)
    # three faces of a box + jitter (the SURVEY 8d synthetic shape), away from the origin
    face = rng.integers(0, 3, n)
    p = rng.random((n, 3)) * 2.0
    p[np.arange(n), face] = 0.0
    p += rng.normal(0, 0.003, p.shape)
    return ((p + 0.5) * scale).astype(np.float32)


def _np_voxel_down_sample(pts, voxel):
    pts = np.asarray(pts, np.float64)
    o = pts.min(0) - voxel * 0.5
    idx = np.floor((pts - o) / voxel).astype(np.int64)
    N = np.floor((pts.max(0) - o) / voxel).astype(np.int64) + 1
    key = idx[:, 0] + N[0] * idx[:, 1] + N[0] * N[1] * idx[:, 2]
    rows = []
    for k in np.unique(key):
        sel = np.nonzero(key == k)[0]
        s = np.zeros(3)
        for i in sel:                      # input order, sequential fp64 sum
            s = s + pts[i]
        rows.append(s / len(sel))
    return np.array(rows)


def test_oracle_voxel_down_sample_vs_numpy(oracle):
    pts = _cloud(0, 4000)
    got = oracle.o3d_voxel_down_sample(pts.astype(np.float64), 0.05)
    want = _np_voxel_down_sample(pts, 0.05)
    assert got.shape == want.shape and 500 < got.shape[0] < 4000
    np.testing.assert_array_equal(got, want)
    # chained second level on the fp64 means (dataset.py:125) and the mean of normals
    got2, nrm2 = oracle.o3d_voxel_down_sample(got, 0.12, normals=np.ones_like(got) * [0.0, 0.0, 2.0])
    np.testing.assert_array_equal(got2, _np_voxel_down_sample(got, 0.12))
    np.testing.assert_allclose(nrm2, np.tile([0.0, 0.0, 2.0], (got2.shape[0], 1)))
    with pytest.raises(ValueError):
        oracle.o3d_voxel_down_sample(got, 0.0)
    assert oracle.o3d_voxel_down_sample(np.zeros((0, 3)), 0.1).shape == (0, 3)


def test_oracle_fast_eigen_vs_eigh(oracle):
    rng = np.random.default_rng(1)
    for _ in range(300):
        a = rng.normal(size=(3, 3)) * rng.choice([1e-3, 1.0, 30.0])
        c = a @ a.T
        if rng.random() < 0.2:
            c = np.diag(np.diag(c))        # the diagonal branch
        w, v = np.linalg.eigh(c)
        if (w[1] - w[0]) < 1e-6 * max(w[2], 1e-300):
            continue
        n = oracle.o3d_fast_eigen3x3([c[0, 0], c[0, 1], c[0, 2], c[1, 1], c[1, 2], c[2, 2]])
        assert abs(np.linalg.norm(n) - 1.0) < 1e-9
        assert abs(abs(n @ v[:, 0]) - 1.0) < 1e-8
    assert np.array_equal(oracle.o3d_fast_eigen3x3(np.zeros(6)), np.zeros(3))
    assert np.array_equal(oracle.o3d_fast_eigen3x3([1, 0, 0, 1, 0, 1]), [0, 0, 1])


def test_oracle_normals_vs_numpy(oracle):
    pts = _cloud(2, 600)
    got = oracle.o3d_estimate_normals(pts, knn=30)
    p64 = pts.astype(np.float64)
    for i in range(0, 600, 7):
        d2 = ((p64 - p64[i]) ** 2).sum(1)
        nb = np.lexsort((np.arange(600), d2))[:30]
        c = np.cov(p64[nb].T, bias=True)
        w, v = np.linalg.eigh(c)
        n = v[:, 0] if v[:, 0] @ (-p64[i]) >= 0 else -v[:, 0]
        assert got[i] @ n > 1 - 1e-5, i
    # fewer than 3 points: identity covariance -> (0,0,1), then oriented towards the camera
    two = np.array([[0, 0, 1], [0, 1, 2]], np.float32)
    np.testing.assert_array_equal(oracle.o3d_estimate_normals(two), [[0, 0, -1], [0, 0, -1]])
    np.testing.assert_array_equal(oracle.o3d_estimate_normals(two, orient=False), [[0, 0, 1], [0, 0, 1]])


@pytest.mark.gpu
@pytest.mark.parametrize("n,voxel,dtype", [(20000, 0.05, np.float32), (20000, 0.013, np.float64), (1, 0.1, np.float32), (7, 5.0, np.float32)])
def test_hip_voxel_down_sample_vs_oracle(oracle, dev, n, voxel, dtype):
    import torch
    from buffer_amd import preprocess
    pts = _cloud(3, n).astype(dtype)
    if dtype == np.float64:
        pts = pts + np.random.default_rng(4).normal(0, 1e-9, pts.shape)
    nrm = np.random.default_rng(5).normal(size=pts.shape).astype(dtype)
    got, got_n = preprocess.voxel_down_sample(torch.from_numpy(pts).to(dev), voxel, normals=torch.from_numpy(nrm).to(dev))
    want, want_n = oracle.o3d_voxel_down_sample(pts.astype(np.float64), voxel, normals=nrm.astype(np.float64))
    assert got.dtype == torch.float64
    np.testing.assert_array_equal(got.cpu().numpy(), want)             # fp64 sums in input order: bit-exact
    np.testing.assert_array_equal(got_n.cpu().numpy(), want_n)
    # chained level on the fp64 means
    got2 = preprocess.voxel_down_sample(got, voxel * 1.75)
    np.testing.assert_array_equal(got2.cpu().numpy(), oracle.o3d_voxel_down_sample(want, voxel * 1.75))


@pytest.mark.gpu
def test_hip_voxel_down_sample_errors(dev):
    import torch
    from buffer_amd import preprocess
    from buffer_amd._lib import BufferHipError
    with pytest.raises(BufferHipError):
        preprocess.voxel_down_sample(torch.zeros((4, 3), device=dev), 0.0)
    with pytest.raises(BufferHipError):
        preprocess.voxel_down_sample(torch.zeros((4, 3)), 0.1)          # host tensor: no CPU path
    assert preprocess.voxel_down_sample(torch.zeros((0, 3), device=dev), 0.1).shape == (0, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("n,knn", [(6000, 30), (500, 30), (40, 30), (20, 30), (2, 30), (3000, 10)])
def test_hip_normals_vs_oracle(oracle, dev, n, knn):
    import torch
    from buffer_amd import preprocess
    pts = _cloud(6, n)
    if n == 6000:
        pts[:50] += 40.0                     # a far, sparse cluster: rows that need several radius doublings
    got = preprocess.estimate_normals(torch.from_numpy(pts).to(dev), knn=knn, camera=(0.1, -0.2, 0.3)).cpu().numpy()
    want = oracle.o3d_estimate_normals(pts, knn=knn, camera=(0.1, -0.2, 0.3))
    assert np.all(np.abs(np.linalg.norm(got, axis=1) - 1.0) < 1e-5)
    dots = (got * want).sum(1)
    # the same neighbour sets and the same solver: agreement to round-off wherever the two smallest eigenvalues
    # are separated; allow a handful of ill-conditioned rows (edges of the box) a looser bound
    assert np.mean(dots > 1 - 1e-6) > 0.995, float(np.mean(dots > 1 - 1e-6))
    assert np.all(dots > 1 - 1e-3), float(dots.min())


@pytest.mark.gpu
def test_prepare_fragment_feeds_the_pipeline(dev):
    """raw cloud -> two voxel levels + normals, shaped like ThreeDMatchDataset's test items."""
    import torch
    from buffer_amd import preprocess
    raw = torch.from_numpy(_cloud(8, 60000, scale=1.5)).to(dev)
    item = preprocess.prepare_fragment(raw, downsample=0.02, voxel_size_0=0.035, seed=1)
    fds, sds = item['fds_pts'], item['sds_pts']
    assert fds.dtype == torch.float32 and sds.shape[1] == 6 and 0 < sds.shape[0] < fds.shape[0] <= 60000
    n = sds[:, 3:]
    assert torch.all((n.norm(dim=1) - 1).abs() < 1e-5)
    assert torch.all((n * (-sds[:, :3])).sum(1) >= 0)                   # oriented towards the camera at the origin
    again = preprocess.prepare_fragment(raw, downsample=0.02, voxel_size_0=0.035, seed=1)
    assert torch.equal(again['sds_pts'], sds) and torch.equal(again['fds_pts'], fds)


@pytest.mark.gpu
def test_raw_clouds_to_pose(dev):
    """raw (dense, unvoxelised) fragment pair -> device pre-processing -> BufferPipeline -> the planted pose."""
    import torch
    from buffer_amd import preprocess, synth
    from buffer_amd.pipeline import BufferPipeline
    s = synth.make_pair(seed=5)
    rng = np.random.default_rng(5)
    items = {}
    for side in ('src', 'tgt'):
        dense = np.concatenate([s[f'{side}_fds_pts'] + rng.normal(0, 0.004, s[f'{side}_fds_pts'].shape) for _ in range(4)])
        it = preprocess.prepare_fragment(torch.from_numpy(dense.astype(np.float32)).to(dev), 0.02, 0.035, seed=11)
        items[f'{side}_fds_pts'] = it['fds_pts'].cpu().numpy()
        items[f'{side}_sds_pts'] = it['sds_pts'].cpu().numpy()
    items['relt_pose'] = s['relt_pose']
    pipe = BufferPipeline(device=dev)
    pipe.calibrate([items])
    T = pipe.register(pipe.upload(items)).cpu().numpy().astype(np.float64)
    gt = s['relt_pose']
    rte = np.linalg.norm(T[:3, 3] - gt[:3, 3])
    rre = np.degrees(np.arccos(np.clip((np.trace(T[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1)))
    assert rte < 0.1 and rre < 3.0, (rte, rre)


@pytest.mark.gpu
def test_prepare_fragment_kitti_scale(oracle, dev):
    """KITTI constants (KITTI/dataset.py:123-178: 0.05 m and 0.30 m voxels) on a LiDAR-shaped scan of ~160 m extent:
    sparse far rings need several radius growths in the k-NN search; voxel means == oracle bit for bit."""
    import torch
    from buffer_amd import preprocess, synth
    s = synth.make_kitti_pair(seed=2)
    raw = np.ascontiguousarray(s['src_fds_pts'][:, :3], dtype=np.float32)
    assert raw.shape[0] > 30000 and np.ptp(raw[:, 0]) > 60
    it = preprocess.prepare_fragment(torch.from_numpy(raw).to(dev), 0.05, 0.30, seed=4)
    fds, sds = it['fds_pts'], it['sds_pts']
    want_fds = oracle.o3d_voxel_down_sample(raw.astype(np.float64), 0.05)
    want_sds = oracle.o3d_voxel_down_sample(want_fds, 0.30)
    assert fds.shape[0] == want_fds.shape[0] and sds.shape[0] == want_sds.shape[0]
    # rows are shuffled by prepare_fragment: compare as sorted sets of f32 rows
    key = lambda a: a[np.lexsort(a.T[::-1])]
    np.testing.assert_array_equal(key(fds.cpu().numpy()), key(want_fds.astype(np.float32)))
    np.testing.assert_array_equal(key(sds[:, :3].cpu().numpy()), key(want_sds.astype(np.float32)))
    n = sds[:, 3:]
    assert torch.all((n.norm(dim=1) - 1).abs() < 1e-5) and torch.all((n * (-sds[:, :3])).sum(1) >= 0)
    want_n = oracle.o3d_estimate_normals(sds[:, :3].cpu().numpy(), knn=30)
    dots = (n.cpu().numpy() * want_n).sum(1)
    assert np.mean(dots > 1 - 1e-6) > 0.99 and np.all(dots > 1 - 1e-3), (float(np.mean(dots > 1 - 1e-6)), float(dots.min()))


@pytest.mark.gpu
def test_prepare_fragments_stacked_normals_equal_one_by_one(dev):
    """prepare_fragments (one stacked normal-estimation pass over all fragments) == prepare_fragment fragment by fragment,
    bit for bit, including a tiny cloud (fewer points than the 30-neighbourhood)"""
    import torch
    from buffer_amd import preprocess, stream
    raws = stream.generate(3, dev, n_raw=120_000)
    clouds = [r[k] for r in raws for k in ('src_raw', 'tgt_raw')] + [raws[0]['src_raw'][:200].contiguous()]
    seeds = [7, 8, 9, 10, 11, 12, 13]
    got = preprocess.prepare_fragments(clouds, 0.02, 0.035, 30000, seeds)
    for c, s, g in zip(clouds, seeds, got):
        want = preprocess.prepare_fragment(c, 0.02, 0.035, 30000, s)
        assert torch.equal(g['fds_pts'], want['fds_pts']) and torch.equal(g['sds_pts'], want['sds_pts'])
    nrm = got[0]['sds_pts'][:, 3:]
    assert torch.allclose(nrm.norm(dim=1), torch.ones_like(nrm[:, 0]), atol=1e-5)


@pytest.mark.gpu
def test_voxel_down_sample_batch_equals_cloud_by_cloud(dev):
    """buf_voxel_downsample_batch: several stacked clouds (own bounding box each, one of them empty, fp32 and fp64 input) give,
    cloud by cloud, exactly the rows of separate voxel_down_sample calls."""
    import torch
    from buffer_amd import preprocess
    g = torch.Generator(device='cpu').manual_seed(4)
    clouds = [torch.rand((n, 3), generator=g) * s + o for n, s, o in ((5000, 2.0, 0.0), (1, 1.0, 5.0), (0, 1.0, 0.0), (12345, 3.0, -7.0), (777, 0.5, 100.0))]
    for dt in (torch.float32, torch.float64):
        cl = [c.to(dt).to(dev) for c in clouds]
        lens = [int(c.shape[0]) for c in cl]
        out, out_lens = preprocess.voxel_down_sample_batch(torch.cat(cl), lens, 0.11)
        lo = 0
        for c, m in zip(cl, out_lens):
            want = preprocess.voxel_down_sample(c, 0.11) if c.shape[0] else torch.zeros((0, 3), dtype=torch.float64, device=dev)
            assert int(m) == want.shape[0]
            assert torch.equal(out[lo:lo + int(m)], want)
            lo += int(m)
        assert lo == out.shape[0]
