"""The reference-named shim packages, called the way the reference calls them, on the GPU."""
import numpy as np
import pytest
import torch

from buffer_amd import synth
from util import assert_neighbors_equal_mod_ties as nbr_eq

pytestmark = pytest.mark.gpu


def test_collate_calls_through_the_shims(oracle, dev):
    """ThreeDMatch/dataloader.py:60-110 call pattern: torch CPU tensors (float64 at layer 0) in, numpy out."""
    import buffer_amd.shims as shims
    shims.install()
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
    import cpp_wrappers.cpp_neighbors.radius_neighbors as cpp_neighbors
    d = synth.make_config1_pair()
    pts64 = torch.from_numpy(np.concatenate([d['src_sds_pts'][:, :3], d['tgt_sds_pts'][:, :3]]))      # float64
    lens = torch.from_numpy(np.array([5000, 5000])).int()
    nb = cpp_neighbors.batch_query(pts64, pts64, lens, lens, radius=0.07)
    assert isinstance(nb, np.ndarray) and nb.dtype == np.int32
    p32 = pts64.numpy().astype(np.float32)
    want = oracle.radius_neighbors(p32, p32, lens.numpy(), lens.numpy(), 0.07)
    assert np.array_equal(nb, want)
    if oracle.have_ref():
        nbr_eq(nb, oracle.ref_radius_neighbors(p32, p32, lens.numpy(), lens.numpy(), 0.07), p32, p32)
    s_points, s_len = cpp_subsampling.subsample_batch(pts64, lens, sampleDl=0.07, max_p=0, verbose=0)
    wp, wl = oracle.grid_subsample_batch(p32, lens.numpy(), 0.07)
    assert s_points.dtype == np.float32 and s_len.dtype == np.int32
    assert np.array_equal(s_len, wl) and np.array_equal(s_points.view(np.uint32), wp.view(np.uint32))
    one = cpp_subsampling.subsample(p32[:5000], sampleDl=0.07)
    assert np.array_equal(one.view(np.uint32), wp[:wl[0]].view(np.uint32))
    # feature means: summed in input order, divided by the count
    sp, sl, sf = cpp_subsampling.subsample_batch(p32, lens, features=p32, sampleDl=0.07)
    assert np.array_equal(sp.view(np.uint32), wp.view(np.uint32))
    np.testing.assert_allclose(sf, sp, rtol=2e-6, atol=1e-7)
    with pytest.raises(RuntimeError):
        cpp_neighbors.batch_query(p32 + 100, p32, lens, lens, radius=0.01)       # no neighbour at all -> "Error"


def test_pointnet2_knn_svd_shims(oracle, dev):
    import buffer_amd.shims as shims
    shims.install()
    import pointnet2_ops.pointnet2_utils as pnt2
    from knn_cuda import KNN
    from torch_batch_svd import svd
    rng = np.random.default_rng(0)
    xyz = (rng.random((1, 3000, 3)).astype(np.float32) + 0.5)
    t = torch.from_numpy(xyz).to(dev)
    idx = pnt2.furthest_point_sample(t, 200)
    assert np.array_equal(idx.cpu().numpy(), oracle.fps(xyz, 200))
    flipped = t.transpose(1, 2).contiguous()
    k = pnt2.gather_operation(flipped, idx).transpose(1, 2).contiguous()           # models/BUFFER.py:268
    assert torch.equal(k[0], t[0][idx[0].long()])
    g = pnt2.ball_query(0.1, 16, t, k)
    assert np.array_equal(g.cpu().numpy(), oracle.ball_query(0.1, 16, xyz, k.cpu().numpy()))
    des = torch.from_numpy(rng.normal(size=(300, 32)).astype(np.float32)).to(dev)
    des2 = torch.from_numpy(rng.normal(size=(280, 32)).astype(np.float32)).to(dev)
    dis, ind = KNN(k=1, transpose_mode=True)(des2.unsqueeze(0), des.unsqueeze(0))   # models/BUFFER.py:347
    wd, wi = oracle.knn(des2[None].cpu().numpy(), des[None].cpu().numpy(), 1)
    assert ind.dtype == torch.int64 and np.array_equal(ind.cpu().numpy(), wi)
    np.testing.assert_allclose(dis.cpu().numpy(), wd, rtol=1e-6)
    a = torch.from_numpy(rng.normal(size=(50, 3, 3)).astype(np.float32)).to(dev)
    u, s, v = svd(a)
    np.testing.assert_allclose(((u * s[:, None]) @ v.transpose(1, 2)).cpu().numpy(), a.cpu().numpy(), atol=2e-5)


def test_subsample_label_voting(dev):
    """classes= of subsample_batch / subsample (grid_subsampling.cpp:97-103): majority label per voxel and label column,
    returned after the features like the CPython module does"""
    import buffer_amd.shims as shims
    shims.install()
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
    d = synth.make_config1_pair()
    pts = np.concatenate([d['src_sds_pts'][:, :3], d['tgt_sds_pts'][:, :3]]).astype(np.float32)
    lens = np.array([5000, 5000], np.int32)
    rng = np.random.default_rng(3)
    labels = np.stack([(pts[:, 0] > np.median(pts[:, 0])).astype(np.int32) * 7 + (rng.random(len(pts)) < 0.2),      # noisy halves
                       rng.integers(-3, 4, len(pts))], 1).astype(np.int32)
    dl = 0.12
    sp, sl, sf, sc = cpp_subsampling.subsample_batch(pts, lens, features=pts, classes=labels, sampleDl=dl)
    assert sc.dtype == np.int32 and sc.shape == (sp.shape[0], 2) and sf.shape == sp.shape
    # independent grouping: the reference's voxel key per batch element (grid_subsampling.cpp:40-60), majority by counting
    lo = 0
    row = 0
    for n in lens:
        p, lab = pts[lo:lo + n], labels[lo:lo + n]
        origin = np.floor(p.min(0) * np.float32(1.0 / dl)) * np.float32(dl)
        key = np.floor((p - origin) / np.float32(dl)).astype(np.int64)
        groups = {}
        for i, k in enumerate(map(tuple, key)):
            groups.setdefault(k, []).append(i)
        bary = {k: p[v].astype(np.float64).mean(0) for k, v in groups.items()}
        m = int(sl[0] if lo == 0 else sl[1])
        assert m == len(groups)
        mine = sp[row:row + m]
        for k, v in groups.items():
            j = int(np.argmin(((mine - bary[k]) ** 2).sum(1)))
            assert np.abs(mine[j] - bary[k]).max() < 1e-5
            for dcol in range(2):
                vals, cnt = np.unique(lab[v, dcol], return_counts=True)
                assert sc[row + j, dcol] == vals[np.argmax(cnt)]          # ties -> smallest label (np.unique is ascending)
        lo += n
        row += m
    one_p, one_c = cpp_subsampling.subsample(pts[:5000], classes=labels[:5000, 0], sampleDl=dl)
    assert np.array_equal(one_c[:, 0], sc[:sl[0], 0]) and np.array_equal(one_p, sp[:sl[0]])
    with pytest.raises(RuntimeError):
        cpp_subsampling.subsample_batch(pts, lens, classes=labels[:-1], sampleDl=dl)
