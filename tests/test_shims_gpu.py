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
