"""Edge cases through the C ABI: sparse extents, the reference's wrapped voxel key, truncation, degenerate
inputs, argument errors (error codes, no crashes)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _eq_sub(oracle, dev, pts, lens, dl, **kw):
    """both forms of the operator against the oracle: one workgroup per element with its bucket table in LDS (the default when every element
    has <= 16384 points) and the global-table counting sort (BUF_VOX_FUSED=0)"""
    import os
    from buffer_amd import ops
    want, wl = oracle.grid_subsample_batch(pts, lens, dl, **{k: v for k, v in kw.items() if k == 'max_p'})
    old = os.environ.get('BUF_VOX_FUSED')
    try:
        for form in ('1', '0'):
            os.environ['BUF_VOX_FUSED'] = form
            got, gl = ops.grid_subsample_batch(torch.from_numpy(pts).to(dev), lens, dl, **kw)
            assert np.array_equal(gl, wl), form
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), form
    finally:
        if old is None:
            os.environ.pop('BUF_VOX_FUSED', None)
        else:
            os.environ['BUF_VOX_FUSED'] = old


def test_subsample_sparse_extent_uses_key_buckets(oracle, dev):
    """an outlier 10 km away: 3e15 voxels in the bounding box, the bucketed table still reproduces the oracle"""
    rng = np.random.default_rng(0)
    pts = rng.random((5000, 3)).astype(np.float32)
    pts[17] = [10000.0, -8000.0, 9000.0]
    pts[4000] = [-5000.0, 3.0, 2.0]
    lens = np.array([3000, 2000], np.int32)
    _eq_sub(oracle, dev, pts, lens, 0.05)
    _eq_sub(oracle, dev, pts, lens, 0.05, max_cells=4096)          # tiny table: many voxels per bucket


def test_subsample_wrapped_key_corner_case(oracle, dev):
    """floor(min*(1/dl))*dl can round ABOVE min (e.g. dl = 0.6 -- KITTI's first pooling -- with min = 6.6): the reference then casts floor(negative) to size_t and the voxel key wraps.  We land where the restated
    reference lands (the wrapped keys are the largest ones)."""
    rng = np.random.default_rng(1)
    for dl, m in ((0.6, np.float32(6.6)), (0.6, np.float32(20.4)), (0.3, np.float32(25.5)), (0.3, np.float32(20.4))):
        d = np.float32(dl)
        o = np.floor(m * (np.float32(1) / d)) * d
        assert np.floor((m - o) / d) < 0, "not a counter-example any more?"
        pts = (rng.random((3000, 3)) * 4 + float(m) + 0.5).astype(np.float32)
        pts[5, 0] = m
        pts[9, 1] = m
        pts[2000, 2] = m
        _eq_sub(oracle, dev, pts, np.array([1800, 1200], np.int32), dl)


def test_subsample_max_p_and_single_voxel(oracle, dev):
    rng = np.random.default_rng(2)
    pts = rng.random((3000, 3)).astype(np.float32)
    lens = np.array([1000, 2000], np.int32)
    _eq_sub(oracle, dev, pts, lens, 0.2, max_p=50)
    _eq_sub(oracle, dev, pts, lens, 100.0)                          # everything in one voxel: a 2000-term ordered sum
    same = np.tile(np.array([[0.3, 0.4, 0.5]], np.float32), (500, 1))
    _eq_sub(oracle, dev, same, np.array([500], np.int32), 0.1)


def test_subsample_forms_agree_on_ragged_batches_features_and_large_elements(oracle, dev):
    """the LDS form and the global-table form: ragged batch with empty elements, features, max_p, an element over the LDS form's 16384
    points (the call falls back as a whole), a fragment-sized batch"""
    import os
    from buffer_amd import ops, synth
    rng = np.random.default_rng(7)
    pts = (rng.random((9000, 3)) * np.array([3.0, 2.0, 1.0])).astype(np.float32)
    lens = np.array([0, 4000, 0, 1, 4999, 0], np.int32)
    _eq_sub(oracle, dev, pts, lens, 0.07)
    _eq_sub(oracle, dev, pts, lens, 0.15, max_p=37)
    feats = rng.normal(size=(9000, 5)).astype(np.float32)
    res = {}
    for form in ('1', '0'):
        os.environ['BUF_VOX_FUSED'] = form
        try:
            for mp in (0, 23):
                p_, l_, f_ = ops.grid_subsample_batch(torch.from_numpy(pts).to(dev), lens, 0.1, max_p=mp, features=torch.from_numpy(feats).to(dev))
                res[form, mp] = (p_.cpu().numpy(), l_.copy(), f_.cpu().numpy())
        finally:
            os.environ.pop('BUF_VOX_FUSED', None)
    for mp in (0, 23):
        for a, b_ in zip(res['1', mp], res['0', mp]):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b_.view(np.uint32) if b_.dtype == np.float32 else b_)
    big = (rng.random((40000 + 3000, 3)) * 2).astype(np.float32)          # 40000 > 16384: global-table form for the whole call
    _eq_sub(oracle, dev, big, np.array([40000, 3000], np.int32), 0.05)
    d = synth.make_pair(11)
    for key, dls in (('sds', (0.07, 0.14, 0.6)), ('fds', (0.07,))):       # sds: 9-15 k points per element (LDS form); fds: > 16384 (global-table form)
        frag = np.concatenate([d[f'src_{key}_pts'][:, :3], d[f'tgt_{key}_pts'][:, :3]]).astype(np.float32)
        for dl in dls:
            _eq_sub(oracle, dev, frag, np.array([len(d[f'src_{key}_pts']), len(d[f'tgt_{key}_pts'])], np.int32), dl)
    dense = (rng.random((16384, 3)) * 0.2).astype(np.float32)             # the LDS form's largest element, 2500 points per voxel
    _eq_sub(oracle, dev, dense, np.array([16384], np.int32), 0.1)


def test_degenerate_point_ops(oracle, dev):
    from buffer_amd import ops
    rng = np.random.default_rng(3)
    xyz = (rng.random((1, 40, 3)) + 1).astype(np.float32)
    t = torch.from_numpy(xyz).to(dev)
    # more samples than points: FPS keeps going (upstream semantics), ball query pads, kNN with k > n pads
    assert np.array_equal(ops.furthest_point_sample(t, 100).cpu().numpy(), oracle.fps(xyz, 100))
    assert np.array_equal(ops.ball_query(0.2, 64, t, t).cpu().numpy(), oracle.ball_query(0.2, 64, xyz, xyz))
    feat = rng.normal(size=(1, 5, 8)).astype(np.float32)
    q = rng.normal(size=(1, 3, 8)).astype(np.float32)
    gd, gi = ops.knn(torch.from_numpy(feat).to(dev), torch.from_numpy(q).to(dev), 7)
    wd, wi = oracle.knn(feat, q, 7)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(np.isinf(gd.cpu().numpy()), np.isinf(wd))
    # a cloud the upstream FPS kernel skips entirely (all points within sqrt(1e-3) of the origin)
    tiny = (rng.random((1, 300, 3)) * 0.01).astype(np.float32)
    assert np.array_equal(ops.furthest_point_sample(torch.from_numpy(tiny).to(dev), 20).cpu().numpy(), oracle.fps(tiny, 20))
    # radius search with queries far outside the support bounding box and a zero radius
    s = (rng.random((500, 3))).astype(np.float32)
    qf = s + 50.0
    out = ops.radius_neighbors(torch.from_numpy(qf).to(dev), torch.from_numpy(s).to(dev), [500], [500], 0.1, k=4)
    assert bool((out == 500).all())
    out = ops.radius_neighbors(torch.from_numpy(s).to(dev), torch.from_numpy(s).to(dev), [500], [500], 0.0, k=3)
    assert bool((out == 500).all())                                 # strict '<': d2 = 0 is not < 0


def test_c_abi_error_codes(dev):
    """bad arguments come back as BUF_E* codes with a message; nothing faults"""
    from buffer_amd import _lib
    L = _lib.lib()
    t = torch.rand(100, 3, device=dev)
    lens = (C.c_int * 1)(100)
    out = torch.empty(100, 4, dtype=torch.int32, device=dev)
    ws = torch.empty(1024, dtype=torch.uint8, device=dev)           # far too small
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.buf_radius_neighbors(t.data_ptr(), 100, t.data_ptr(), 100, lens, lens, 1, 0.1, 4, out.data_ptr(), None, None,
                                ws.data_ptr(), 1024, s)
    assert rc == -3 and b"workspace" in L.buf_last_error()
    bad = (C.c_int * 1)(99)                                         # lengths do not sum to n
    big = torch.empty(L.buf_grid_ws_bytes(100, 1, 0) + 400, dtype=torch.uint8, device=dev)
    rc = L.buf_radius_neighbors(t.data_ptr(), 100, t.data_ptr(), 100, bad, lens, 1, 0.1, 4, out.data_ptr(), None, None,
                                big.data_ptr(), big.numel(), s)
    assert rc == -1 and b"sum" in L.buf_last_error()
    rc = L.buf_knn(t.data_ptr(), t.data_ptr(), 1, 100, 100, 3, 100, None, None, None, 0, s)
    assert rc == -1                                                 # k > 64
    rc = L.buf_fps(None, 1, 100, 10, None, None, 0, s)
    assert rc == -1
    rc = L.buf_vn_gather_block(None, None, None, None, 10, 10, 4, 2, 4, 3, 1.0, None, None, None, None, 0.2, None, s)
    assert rc == -1 and b"mode" in L.buf_last_error()
    # round-2 entry points: batched patch selection (workspace, lengths), keyed cloud permutation (null cloud)
    pat = torch.empty(8, 16, 3, device=dev)
    rc = L.buf_select_patches_batched(t.data_ptr(), lens, 1, t.data_ptr(), 8, 0.1, 16, pat.data_ptr(), ws.data_ptr(), 1024, s)
    assert rc == -3 and b"workspace" in L.buf_last_error()
    neg = (C.c_int * 1)(-5)
    rc = L.buf_select_patches_batched(t.data_ptr(), neg, 1, t.data_ptr(), 8, 0.1, 16, pat.data_ptr(), big.data_ptr(), big.numel(), s)
    assert rc == -1
    rc = L.buf_select_patches_batched(t.data_ptr(), lens, 1, t.data_ptr(), 8, -1.0, 16, pat.data_ptr(), big.data_ptr(), big.numel(), s)
    assert rc == -1 and b"radius" in L.buf_last_error()
    ptrs, keys = (C.c_void_p * 1)(None), (C.c_ulonglong * 1)(1)
    rc = L.buf_permute_clouds(ptrs, lens, keys, 1, t.data_ptr(), s)
    assert rc == -1 and b"cloud 0" in L.buf_last_error()
    assert L.buf_permute_clouds(ptrs, lens, keys, 0, None, s) == 0          # nothing to do
    # Winograd-form descriptor CNN: widths it does not support, missing weights; batched voxel down-sampling: lengths, voxel size
    x = torch.zeros(2, 48, 140, device=dev)
    y = torch.zeros(2, 32, 140, device=dev)
    wts = [torch.zeros(16 * 128 * 128, device=dev) for _ in range(8)]
    bs = [torch.zeros(128, device=dev) for _ in range(8)]
    wp, bp = (C.c_void_p * 8)(*[w.data_ptr() for w in wts]), (C.c_void_p * 8)(*[b.data_ptr() for b in bs])
    relu = (C.c_int * 8)(1, 1, 1, 1, 1, 1, 1, 0)
    good_in, good_out = (48, 64, 64, 128, 128, 64, 64, 32), (64, 64, 128, 128, 64, 64, 32, 32)
    assert L.buf_cylindrical_net_wg(x.data_ptr(), 0, wp, bp, (C.c_int * 8)(*good_in), (C.c_int * 8)(*good_out), relu, y.data_ptr(), s) == 0
    assert L.buf_cylindrical_net_wg(x.data_ptr(), 2, wp, bp, (C.c_int * 8)(*good_in), (C.c_int * 8)(*good_out), relu, y.data_ptr(), s) == 0
    bad_in = (C.c_int * 8)(40, 64, 64, 128, 128, 64, 64, 32)                    # 40 input channels: not a multiple of 16
    rc = L.buf_cylindrical_net_wg(x.data_ptr(), 2, wp, bp, bad_in, (C.c_int * 8)(*good_out), relu, y.data_ptr(), s)
    assert rc == -1 and b"unsupported widths" in L.buf_last_error()
    bad_out = (C.c_int * 8)(64, 64, 128, 128, 64, 64, 32, 64)                   # the last layer must have 32 channels
    rc = L.buf_cylindrical_net_wg(x.data_ptr(), 2, wp, bp, (C.c_int * 8)(*good_in), bad_out, relu, y.data_ptr(), s)
    assert rc == -1
    # a wide layer behind a 32-output layer: the K-split hand-over of the 32-output layers overwrites the padding zeros of channels 64..95
    wide_in, wide_out = (C.c_int * 8)(48, 64, 32, 64, 128, 128, 64, 32), (C.c_int * 8)(64, 32, 64, 128, 128, 64, 32, 32)
    rc = L.buf_cylindrical_net_wg(x.data_ptr(), 2, wp, bp, wide_in, wide_out, relu, y.data_ptr(), s)
    assert rc == -1 and b"may follow a 32-output layer" in L.buf_last_error()
    wp_null = (C.c_void_p * 8)(*([w.data_ptr() for w in wts[:7]] + [None]))
    rc = L.buf_cylindrical_net_wg(x.data_ptr(), 2, wp_null, bp, (C.c_int * 8)(*good_in), (C.c_int * 8)(*good_out), relu, y.data_ptr(), s)
    assert rc == -1 and b"null weights" in L.buf_last_error()
    from buffer_amd import preprocess
    with pytest.raises(_lib.BufferHipError):
        preprocess.voxel_down_sample_batch(t, [60, 30], 0.1)                  # lengths do not sum to the number of points
    with pytest.raises(_lib.BufferHipError):
        preprocess.voxel_down_sample_batch(t, [100], 0.0)
    out64 = torch.empty(100, 3, dtype=torch.float64, device=dev)
    olen = (C.c_int * 1)(0)
    rc = L.buf_voxel_downsample_batch(t.data_ptr(), 0, 100, lens, 1, 0.1, out64.data_ptr(), olen, 1 << 16, ws.data_ptr(), 1024, s)
    assert rc == -3 and b"workspace" in L.buf_last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("npts,rad_n,azi_n,ele_n,nsample", [(200, 3, 20, 7, 10), (512, 2, 8, 4, 4), (777, 3, 20, 7, 16), (33, 1, 4, 2, 1)])
def test_voxelize_other_shapes_vs_oracle(dev, npts, rad_n, azi_n, ele_n, nsample):
    """Fused voxelisation with other patch sizes / voxel grids / sample counts than the 3DMatch constants (hit masks not a
    multiple of 32 points, centres not 420, nsample up to 16): == oracle SPT + point MLP; KITTI-style R = I as well."""
    import torch
    from buffer_amd import ops
    from buffer_amd.weights import load_weights
    from oracle import torch_ref as T
    W = load_weights("3dmatch")
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    g = torch.Generator(device='cpu').manual_seed(npts)
    P = 29
    patches = (torch.rand((P, npts, 3), generator=g) * 2 - 1) * 0.25 + torch.tensor([1.0, -2.0, 0.5])
    patches[:, -1] = torch.tensor([1.0, -2.0, 0.5])                       # keypoint in the last slot
    patches[3, :5] = patches[3, -1]                                        # points on the centre: hit many balls, incl. point 0
    axis = torch.nn.functional.normalize(torch.randn((P, 3), generator=g), dim=1)
    centres = T.voxel_centres(rad_n, azi_n, ele_n)
    ang = -torch.arange(azi_n, dtype=torch.float64) * 2 * np.pi / azi_n
    azi_cs = torch.stack([torch.cos(ang), torch.sin(ang)], 1).float()
    s = Wt['Desc.pnt_layer.1.weight'] / torch.sqrt(Wt['Desc.pnt_layer.1.running_var'] + 1e-5)
    t = Wt['Desc.pnt_layer.1.bias'] - Wt['Desc.pnt_layer.1.running_mean'] * s
    for ax, dataset in ((axis, '3DMatch'), (None, 'KITTI')):
        x, R, ra, pn = ops.patch_voxelize(patches.to(dev), None if ax is None else ax.to(dev), 0.3, centres.to(dev), azi_cs.to(dev),
                                          0.8 / rad_n, nsample, W['Desc.pnt_layer.0.weight'].reshape(16, 3), W['Desc.pnt_layer.0.bias'],
                                          s.numpy(), t.numpy(), azi_n, True)
        with torch.no_grad():
            al, rand_axis, Rw = T.axis_align(patches, axis, dataset)
            inv = T.spt(al / 0.3, rad_n, azi_n, ele_n, 0.8, nsample)
            want = T.point_mlp_max(inv, Wt).reshape(P, 16, -1)
        np.testing.assert_allclose(pn.cpu().numpy(), (al / 0.3).numpy(), rtol=0, atol=3e-6)
        np.testing.assert_allclose(R.cpu().numpy(), Rw.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(x.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


def test_voxelize_points_on_the_ball_surfaces(dev):
    """k_patch_voxelize finds the hit masks with a split-f16 distance form on the matrix pipe and hands every pair it cannot call
    (|d^2 - r^2| < eps) to the reference's own fp32 test.  Here EVERY point sits on the surface of some voxel ball, centre + r u
    (1 + delta) with delta from 0 to +-1e-5, plus points beyond every ball's reach and non-finite ones: the sample lists (first
    nsample hits in index order) must be the oracle's -- one flipped pair changes a sample, i.e. the output, far beyond 1e-4."""
    import torch
    from buffer_amd import ops
    from buffer_amd.weights import load_weights
    from oracle import torch_ref as T
    W = load_weights("3dmatch")
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    rad_n, azi_n, ele_n, nsample, npts, P = 3, 20, 7, 10, 512, 48
    g = torch.Generator(device='cpu').manual_seed(77)
    centres = T.voxel_centres(rad_n, azi_n, ele_n)
    r = 0.8 / rad_n
    ci = torch.randint(0, centres.shape[0], (P, npts), generator=g)
    u = torch.nn.functional.normalize(torch.randn((P, npts, 3), generator=g), dim=-1)
    deltas = torch.tensor([0, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 1e-5, -1e-5], dtype=torch.float64)
    dl = deltas[torch.randint(0, len(deltas), (P, npts), generator=g)]
    patches = (centres[ci].double() + r * u.double() * (1 + dl[..., None])).float()
    patches[:, 7] = torch.tensor([3.0, -2.0, 1.5])                         # beyond the reach of every ball
    patches[:, 9] = torch.tensor([1e30, 0.0, 0.0])
    patches[:, -1] = 0                                                     # keypoint (origin) in the last slot
    ang = -torch.arange(azi_n, dtype=torch.float64) * 2 * np.pi / azi_n
    azi_cs = torch.stack([torch.cos(ang), torch.sin(ang)], 1).float()
    s = Wt['Desc.pnt_layer.1.weight'] / torch.sqrt(Wt['Desc.pnt_layer.1.running_var'] + 1e-5)
    t = Wt['Desc.pnt_layer.1.bias'] - Wt['Desc.pnt_layer.1.running_mean'] * s
    x, R, ra, pn = ops.patch_voxelize(patches.to(dev), None, 1.0, centres.to(dev), azi_cs.to(dev), r, nsample,
                                      W['Desc.pnt_layer.0.weight'].reshape(16, 3), W['Desc.pnt_layer.0.bias'], s.numpy(), t.numpy(),
                                      azi_n, True)
    assert torch.equal(pn.cpu(), patches)                                  # R = I, des_r = 1, keypoint at the origin
    with torch.no_grad():
        want = T.point_mlp_max(T.spt(patches, rad_n, azi_n, ele_n, 0.8, nsample), Wt).reshape(P, 16, -1)
    np.testing.assert_allclose(x.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


def test_voxelize_scaled_geometry(dev):
    """The same patches at 128 x the scale (centres, voxel radius and patch radius scaled by the power of two, MLP weights by its
    inverse: every fp32 distance and product scales exactly) give the same output.  At that scale the voxel grid is past what the
    split-f16 distance form of k_patch_voxelize may represent (max |c| > 64), so every pair takes the fp32 test."""
    import torch
    from buffer_amd import ops
    from buffer_amd.weights import load_weights
    from oracle import torch_ref as T
    W = load_weights("3dmatch")
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    g = torch.Generator(device='cpu').manual_seed(5)
    P, npts = 40, 512
    u = torch.nn.functional.normalize(torch.randn((P, npts, 3), generator=g), dim=-1) * torch.rand((P, npts, 1), generator=g).sqrt() * 0.3
    u[:, -1] = 0
    patches = u + torch.tensor([0.5, -1.0, 2.0])
    axis = torch.nn.functional.normalize(torch.randn((P, 3), generator=g), dim=1)
    centres = T.voxel_centres(3, 20, 7)
    ang = -torch.arange(20, dtype=torch.float64) * 2 * np.pi / 20
    azi_cs = torch.stack([torch.cos(ang), torch.sin(ang)], 1).float()
    s = Wt['Desc.pnt_layer.1.weight'] / torch.sqrt(Wt['Desc.pnt_layer.1.running_var'] + 1e-5)
    t = Wt['Desc.pnt_layer.1.bias'] - Wt['Desc.pnt_layer.1.running_mean'] * s
    w0 = W['Desc.pnt_layer.0.weight'].reshape(16, 3)
    out = []
    for k in (1.0, 128.0):
        x, R, ra, pn = ops.patch_voxelize(patches.to(dev), axis.to(dev), 0.3 / k, (centres * k).to(dev), azi_cs.to(dev), k * 0.8 / 3, 10,
                                          w0 / k, W['Desc.pnt_layer.0.bias'], s.numpy(), t.numpy(), 20, True)
        out.append((x.cpu().numpy(), pn.cpu().numpy() / k))
    assert np.array_equal(out[0][1], out[1][1])
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=1e-5, atol=1e-5)     # (the f16 pieces of w / 128 round differently; a flipped hit shows at 1e-3 and up)


@pytest.mark.parametrize("n", [0, 1, 3, 600])
def test_fused_cnns_small_and_odd_batches(dev, n):
    """k_cyl_net / k_desc_head / k_cost_net with 0, 1, 3 and a non-round number of workgroups == library convolutions."""
    import torch
    from buffer_amd import registration
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    from buffer_amd.weights import load_weights
    W = load_weights("3dmatch")
    pe = PatchEmbedder(W, dev, THREEDMATCH)
    cv = registration.CostVolume(W, dev)
    g = torch.Generator(device='cpu').manual_seed(10 + n)
    x = torch.rand((n, 16, 420), generator=g).to(dev)
    y = pe.fused(x)
    assert y.shape == (n, 32, 7, 20)
    d, e = pe.head(y)
    assert d.shape == (n, 32) and e.shape == (n, 32, 7, 20)
    a = torch.nn.functional.normalize(torch.rand((n, 32, 5, 20), generator=g), dim=1).to(dev)
    b = torch.nn.functional.normalize(torch.rand((n, 32, 5, 20), generator=g), dim=1).to(dev)
    ind = cv(a, b)
    assert ind.shape == (n,)
    if n:
        from oracle import torch_ref as T          # library convolutions: the test-side torch restatement, on the device
        Wd = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in W.items()}
        want = T.cylindrical_net(x.view(-1, 16, 3, 7, 20), Wd)
        assert (y - want).abs().max().item() < 2e-5 * max(want.abs().max().item(), 1.0)
        wd, we = T.desc_head(y, Wd)
        np.testing.assert_allclose(d.cpu().numpy(), wd.cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ind.cpu().numpy(), T.cost_volume(a, b, Wd).cpu().numpy(), rtol=1e-4, atol=2e-4)
