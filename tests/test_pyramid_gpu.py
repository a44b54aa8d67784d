"""A1/A2 parity on the GPU: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest
import torch

from buffer_amd import synth

pytestmark = pytest.mark.gpu


def _stack(d):
    pts = np.concatenate([d['src_sds_pts'][:, :3], d['tgt_sds_pts'][:, :3]]).astype(np.float32)
    lens = np.array([len(d['src_sds_pts']), len(d['tgt_sds_pts'])], np.int32)
    return pts, lens


CASES = {
    "config1": lambda: _stack(synth.make_config1_pair()),
    "3dmatch": lambda: _stack(synth.make_pair(3)),
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("radius", [0.07, 0.14])
def test_radius_neighbors_self(case, radius, oracle, dev):
    from buffer_amd import ops
    pts, lens = CASES[case]()
    want = oracle.radius_neighbors(pts, pts, lens, lens, radius)
    got = ops.radius_neighbors(torch.from_numpy(pts).to(dev), torch.from_numpy(pts).to(dev), lens, lens, radius)
    got = got.cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_radius_neighbors_truncated_and_order(oracle, dev):
    from buffer_amd import ops
    pts, lens = CASES["3dmatch"]()
    want = oracle.radius_neighbors(pts, pts, lens, lens, 0.07)
    t = torch.from_numpy(pts).to(dev)
    grid = ops.CellGrid(t, lens, 0.07)
    order = grid.order
    assert np.array_equal(np.sort(order.cpu().numpy()), np.arange(len(pts)))
    for k in (1, 7, 16):
        got, cnt = grid.query(t, lens, k, q_order=order, counts=True)
        assert np.array_equal(got.cpu().numpy(), want[:, :k])
        assert np.array_equal(cnt.cpu().numpy(), (want < len(pts)).sum(1))


def test_radius_neighbors_long_rows(oracle, dev):
    """rows longer than the in-LDS list (multi-sweep path) and k > 64"""
    from buffer_amd import ops
    pts, lens = CASES["config1"]()
    want = oracle.radius_neighbors(pts, pts, lens, lens, 0.5)
    assert want.shape[1] > 130
    got = ops.radius_neighbors(torch.from_numpy(pts).to(dev), torch.from_numpy(pts).to(dev), lens, lens, 0.5)
    assert np.array_equal(got.cpu().numpy(), want)


def test_radius_neighbors_cross_sets(oracle, dev):
    from buffer_amd import ops
    pts, lens = CASES["3dmatch"]()
    sub, sl = oracle.grid_subsample_batch(pts, lens, 0.07)
    for q, ql, s, sl_, r in ((sub, sl, pts, lens, 0.07), (pts, lens, sub, sl, 0.14)):
        want = oracle.radius_neighbors(q, s, ql, sl_, r)
        got = ops.radius_neighbors(torch.from_numpy(q).to(dev), torch.from_numpy(s).to(dev), ql, sl_, r)
        assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("dl", [0.07, 0.14])
def test_grid_subsample(case, dl, oracle, dev):
    from buffer_amd import ops
    pts, lens = CASES[case]()
    want, wl = oracle.grid_subsample_batch(pts, lens, dl)
    got, gl = ops.grid_subsample_batch(torch.from_numpy(pts).to(dev), lens, dl)
    assert np.array_equal(gl, wl)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_empty_and_ragged(oracle, dev):
    from buffer_amd import ops
    rng = np.random.default_rng(0)
    pts = rng.random((1000, 3)).astype(np.float32)
    lens = np.array([0, 700, 0, 300], np.int32)
    want = oracle.radius_neighbors(pts, pts, lens, lens, 0.1)
    got = ops.radius_neighbors(torch.from_numpy(pts).to(dev), torch.from_numpy(pts).to(dev), lens, lens, 0.1)
    assert np.array_equal(got.cpu().numpy(), want)
    want, wl = oracle.grid_subsample_batch(pts, lens, 0.1)
    got, gl = ops.grid_subsample_batch(torch.from_numpy(pts).to(dev), lens, 0.1)
    assert np.array_equal(gl, wl)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
