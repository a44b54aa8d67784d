"""Size-independent properties at BASELINE full sizes (where the oracle would take too long):
sortedness, symmetry, self-at-column-0, mass conservation of the barycentres, FPS monotonicity."""
import numpy as np
import pytest
import torch

from buffer_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(dev):
    """16 stacked 3DMatch-shape pairs (32 clouds, ~340k points)"""
    pairs = [synth.make_pair(4000 + i) for i in range(4)]
    pts, lens = [], []
    for i in range(16):
        s = pairs[i % 4]
        off = np.array([0.0, 0.0, 0.01 * (i // 4)])          # distinct copies
        pts += [s['src_sds_pts'][:, :3] + off, s['tgt_sds_pts'][:, :3] + off]
        lens += [len(s['src_sds_pts']), len(s['tgt_sds_pts'])]
    P = torch.from_numpy(np.concatenate(pts).astype(np.float32)).to(dev)
    return P, np.array(lens, np.int32)


def _d2(P, tab):
    n = P.shape[0]
    Pp = torch.cat([P, torch.full((1, 3), float('inf'), device=P.device)])
    d = P[:, None, :] - Pp[tab.long()]
    sq = d * d
    out = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
    out[tab >= n] = float('inf')
    return out


def test_radius_rows_sorted_self_first_symmetric(big, dev):
    from buffer_amd import ops
    P, lens = big
    n = P.shape[0]
    r = 0.07
    grid = ops.CellGrid(P, lens, r)
    mc = torch.zeros(1, dtype=torch.int32, device=dev)
    grid.query(P, lens, 0, max_count=mc)
    K = int(mc.item())
    tab, cnt = grid.query(P, lens, K, q_order=grid.order, counts=True)
    assert torch.equal(tab[:, 0].long(), torch.arange(n, device=dev)), "self must be the nearest neighbour"
    d2 = _d2(P, tab)
    assert bool((d2[:, 1:] >= d2[:, :-1]).all()), "rows must ascend by squared distance"
    real = tab < n
    assert torch.equal(real.sum(1).int(), cnt)
    assert bool((d2[real] < np.float32(r) * np.float32(r)).all())
    # padding only at the tail, with the stacked support count
    assert bool((real[:, 1:] <= real[:, :-1]).all()) and bool((tab[~real] == n).all())
    # symmetry of the relation (same cloud, same radius): edge (i,j) present <=> (j,i) present
    i = torch.arange(n, device=dev)[:, None].expand_as(tab)[real]
    j = tab[real].long()
    fwd = torch.unique(i * n + j)
    bwd = torch.unique(j * n + i)
    assert torch.equal(fwd, bwd)
    # neighbours never cross batch elements
    off = np.concatenate([[0], np.cumsum(lens)])
    elem = torch.from_numpy(np.repeat(np.arange(len(lens)), lens)).to(dev)
    assert torch.equal(elem[i], elem[j])


def test_truncated_table_is_prefix_of_full(big, dev):
    from buffer_amd import ops
    P, lens = big
    grid = ops.CellGrid(P, lens, 0.07)
    full = grid.query(P, lens, 40)
    for k in (1, 17, 25):
        assert torch.equal(grid.query(P, lens, k), full[:, :k])
    # the result does not depend on the processing order
    assert torch.equal(grid.query(P, lens, 17, q_order=grid.order), full[:, :17])
    # ... including an order that moves queries into another batch element's range of slots (any permutation is allowed:
    # those queries take the lane-per-query pass), with and without rows, and on a query cloud other than the supports
    g = torch.Generator(device='cpu').manual_seed(7)
    perm = torch.randperm(P.shape[0], generator=g).int().to(dev)
    _, cnt_ref = grid.query(P, lens, 17, counts=True)
    tab, cnt = grid.query(P, lens, 17, q_order=perm, counts=True)
    assert torch.equal(tab, full[:, :17]) and torch.equal(cnt, cnt_ref)
    mc = torch.zeros(1, dtype=torch.int32, device=dev)
    _, cnt0 = grid.query(P, lens, 0, q_order=perm, counts=True, max_count=mc)
    assert torch.equal(cnt0, cnt_ref) and int(mc.item()) == int(cnt_ref.max().item())
    Q = P + 0.01
    swap = torch.arange(Q.shape[0], dtype=torch.int32, device=dev).flip(0)          # every slot holds the other element's query
    assert torch.equal(grid.query(Q, lens, 20, q_order=swap), grid.query(Q, lens, 20))


def test_subsample_mass_and_voxel_membership(big, dev):
    from buffer_amd import ops
    P, lens = big
    dl = 0.07
    sub, sl = ops.grid_subsample_batch(P, lens, dl)
    assert sl.sum() == sub.shape[0] and (sl > 0).all()
    # every input point has a barycentre of its own voxel within one voxel diagonal
    g2 = ops.CellGrid(sub, sl, dl * 1.8)
    nn = g2.query(P, lens, 1)
    assert bool((nn[:, 0] < sub.shape[0]).all())
    # row count == number of distinct reference voxel keys; barycentres are convex combinations of their voxel
    o_in = np.concatenate([[0], np.cumsum(lens)])
    o_out = np.concatenate([[0], np.cumsum(sl)])
    for b in (0, 7, len(lens) - 1):
        p32 = P[o_in[b]:o_in[b + 1]].cpu().numpy()
        out = sub[o_out[b]:o_out[b + 1]].cpu().numpy()
        inv = np.float32(1) / np.float32(dl)
        org = np.floor(p32.min(0) * inv) * np.float32(dl)
        key = np.floor((p32 - org) / np.float32(dl)).astype(np.int64)
        uniq, inverse, counts = np.unique(key, axis=0, return_inverse=True, return_counts=True)
        assert len(uniq) == out.shape[0]
        # mass conservation (fp64 reference of the fp32 sums): sum_v count_v * bary_v == sum of points
        sums = np.zeros((len(uniq), 3))
        np.add.at(sums, inverse.reshape(-1), p32.astype(np.float64))
        want = sums / counts[:, None]
        order_w = np.lexsort(want.T[::-1])
        order_g = np.lexsort(out.T[::-1])
        np.testing.assert_allclose(out[order_g], want[order_w], rtol=0, atol=2e-6)


def test_fps_properties_full_size(dev):
    from buffer_amd import ops
    s = synth.make_pair(4100)
    clouds = [s['src_sds_pts'][:, :3].astype(np.float32), s['tgt_sds_pts'][:, :3].astype(np.float32)]
    P = torch.from_numpy(np.concatenate(clouds)).to(dev)
    lens = [len(c) for c in clouds]
    m = 5000
    idx = ops.furthest_point_sample_ragged(P, lens, m)
    batched_equal = ops.furthest_point_sample(torch.from_numpy(clouds[0][None]).to(dev), m)
    assert torch.equal(idx[0], batched_equal[0]), "ragged batch == fixed-size batch"
    o = 0
    for b, c in enumerate(clouds):
        ii = idx[b].long()
        assert int(ii[0]) == 0
        assert len(torch.unique(ii)) == m, "samples are distinct while m <= n"
        sel = torch.from_numpy(c).to(dev)[ii].double()
        # distance of sample r to the set of earlier samples is non-increasing in r (definition of FPS)
        d = torch.cdist(sel[:600], sel[:600])
        d = d + torch.triu(torch.full_like(d, float('inf')))      # keep j < i
        dmin = d.min(1)[0][1:]
        assert bool((dmin[1:] <= dmin[:-1] + 1e-12).all())
        o += len(c)
