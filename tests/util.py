"""Helpers shared by the parity tests."""
import numpy as np


def d2_table(q, s, tab):
    """fp32 ((dx*dx + dy*dy) + dz*dz) for every table entry; shadow entries -> +inf."""
    q = np.asarray(q, np.float32)
    s_pad = np.concatenate([np.asarray(s, np.float32), np.full((1, 3), np.inf, np.float32)], 0)
    tab = np.asarray(tab)
    d = q[:, None, :] - s_pad[np.minimum(tab, len(s_pad) - 1)]
    sq = d * d
    out = (sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]
    out = out.astype(np.float32)
    out[tab >= len(s)] = np.inf
    return out


def assert_neighbors_equal_mod_ties(ours, ref, q, s):
    """Index-exact except inside groups of EXACTLY equal d2 (the reference's order there is whatever
    std::sort leaves from the KD-tree traversal; ours is ascending index).  Distance profiles must be
    identical, rows duplicate-free, and every disagreement must sit inside an exact tie."""
    ours, ref = np.asarray(ours), np.asarray(ref)
    assert ours.shape == ref.shape
    do, dr = d2_table(q, s, ours), d2_table(q, s, ref)
    assert np.array_equal(do, dr), "distance profiles differ"
    diff = ours != ref
    if diff.any():
        rows = np.where(diff.any(1))[0]
        for r in rows:
            cols = np.where(diff[r])[0]
            for c in cols:
                tie = (c > 0 and do[r, c - 1] == do[r, c]) or (c + 1 < do.shape[1] and do[r, c + 1] == do[r, c]) \
                    or c == do.shape[1] - 1
                assert tie, f"row {r} col {c}: {ours[r, c]} vs {ref[r, c]} without a d2 tie"
            real = ours[r][ours[r] < len(s)]
            assert len(np.unique(real)) == len(real)
    return int(diff.any(1).sum())
