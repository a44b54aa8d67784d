"""buffer_amd/diagnose.py (pure numpy): the per-row explanation of a descriptor difference (INTEGRATION.md caveat 5) accepts exactly the
rows it should -- a point ON a voxel ball surface whose fp32 decision flips under a last-bits coordinate difference -- and nothing else."""
import numpy as np

from buffer_amd import diagnose
from buffer_amd.patch_embedder import voxel_centres

C = voxel_centres(3, 20, 7)
R = 0.8 / 3


def _patch(seed=0):
    rng = np.random.default_rng(seed)
    p = (rng.normal(size=(512, 3)) * 0.35).astype(np.float32)
    p[-1] = 0.0                                    # the keypoint slot
    return p


def _on_surface(p, k=7, c=123, inside=True):
    """move point k onto the ball surface of centre c: the last fp32 position along a ray that the reference's fp32 test still counts
    as inside the ball, or the first one it counts as outside"""
    d = np.array([0.6, -0.64, 0.48], np.float64)
    d /= np.linalg.norm(d)
    p = p.copy()
    last_in = first_out = None
    for t in np.linspace(1 - 4e-7, 1 + 4e-7, 161):
        p[k] = (C[c].astype(np.float64) + d * (R * t)).astype(np.float32)
        hit = bool(diagnose.hit_masks_fp32(p[k:k + 1], C[c:c + 1], R)[0, 0])
        if hit:
            last_in = p[k].copy()
        elif first_out is None and last_in is not None:
            first_out = p[k].copy()
    assert last_in is not None and first_out is not None
    p[k] = last_in if inside else first_out
    return p


def test_hit_masks_follow_the_reference_operation_order():
    p = _patch()
    m = diagnose.hit_masks_fp32(p, C, R)
    d = C[:, None, :].astype(np.float32) - p[None]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    assert m.shape == (420, 512) and np.array_equal(m, d2 < np.float32(R) * np.float32(R)) and 0 < m.sum() < m.size


def test_a_random_patch_has_no_surface_pair_and_explains_nothing():
    p = _patch(1)
    e = diagnose.explain_row(p, C, R)
    assert e['near_surface_pairs'] == 0 and e['min_margin'] > 1.0 and e['explained'] is False
    # the other implementation's patch differs in its last bits, nothing flips: a descriptor difference would NOT be explained
    q = p.copy()
    q.view(np.int32)[:, 0] += 1
    e2 = diagnose.explain_row(p, C, R, theirs=q)
    assert e2['coords_agree'] and e2['mask_flips'] == 0 and e2['explained'] is False


def test_a_flip_on_a_ball_surface_is_explained_and_only_that():
    base = _patch(2)
    ours = _on_surface(base, inside=True)
    theirs = _on_surface(base, inside=False)
    ma, mb = diagnose.hit_masks_fp32(ours, C, R), diagnose.hit_masks_fp32(theirs, C, R)
    assert (ma != mb).sum() >= 1 and ma[123, 7] != mb[123, 7]
    e = diagnose.explain_row(ours, C, R, theirs=theirs)
    assert e['coords_agree'] and e['mask_flips'] >= 1 and e['flips_within_coordinate_difference'] and e['explained'] is True
    assert diagnose.explain_row(ours, C, R)['near_surface_pairs'] >= 1                   # the weak form sees the pair too
    # a patch that disagrees by far more than two fp32 rotations can: not "last bits", not explained even though masks flip
    far = ours.copy()
    far[:, 1] += np.float32(3e-4)
    e3 = diagnose.explain_row(ours, C, R, theirs=far)
    assert e3['mask_flips'] > 0 and e3['coords_agree'] is False and e3['explained'] is False
    # a flip whose margin is larger than the observed coordinate difference can move (point well inside one ball in `ours`, pushed out in
    # `theirs` by changing ANOTHER coordinate representation than the one compared): emulate by flipping a mask bit through a big move of
    # one point while every other point is identical -> the moved point's own difference explains it only if the move is that large
    moved = ours.copy()
    k = int(np.argmax(ma.sum(0)[:-1]))                                                   # a point inside several balls
    moved[k] += np.float32(1e-6)                                                         # tiny move: no flip expected for an interior point
    e4 = diagnose.explain_row(ours, C, R, theirs=moved)
    assert e4['explained'] is (e4['mask_flips'] > 0 and e4['flips_within_coordinate_difference'])
