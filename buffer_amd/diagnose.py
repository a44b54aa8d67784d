"""Parity diagnostics of the patch voxelisation (A10, utils/common.py:431-469): why a descriptor row may differ from the reference's.

The voxel ball queries decide `d^2 < r^2` in fp32 for 420 centres x 512 points of an aligned, normalised patch.  The aligned
coordinates come out of a 3 x 3 rotation of fp32 differences (models/patch_embedder.py:123-135): two correct fp32 evaluations of
that product (torch's CPU matmul in the reference, the kernel's un-contracted Rodrigues product here) may differ in the last bits of
a coordinate, and a point that sits ON a ball's surface then falls on the other side: that voxel samples another point and the
descriptor row moves by far more than round-off.  (The rotation matrices themselves agree to ~1e-6 only -- acos / sin / cos of two math
libraries -- so "last bits" means up to ~1e-6 of a coordinate, measured per row.)  These helpers make that explanation checkable per row (numpy only; used by
tests/ and by bench.py's parity record -- nothing here touches the oracle)."""
import numpy as np

COORD_ULPS = 3.0          # |dq_a| <= COORD_ULPS * 2^-24 * ||q||_2: two fp32 evaluation orders of a 3-term dot product of bounded terms


def hit_masks_fp32(patch, centres, radius):
    """bool[C,S]: the reference's fp32 decision for every (centre, point) pair, operations in pointnet2's ball_query order
    ((cx - x)^2 + (cy - y)^2 + (cz - z)^2 < r^2, no contraction)."""
    q = np.asarray(patch, np.float32)
    c = np.asarray(centres, np.float32)
    d = c[:, None, :] - q[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    r = np.float32(radius)
    return d2 < r * r


def surface_margin(patch, centres, radius):
    """For every (centre, point) pair: |d^2 - r^2| in float64 of the fp32 coordinates, divided by the change of d^2 that a last-bits
    difference of the point's coordinates can cause (2 ||c - q||_1 x COORD_ULPS 2^-24 ||q||_2, plus 4 ulp of r^2 for the evaluation of
    d^2 itself) -> float64[C,S]; a value <= 1 means the pair's decision depends on the last bits of the alignment."""
    q = np.asarray(patch, np.float64)
    c = np.asarray(centres, np.float64)
    d = c[:, None, :] - q[None, :, :]
    d2 = (d * d).sum(-1)
    r2 = float(np.float32(radius)) ** 2
    eps_q = COORD_ULPS * 2.0 ** -24 * np.sqrt((q * q).sum(-1))                       # [S]
    tol = 2.0 * np.abs(d).sum(-1) * eps_q[None, :] + 4.0 * float(np.spacing(np.float32(r2)))
    return np.abs(d2 - r2) / tol


COORD_TOL = 4e-6          # two implementations' aligned coordinates agree to the accuracy of their rotation matrices (R: tested to 2e-6 absolute;
#                           acos / sin / cos of two math libraries) times the patch scale (<= 1 after normalisation), plus the product's rounding


def explain_row(ours, centres, radius, theirs=None, coord_tol=COORD_TOL):
    """ours / theirs: the aligned, normalised patch f32[S,3] of ONE keypoint in the two implementations.
    -> dict(near_surface_pairs, min_margin, [coord_max_abs_diff, coords_agree, mask_flips, flips_within_coordinate_difference, explained]).
    explained (with `theirs`): (i) the two patches agree to `coord_tol` (the accuracy of two fp32 Rodrigues rotations), (ii) their fp32 hit
    masks differ in at least one (centre, point) pair and (iii) for EVERY differing pair the decision margin |d^2 - r^2| is within what the
    OBSERVED coordinate difference of that point (plus the rounding of the product) can move d^2:
        |d^2 - r^2| <= 2 ||c - q||_1 (|dq|_inf + COORD_ULPS 2^-24 ||q||) + 4 ulp(r^2).
    Without `theirs`: at least one pair lies within the last-bit uncertainty of a surface (surface_margin <= 1)."""
    m = surface_margin(ours, centres, radius)
    out = dict(near_surface_pairs=int((m <= 1.0).sum()), min_margin=float(m.min()))
    if theirs is None:
        out['explained'] = bool(out['near_surface_pairs'] > 0)
        return out
    a32, b32 = np.asarray(ours, np.float32), np.asarray(theirs, np.float32)
    a, b = a32.astype(np.float64), b32.astype(np.float64)
    c = np.asarray(centres, np.float64)
    dqp = np.abs(a - b).max(-1)                                                      # [S] observed difference per point
    agree = bool(dqp.max() <= coord_tol * max(1.0, float(np.abs(a).max())))
    flips = hit_masks_fp32(a32, centres, radius) != hit_masks_fp32(b32, centres, radius)
    d = c[:, None, :] - a[None, :, :]
    r2 = float(np.float32(radius)) ** 2
    tol = 2.0 * np.abs(d).sum(-1) * (dqp + COORD_ULPS * 2.0 ** -24 * np.sqrt((a * a).sum(-1)))[None, :] + 4.0 * float(np.spacing(np.float32(r2)))
    within = np.abs((d * d).sum(-1) - r2) <= tol
    out.update(coord_max_abs_diff=float(dqp.max()), coords_agree=agree, mask_flips=int(flips.sum()),
               flips_within_coordinate_difference=bool(within[flips].all()))
    out['explained'] = bool(agree and out['mask_flips'] > 0 and out['flips_within_coordinate_difference'])
    return out
