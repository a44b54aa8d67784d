"""TEST INFRASTRUCTURE ONLY (see oracle/buffer_oracle.h): restatement of open3d 0.13.0's
`registration_ransac_based_on_correspondence`, the call the reference makes at models/BUFFER.py:314-326
(PointToPoint(with_scaling=False), ransac_n = 3, EdgeLength(similar_th) + Distance(dist_th) checkers,
RANSACConvergenceCriteria(iter_n = 50000, confidence)).

open3d is a pip dependency of the reference (README.md:28: open3d==0.13.0), absent from /root/reference and from this
image: PARITY UNPINNED -- the loop below follows the published algorithm of that release as recalled:

    for itr in 0 .. max_iteration-1, while itr < exit_itr:
        sample ransac_n correspondences uniformly WITH replacement (UniformRandInt per slot)
        T = Umeyama / Kabsch (no scaling) on the sample, in double
        edge-length checker: every pair (a, b) of the sample must satisfy |s_a - s_b| >= sim * |t_a - t_b| and
                             |t_a - t_b| >= sim * |s_a - s_b|
        distance checker:    |T s_a - t_a| <= dist for every sampled a
        evaluate over ALL correspondences: inliers = |T s - t| < max_dist; fitness = inliers / n, rmse over inliers
        better = higher fitness, then lower rmse;  on improvement: exit_itr = ceil(log(1 - confidence) / log(1 - fitness^3))
                                                   when that is below max_iteration

open3d runs the loop under OpenMP with a random_device-seeded mt19937 and is not bit-reproducible; this restatement is
its one-thread schedule with a seeded numpy generator, so recall statistics -- not poses -- are what can be compared."""
import numpy as np


def kabsch(src, dst):
    """Eigen::umeyama(src, dst, with_scaling=false): rigid T with dst ~ R src + t.  src, dst f64[..., n, 3] -> [..., 4, 4]."""
    mu_s, mu_d = src.mean(-2, keepdims=True), dst.mean(-2, keepdims=True)
    sigma = np.swapaxes(dst - mu_d, -1, -2) @ (src - mu_s) / src.shape[-2]
    U, _, Vt = np.linalg.svd(sigma)
    S = np.ones(sigma.shape[:-2] + (3,))
    S[..., 2] = np.where(np.linalg.det(U) * np.linalg.det(Vt) < 0, -1.0, 1.0)
    R = (U * S[..., None, :]) @ Vt
    T = np.zeros(sigma.shape[:-2] + (4, 4))
    T[..., :3, :3] = R
    T[..., :3, 3] = mu_d[..., 0, :] - (R @ mu_s[..., 0, :, None])[..., 0]
    T[..., 3, 3] = 1.0
    return T


def ransac_correspondence(src, tgt, corr, max_dist, edge_similarity=0.8, distance_threshold=None, max_iteration=50000,
                          confidence=0.999, seed=0, chunk=2048):
    """src f64[n,3], tgt f64[m,3], corr int[k,2] -> dict(T f64[4,4], fitness, inlier_rmse, iterations, evaluated)."""
    src, tgt = np.asarray(src, np.float64), np.asarray(tgt, np.float64)
    corr = np.asarray(corr, np.int64).reshape(-1, 2)
    best = dict(T=np.eye(4), fitness=0.0, inlier_rmse=0.0, iterations=0, evaluated=0)
    k = corr.shape[0]
    if k < 3 or max_dist <= 0:
        return best
    dist_th = max_dist if distance_threshold is None else distance_threshold
    rng = np.random.default_rng(seed)
    S, D = src[corr[:, 0]], tgt[corr[:, 1]]
    exit_itr = max_iteration
    itr = 0
    while itr < min(exit_itr, max_iteration):
        nb = min(chunk, max_iteration - itr)
        pick = rng.integers(0, k, size=(nb, 3))                       # with replacement, like UniformRandInt per slot
        s, d = S[pick], D[pick]                                       # [nb,3,3]
        T = kabsch(s, d)
        ok = np.ones(nb, bool)
        for a, b in ((0, 1), (0, 2), (1, 2)):                         # CorrespondenceCheckerBasedOnEdgeLength
            ls, lt = np.linalg.norm(s[:, a] - s[:, b], axis=1), np.linalg.norm(d[:, a] - d[:, b], axis=1)
            ok &= (ls >= lt * edge_similarity) & (lt >= ls * edge_similarity)
        moved = s @ np.swapaxes(T[:, :3, :3], 1, 2) + T[:, None, :3, 3]
        ok &= (np.linalg.norm(moved - d, axis=2) <= dist_th).all(1)   # CorrespondenceCheckerBasedOnDistance
        ok &= np.isfinite(T).all((1, 2))
        for j in np.nonzero(ok)[0]:
            if itr + j >= exit_itr:
                break
            d2 = ((S @ T[j, :3, :3].T + T[j, :3, 3] - D) ** 2).sum(1)
            inl = d2 < max_dist * max_dist
            good = int(inl.sum())
            best['evaluated'] += 1
            if good == 0:
                continue
            fit, rmse = good / k, float(np.sqrt(d2[inl].sum() / good))
            if fit > best['fitness'] or (fit == best['fitness'] and rmse < best['inlier_rmse']):
                best.update(T=T[j].copy(), fitness=fit, inlier_rmse=rmse)
                with np.errstate(divide='ignore'):
                    e = np.log(1.0 - confidence) / np.log(1.0 - fit ** 3) if fit < 1.0 and confidence < 1.0 else \
                        (0.0 if fit >= 1.0 else np.inf)
                if e < max_iteration:
                    exit_itr = int(np.ceil(e))
        itr += nb
    best['iterations'] = int(min(itr, exit_itr, max_iteration))
    return best
