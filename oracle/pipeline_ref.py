"""TEST INFRASTRUCTURE ONLY -- the whole inference branch of buffer.forward (models/BUFFER.py:231-333)
on CPU, composed from oracle/cpu.py (plain-C index operators, or the compiled reference cores when
use_ref) and oracle/torch_ref.py (torch fp32 restatement).  Used by the end-to-end parity test,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import time

import numpy as np
import torch

from . import cpu
from . import torch_ref as T

_MASK = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _MASK
    return x ^ (x >> 31)


def ransac_kabsch(src, tgt, corr, nhyp=4096, seed=0, max_dist=0.10, edge_sim=0.8):
    """Restatement of OUR deterministic RANSAC (csrc/registration.hip k_ransac; stands in for open3d's
    registration_ransac_based_on_correspondence, models/BUFFER.py:314-326): same sampler, same checks."""
    src, tgt = np.asarray(src, np.float32), np.asarray(tgt, np.float32)
    corr = np.asarray(corr, np.int64)
    n = len(corr)
    if n < 3:
        return np.eye(4, dtype=np.float32), (0, -1)
    best_key, best_T, best_h = (0, np.inf), np.eye(4, dtype=np.float32), -1
    S, G = src[corr].astype(np.float64), tgt[corr].astype(np.float64)
    for h in range(nhyp):
        i0 = _splitmix64(seed + 3 * h) % n
        i1 = _splitmix64(seed + 3 * h + 1) % (n - 1)
        i2 = _splitmix64(seed + 3 * h + 2) % (n - 2)
        if i1 >= i0:
            i1 += 1
        lo, hi = min(i0, i1), max(i0, i1)
        if i2 >= lo:
            i2 += 1
        if i2 >= hi:
            i2 += 1
        ids = [i0, i1, i2]
        a, b = S[ids], G[ids]
        ok = True
        for p in range(3):
            q = (p + 1) % 3
            ds, dt = np.linalg.norm(a[p] - a[q]), np.linalg.norm(b[p] - b[q])
            if not (ds >= dt * edge_sim and dt >= ds * edge_sim):
                ok = False
                break
        if not ok:
            continue
        ca, cb = a.mean(0), b.mean(0)
        H = (a - ca).T @ (b - cb)
        U, _, Vt = np.linalg.svd(H)
        V = Vt.T
        d = np.linalg.det(V @ U.T)
        R = V @ np.diag([1, 1, d]) @ U.T
        t = cb - R @ ca
        if (np.linalg.norm(a @ R.T + t - b, axis=1) > max_dist).any():
            continue
        dist = np.linalg.norm(S @ R.T + t - G, axis=1)
        inl = dist < max_dist
        cnt = int(inl.sum())
        if cnt == 0:
            continue
        mse = float((dist[inl] ** 2).sum() / cnt)
        if cnt > best_key[0] or (cnt == best_key[0] and mse < best_key[1]):
            best_key, best_h = (cnt, mse), h
            best_T = np.eye(4, dtype=np.float32)
            best_T[:3, :3], best_T[:3, 3] = R, t
    return best_T, (best_key[0], best_h)


def register_pair(sample, W, limits, cfg, seed=0, perms=None, num_keypts=None, use_ref=False, ransac=True,
                  timings=None):
    """-> (pose f32[4,4], detail dict).  perms: the two support-cloud permutations (select_patches)."""
    tm = {} if timings is None else timings

    def tick(name, t0):
        tm[name] = tm.get(name, 0.0) + time.perf_counter() - t0

    P = cfg.num_keypts if num_keypts is None else num_keypts
    t0 = time.perf_counter()
    batch = T.collate(sample, limits, cfg.voxel_size_0, cfg.conv_radius, use_ref)
    tick('pyramid', t0)
    n_src = int(batch['stack_lengths'][0][0])
    with torch.no_grad():
        t0 = time.perf_counter()
        axis, eps, bottle, skips = T.efcnn_forward(batch, W, cfg.scale)
        score = T.detnet_forward(batch, bottle, skips, W)
        tick('point_learner', t0)
        pts0 = batch['points'][0]
        kp, ka, res = [], [], []
        t0 = time.perf_counter()
        for lo, hi in ((0, n_src), (n_src, pts0.shape[0])):
            p = pts0[lo:hi]
            a = T.orient_axes(axis[lo:hi], p)
            keep = torch.where(score[lo:hi, 0] > cfg.keypts_th)[0]
            p, a = p[keep], a[keep]
            idx = torch.from_numpy(cpu.fps(p[None].numpy(), P)).long()[0]
            kp.append(p[idx])
            ka.append(a[idx])
        tick('keypoints', t0)
        t0 = time.perf_counter()
        for i, raw in enumerate((batch['src_pcd_raw'], batch['tgt_pcd_raw'])):
            perm = torch.as_tensor(perms[i]).long()
            res.append(T.desc_forward(raw, kp[i], ka[i], perm, W, cfg.des_r, cfg.num_points_per_patch, cfg.dataset))
        tick('descriptors', t0)
        t0 = time.perf_counter()
        s_mids, t_mids = T.mutual_matching(res[0]['desc'], res[1]['desc'])
        ss_kpts, tt_kpts = kp[0][s_mids], kp[1][t_mids]
        e = cfg.ele_n
        inds = []
        for s in range(0, len(s_mids), 128):
            inds.append(T.cost_volume(res[0]['equi'][s_mids[s:s + 128]][:, :, 1:e - 1],
                                      res[1]['equi'][t_mids[s:s + 128]][:, :, 1:e - 1], W))
        ind = torch.cat(inds)
        R, t = T.hypotheses(ind, ss_kpts, tt_kpts, res[0]['R'][s_mids], res[1]['R'][t_mids], cfg.azi_n)
        num, best, inl = T.score_hypotheses(R, t, ss_kpts, tt_kpts, cfg.azi_n, cfg.inlier_th)
        tick('matching', t0)
        t0 = time.perf_counter()
        if ransac:
            Tm, info = ransac_kabsch(ss_kpts.numpy(), tt_kpts.numpy(), inl.numpy(), cfg.ransac_hypotheses, seed,
                                     cfg.dist_th, cfg.similar_th)
        else:
            Tm = np.eye(4, dtype=np.float32)
            Tm[:3, :3], Tm[:3, 3] = R[best].numpy(), t[best].numpy()
            info = (int(num[best]), best)
        pose = torch.from_numpy(Tm)[None]
        if cfg.pose_refine:
            pose = T.post_refinement(pose, ss_kpts[None], tt_kpts[None], cfg.refine_threshold)
        tick('pose', t0)
    return pose[0].numpy(), dict(axis=axis, score=score, kpts=kp, desc=res, s_mids=s_mids, t_mids=t_mids, ind=ind,
                                 inlier_num=num, best=best, inlier_ind=inl, ransac_info=info)
