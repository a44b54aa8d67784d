"""A0/A3 -- the collate stage on device (counterpart of ThreeDMatch/dataloader.py:18-51,115-245).

The reference runs 7 KD-tree radius searches and 2 grid subsamplings per pair on CPU worker
processes; here the stacked 3.5 cm clouds are already in HBM and three cell grids (one per layer)
serve all seven searches.
"""
import numpy as np
import torch

from . import ops


def build_pyramid(points0, lengths0, limits, cfg, want_counts=False):
    """points0 f32[N0,3] (src then tgt stacked, device), lengths0 int[2] ->
    dict(points[3], lengths[3] (numpy int32), neighbors[3], pools[2], upsamples[2]) with int32 tables
    (shadow index = number of support rows).  Layer l+1 rows are in ascending voxel-key order."""
    r = cfg.voxel_size_0 * cfg.conv_radius                       # dataloader.py:142
    pts, lens = points0, np.asarray(lengths0, np.int32)
    P, L, N, PO, UP, CNT = [], [], [], [], [], []
    grid = ops.CellGrid(pts, lens, r)
    order = grid.order
    for layer in range(3):
        k = int(limits[layer])
        if want_counts:
            conv, cnt = grid.query(pts, lens, k, q_order=order, counts=True)
            CNT.append(cnt)
        else:
            conv = grid.query(pts, lens, k, q_order=order)
        P.append(pts); L.append(lens); N.append(conv)
        if layer == 2:
            break
        dl = 2 * r / cfg.conv_radius                              # dataloader.py:187
        sub, sl = ops.grid_subsample_batch(pts, lens, dl)
        PO.append(grid.query(sub, sl, k))                         # pool: queries l+1, supports l, radius r
        grid_next = ops.CellGrid(sub, sl, 2 * r)
        UP.append(grid_next.query(pts, lens, k, radius=2 * r, q_order=order))   # upsample: radius 2r (:200)
        # layer l+1 rows are already cell-coherent: no q_order.  (With the grid's own order these self queries would run on the cell-centric
        # kernel; at the coarser layers a third of the cells hold more candidates than its LDS stage and the lane-per-query pass that
        # takes them over costs more than the kernel saves: measured, 1.4 -> 2.1 ms of A2 per 32-pair step.)
        pts, lens, grid, order = sub, sl, grid_next, None
        r *= 2.0
    out = dict(points=P, lengths=L, neighbors=N, pools=PO, upsamples=UP)
    if want_counts:
        out['counts'] = CNT
    return out


def calibrate_limits(samples, cfg, device, keep_ratio=0.8, samples_threshold=2000):
    """calibrate_neighbors (dataloader.py:18-51): 80th percentile of the neighbour-count histogram per
    layer, over samples until every layer holds > 2000 counts.  samples: iterable of sample dicts."""
    hist_n = cfg.hist_n
    hists = np.zeros((3, hist_n), np.int64)
    for s in samples:
        pts, lens = stack_sample(s, device)[:2]
        pyr = build_pyramid(pts, lens, [hist_n] * 3, cfg, want_counts=True)
        for l, c in enumerate(pyr['counts']):
            c = torch.clamp(c, max=hist_n).long()                 # the table is cut at hist_n columns
            hists[l] += torch.bincount(c, minlength=hist_n + 1)[:hist_n].cpu().numpy()
        if np.min(np.sum(hists, axis=1)) > samples_threshold:
            break
    cumsum = np.cumsum(hists.T, axis=0)
    return np.sum(cumsum < (keep_ratio * cumsum[hist_n - 1, :]), axis=0)


def stack_sample(sample, device):
    """sample dict (ThreeDMatch/dataset.py:155-161) -> stacked device tensors (dataloader.py:121-141)."""
    src, tgt = sample['src_sds_pts'], sample['tgt_sds_pts']
    pts = torch.from_numpy(np.concatenate([src[:, :3], tgt[:, :3]], 0).astype(np.float32)).to(device)
    feats = torch.from_numpy(np.concatenate([src[:, 3:], tgt[:, 3:]], 0).astype(np.float32)).to(device)
    lens = np.array([len(src), len(tgt)], np.int32)
    src_raw = torch.from_numpy(sample['src_fds_pts'].astype(np.float32)).to(device)
    tgt_raw = torch.from_numpy(sample['tgt_fds_pts'].astype(np.float32)).to(device)
    return pts, lens, feats, src_raw, tgt_raw
