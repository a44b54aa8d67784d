"""A12-A16 -- matching and pose recovery on device (models/BUFFER.py:283-333,335-359,382-464)."""
import numpy as np
import torch

from . import ops
from .patch_embedder import _fold_bn


def mutual_matching(src_des, tgt_des):
    """buffer.mutual_matching (BUFFER.py:335-359): 1-NN both ways (csrc/pointops.hip k_knn), mutual check.
    -> (s_mids, t_mids) int64 device tensors (ascending s_mids, as np.where yields them)."""
    _, s_idx = ops.knn(tgt_des[None], src_des[None], 1)
    _, t_idx = ops.knn(src_des[None], tgt_des[None], 1)
    s_nn, t_nn = s_idx[0, :, 0], t_idx[0, :, 0]
    ar = torch.arange(s_nn.shape[0], device=s_nn.device)
    s_mids = torch.nonzero(t_nn[s_nn] == ar).flatten()
    return s_mids, s_nn[s_mids]


class CostVolume:
    """CostVolume + CostNet (BUFFER.py:37-66, models/patchnet.py:88-147): one fused fp32-MFMA kernel
    (csrc/costnet.hip).  The kernel is written for the released geometry (32 channels, 5 elevation rows,
    azi_n = 20); anything else raises -- there is no library-convolution path in the product."""

    def __init__(self, W, device, azi_n=20, arith='f32'):
        if arith not in ('f32', 'split'):
            raise ValueError(f"cnn_arith must be 'f32' or 'split', got {arith!r}")
        if azi_n != 20:
            raise NotImplementedError(f'CostVolume: the fused kernel is built for azi_n = 20 (got {azi_n})')
        self.azi_n = azi_n
        p = 'Inlier.conv.ops'
        layers = []
        for i, bn in ((0, 1), (3, 4), (6, 7), (9, 10), (12, 13), (15, 16), (18, 19), (21, 22), (24, 25)):
            layers.append(_fold_bn(W[f'{p}.{i}.weight'], W[f'{p}.{i}.bias'], W[f'{p}.{bn}.running_mean'], W[f'{p}.{bn}.running_var']))
        layers.append((np.asarray(W[f'{p}.27.weight'], np.float32), np.asarray(W[f'{p}.27.bias'], np.float32)))
        # 'split': csrc/costnet_h3.hip -- the same network on the f16 matrix pipe, operands split hi + 2^-11 lo' (fp32-equivalent)
        self.fused = ops.CostVolumeNetSplit(layers, device) if arith == 'split' else ops.CostVolumeNet(layers, device)

    def __call__(self, d1, d2):
        """d1,d2 f32[M,32,5,20] -> expected azimuth shift f32[M]."""
        if tuple(d1.shape[1:]) != (32, 5, 20) or d1.shape != d2.shape:
            raise ValueError(f'CostVolume: expected two [M,32,5,20] maps, got {tuple(d1.shape)} and {tuple(d2.shape)}')
        return self.fused(d1, d2)

    def gathered(self, equi, s_rows, t_rows):
        """full maps equi f32[rows,32,7,20] + matched row ids -> expected azimuth shift f32[M] (gather and elevation slice fused)."""
        return self.fused.gathered(equi, s_rows, t_rows)


def recover_pose(ind, ss_kpts, tt_kpts, ss_R, tt_R, cfg, seed=0):
    """BUFFER.py:295-333: hypotheses, all-vs-all scoring, RANSAC on the winner's inliers, refinement.
    -> (pose f32[4,4] device, dict of diagnostics)."""
    R, t, num, best, mask = ops.hypotheses_score(ind, ss_kpts, tt_kpts, ss_R, tt_R, cfg.azi_n, cfg.inlier_th)
    # the winner's inlier list stays on the device (mask -> index list inside the RANSAC entry point): the whole
    # recovery is enqueued without a host round trip
    T, info = ops.ransac_kabsch_masked(ss_kpts, tt_kpts, mask, cfg.ransac_hypotheses, seed, cfg.dist_th, cfg.similar_th)
    if cfg.pose_refine:
        T, rinfo = ops.post_refine(T, ss_kpts, tt_kpts, cfg.refine_threshold, 20)
    return T, dict(inlier_num=num, best=best, inlier_mask=mask, ransac_info=info, R_hyp=R, t_hyp=t)
