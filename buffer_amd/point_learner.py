"""A4/A5 -- point-wise learner on device: EFCNN (reference axis + eps) and DetNet (saliency)
(models/point_learner.py:138-204) driven over the fused VN kernels of csrc/vn.hip."""
import os

import torch
import torch.nn.functional as F

from . import ops


class _ScoreHead:
    """VNStdFeature + Conv1d/InstanceNorm1d head (point_learner.py:128-136,163-171; vn_layers.py:169-222)."""

    def __init__(self, W, p, device, final):
        self.vn1 = ops.VnLayer(W, f'{p}.0.vn1', device, slope=0.0)
        self.vn2 = ops.VnLayer(W, f'{p}.0.vn2', device, slope=0.0)
        self.lin = ops.VnLayer(W, f'{p}.0.vn_lin', device, linear_only=True)
        t = lambda k: torch.as_tensor(W[k], dtype=torch.float32, device=device)
        self.w = [t(f'{p}.{i}.weight')[:, :, 0].contiguous() for i in (1, 3, 5)]
        self.b = [t(f'{p}.{i}.bias').contiguous() for i in (1, 3, 5)]
        self.final = final
        self.fused = not os.environ.get('BUF_SCORE_HEAD_UNFUSED')       # development switch: the 17-launch path

    @staticmethod
    def _inorm(h, seg=None):
        # InstanceNorm1d over the stacked (src+tgt) point axis OF ONE PAIR, biased variance, eps 1e-5 (:131,133).
        # seg = rows per pair (host int array [B]) when several pairs are stacked in one batch; one pair = one segment.
        return ops.segment_instance_norm(h, [h.shape[0]] if seg is None else seg)

    def __call__(self, x, seg=None):
        if self.fused and ops.score_head_supported(x.shape[1], self.vn1, self.vn2, self.lin, self.w):
            # one launch up to the first InstanceNorm, its statistics in two, the normalisation folded into the next Conv1d
            # (csrc/vn.hip buf_score_head: 7 launches instead of 17, bit-identical)
            return ops.score_head(x, [x.shape[0]] if seg is None else seg, self.vn1, self.vn2, self.lin, self.w, self.b, self.final)
        z = ops.vn_pointwise(self.vn1, x)
        z = ops.vn_pointwise(self.vn2, z)
        z = ops.vn_pointwise(self.lin, z)                          # [N, 9]
        h = ops.vn_std(x, z)                                       # [N, 30]
        h = self._inorm(ops.row_linear(h, self.w[0], self.b[0]), seg)
        h = self._inorm(ops.row_linear(h, self.w[1], self.b[1]), seg)
        return ops.row_linear(h, self.w[2], self.b[2], self.final)


class _Decoder:
    def __init__(self, W, p, device):
        self.d1 = ops.VnLayer(W, f'{p}.decoder_blocks.1.mlp', device)
        self.d3 = ops.VnLayer(W, f'{p}.decoder_blocks.3.mlp', device)

    def __call__(self, bottle, skips, ups):
        # nearest_upsample (column 0 of the upsample table, zero for shadows) + skip concat + VNBlock
        y = ops.vn_pointwise(self.d1, skips[1], a=bottle, ind_a=ups[1])
        return ops.vn_pointwise(self.d3, skips[0], a=y, ind_a=ups[0])


class PointLearner:
    def __init__(self, W, device, scale=1.0):
        self.scale = float(scale)
        L = lambda k, **kw: ops.VnLayer(W, k, device, **kw)
        self.b0 = L('Ref.encoder_blocks.0.conv')
        self.res = [dict(conv=L(f'Ref.encoder_blocks.{i}.conv'), unary=L(f'Ref.encoder_blocks.{i}.unary'),
                         short=L(f'Ref.encoder_blocks.{i}.unary_shortcut')) for i in (1, 2, 3, 4)]
        self.ref_dec = _Decoder(W, 'Ref', device)
        self.fc0, self.fc1 = L('Ref.fc_layer.0'), L('Ref.fc_layer.1')
        self.eps_head = _ScoreHead(W, 'Ref.inv_layer', device, 'sigmoid')
        self.key_dec = _Decoder(W, 'Keypt', device)
        self.key_head = _ScoreHead(W, 'Keypt.invar_layer', device, 'softplus')

    def _resnet(self, blk, feats, q, s, idx, strided):
        x = ops.vn_gather_block(blk['conv'], q, s, feats, idx, 1, self.scale)
        sc = ops.gather_max(feats, idx) if strided else feats       # point_learner.py:571-574
        sc = ops.vn_pointwise(blk['short'], sc)
        return ops.vn_pointwise(blk['unary'], x, residual=sc)

    def efcnn(self, pyr, features, seg=None):
        """-> axis f32[N0,3], eps f32[N0,1], bottle f32[N2,120], skips [f32[N0,30], f32[N1,60]]"""
        P, N, PO, UP = pyr['points'], pyr['neighbors'], pyr['pools'], pyr['upsamples']
        x0 = ops.vn_gather_block(self.b0, P[0], P[0], features.contiguous(), N[0], 6, self.scale)
        x1 = self._resnet(self.res[0], x0, P[1], P[0], PO[0], True)
        x2 = self._resnet(self.res[1], x1, P[1], P[1], N[1], False)
        x3 = self._resnet(self.res[2], x2, P[2], P[1], PO[1], True)
        x4 = self._resnet(self.res[3], x3, P[2], P[2], N[2], False)
        skips = [x0, x2]
        y = self.ref_dec(x4, skips, UP)
        axis = ops.vn_pointwise(self.fc1, ops.vn_pointwise(self.fc0, y))
        eps = self.eps_head(y, seg)
        return axis, eps, x4, skips, [x0, x1, x2, x3, x4, y]

    def detnet(self, pyr, bottle, skips, seg=None):
        return self.key_head(self.key_dec(bottle, skips, pyr['upsamples']), seg)


def orient_axes(axis, pts):
    """models/BUFFER.py:244-249."""
    axis = F.normalize(axis, p=2, dim=1)
    mask = (torch.sum(-axis * pts, dim=1) < 0).float().unsqueeze(1)
    return axis * (1 - mask) - axis * mask
