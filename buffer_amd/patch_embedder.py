"""A8-A11 -- patch-wise embedder (MiniSpinNet, models/patch_embedder.py) on device.

Every stage is a hand-written kernel behind the C ABI: select_patches (csrc/pointops.hip), the fused
align / voxelise / point-MLP (csrc/voxelize.hip), the dense Cylindrical_Net stack (models/patchnet.py:15-85)
as one fp32-MFMA implicit GEMM with the circular-azimuth / zero-elevation padding folded into its addressing,
and the attention-pooling head (csrc/convnet.hip).  No library convolution runs in the product.
"""
import os

import numpy as np
import torch

from . import ops


def voxel_centres(rad_n, azi_n, ele_n, radius=1.0):
    """get_voxel_coordinate (utils/common.py:248-262,422-428): f32[rad_n*ele_n*azi_n,3], rad->ele->azi."""
    beta = np.linspace(0, np.pi, ele_n, endpoint=False) + np.pi / ele_n / 2
    alpha = np.linspace(0, 2 * np.pi, azi_n, endpoint=False) + np.pi / azi_n
    B, A = np.meshgrid(beta, alpha, indexing='ij')
    B, A = B.flatten(), A.flatten()
    s2 = np.stack([radius * np.sin(B) * np.cos(A), radius * np.sin(B) * np.sin(A), radius * np.cos(B)], 1)
    scale = (np.arange(rad_n) / rad_n + 1 / (2 * rad_n)).reshape(rad_n, 1, 1)
    return (scale * s2[None]).reshape(-1, 3).astype(np.float32)


def _fold_bn(w, b, mean, var, gamma=None, beta=None):
    """conv + BatchNorm(eval) -> one conv (fp64 fold, fp32 storage)."""
    w, b, mean, var = (np.asarray(a, np.float64) for a in (w, b, mean, var))
    s = 1.0 / np.sqrt(var + 1e-5)
    if gamma is not None:
        s = s * np.asarray(gamma, np.float64)
    w2 = w * s.reshape((-1,) + (1,) * (w.ndim - 1))
    b2 = (b - mean) * s
    if beta is not None:
        b2 = b2 + np.asarray(beta, np.float64)
    return w2.astype(np.float32), b2.astype(np.float32)


class PatchEmbedder:
    def __init__(self, W, device, cfg):
        self.cfg, self.device = cfg, device
        self.centres = torch.from_numpy(voxel_centres(cfg.rad_n, cfg.azi_n, cfg.ele_n)).to(device)
        ang = -np.arange(cfg.azi_n) * (2 * np.pi / cfg.azi_n)         # var_to_invar, utils/common.py:485-491
        self.azi_cs = torch.from_numpy(np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32)).to(device)
        g = np.asarray(W['Desc.pnt_layer.1.weight'], np.float64)
        s = g / np.sqrt(np.asarray(W['Desc.pnt_layer.1.running_var'], np.float64) + 1e-5)
        self.mlp_w = np.asarray(W['Desc.pnt_layer.0.weight'], np.float32).reshape(16, 3)
        self.mlp_b = np.asarray(W['Desc.pnt_layer.0.bias'], np.float32)
        self.mlp_s = s.astype(np.float32)
        self.mlp_t = (np.asarray(W['Desc.pnt_layer.1.bias'], np.float64)
                      - np.asarray(W['Desc.pnt_layer.1.running_mean'], np.float64) * s).astype(np.float32)
        p = 'Desc.conv_net.ops'
        # fused fp32-MFMA stack (csrc/convnet.hip): BN folded, layer 0's radial depth folds into the channels (c16*3 + d)
        layers = []
        for n, (i, bn) in enumerate(((0, 1), (3, 4), (6, 7), (9, 10), (12, 13), (15, 16), (18, 19))):
            w, b = _fold_bn(W[f'{p}.{i}.weight'], W[f'{p}.{i}.bias'], W[f'{p}.{bn}.running_mean'], W[f'{p}.{bn}.running_var'])
            if n == 0:
                w = w.reshape(w.shape[0], w.shape[1] * w.shape[2], 3, 3)
            layers.append((w, b, True))
        layers.append((np.asarray(W[f'{p}.21.weight'], np.float32), np.asarray(W[f'{p}.21.bias'], np.float32), False))
        self.layers = layers
        arith = getattr(cfg, 'cnn_arith', 'f32')
        if arith not in ('f32', 'split'):
            raise ValueError(f"cnn_arith must be 'f32' or 'split', got {arith!r}")
        # 'split': csrc/convnet_h3.hip -- the same convolutions on the f16 matrix pipe, operands split hi + 2^-11 lo' (fp32-equivalent)
        self.fused = ops.CylindricalNetSplit(layers, device) if arith == 'split' else ops.CylindricalNet(layers, device)
        q = 'Desc.pool_layer'
        w0, b0 = _fold_bn(W[f'{q}.0.weight'], W[f'{q}.0.bias'], W[f'{q}.1.running_mean'], W[f'{q}.1.running_var'],
                          W[f'{q}.1.weight'], W[f'{q}.1.bias'])
        w3, b3 = _fold_bn(W[f'{q}.3.weight'], W[f'{q}.3.bias'], W[f'{q}.4.running_mean'], W[f'{q}.4.running_var'],
                          W[f'{q}.4.weight'], W[f'{q}.4.bias'])
        self.fused_head = ops.DescriptorHead(w0, b0, w3, b3, device)

    def head(self, x):
        """attention pooling + normalisation (patch_embedder.py:81-84), one fused launch."""
        return self.fused_head(x)

    def embed_patches(self, patches, axis, want_patches=False):
        """patches f32[Q,S,3] (any number of clouds stacked), axis f32[Q,3] -> dict(desc, equi, R, rand_axis)."""
        cfg = self.cfg
        ax = axis.contiguous() if cfg.dataset in ('3DMatch', '3DLoMatch') else None
        x, R, rand_axis, pn = ops.patch_voxelize(patches, ax, cfg.des_r, self.centres, self.azi_cs,
                                                 cfg.delta / cfg.rad_n, cfg.voxel_sample, self.mlp_w, self.mlp_b,
                                                 self.mlp_s, self.mlp_t, cfg.azi_n, want_patches)
        if hasattr(self.fused, 'with_head') and not os.environ.get('BUF_NO_FUSED_HEAD'):   # split-f16 kernel: the head runs behind the last layer, in LDS (inside the fp32-MFMA kernel it costs more than it saves: DESIGN section 5)
            f, e = self.fused.with_head(x, self.fused_head)
        else:
            f, e = self.head(self.fused(x))
        return dict(desc=f, equi=e, R=R, rand_axis=rand_axis, x=x, patches=pn)

    def __call__(self, pts, kpts, axis, perm=None, want_patches=False):
        """pts f32[N,3] (2 cm cloud), kpts f32[P,3], axis f32[P,3] -> dict(desc, equi, R, rand_axis[, patches])."""
        cfg = self.cfg
        if perm is None:
            perm = torch.randperm(pts.shape[0], device=pts.device)        # patch_embedder.py:97-98
        sup = pts[perm].contiguous()
        patches = ops.select_patches(sup, kpts.contiguous(), cfg.des_r, cfg.num_points_per_patch)
        out = self.embed_patches(patches, axis, want_patches)
        if want_patches:
            out['init_patches'] = patches
        return out
