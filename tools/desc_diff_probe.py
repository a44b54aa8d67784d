#!/usr/bin/env python3
"""Development aid (round 5): where do HIP descriptors differ from the CPU oracle's at full size?  One cloud of a full-size pair,
P keypoints: rows with |d desc| > 1e-4, and for them the differences of patches / voxelised maps / conv maps / pooled norm."""
import os
import sys
from dataclasses import replace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from buffer_amd import synth  # noqa: E402
from buffer_amd.config import THREEDMATCH  # noqa: E402
from buffer_amd.pipeline import BufferPipeline  # noqa: E402
from buffer_amd.weights import load_weights  # noqa: E402
from oracle import cpu, torch_ref as T  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device('cuda:0')
cpu.build(ref=False)
cfg = replace(THREEDMATCH, num_keypts=P)
s = synth.make_pair(seed)
pipe = BufferPipeline(cfg, dev)
pipe.calibrate([synth.make_pair(1000)])
rng = np.random.default_rng(0)
perms = [rng.permutation(len(s['src_fds_pts'])), rng.permutation(len(s['tgt_fds_pts']))]
inp = pipe.upload(s)
pose, d = pipe.register(inp, seed=0, perms=[torch.from_numpy(p).to(dev) for p in perms], detail=True)
W = {k: torch.from_numpy(v) for k, v in load_weights(cfg.weights).items()}
for c in range(2):
    kp, ka = d['kpts'][c], d['kaxis'][c]
    raw = inp['src_raw' if c == 0 else 'tgt_raw']
    got = pipe.desc(raw, kp, ka, torch.from_numpy(perms[c]).to(dev), want_patches=True)
    want = T.desc_forward(raw.cpu(), kp.cpu(), ka.cpu(), torch.from_numpy(perms[c]), W, cfg.des_r, cfg.num_points_per_patch, cfg.dataset)
    dd = (got['desc'].cpu() - want['desc']).abs().max(1)[0]
    de = (got['equi'].cpu() - want['equi']).abs().amax((1, 2, 3))
    dp = (got['patches'].cpu() - want['patches']).abs().amax((1, 2))
    di = (got['init_patches'].cpu() - want['init_patches']).abs().amax((1, 2))
    print(f'cloud {c}: max d desc {dd.max():.3e} rows > 1e-4: {(dd > 1e-4).sum().item()}  max d equi {de.max():.3e} rows > 1e-4: {(de > 1e-4).sum().item()}'
          f'  max d patches {dp.max():.3e}  init patches differ in {(di > 0).sum().item()} rows')
    bad = torch.nonzero(dd > 1e-4).flatten().tolist()[:12]
    if not bad:
        continue
    # oracle intermediates for the bad rows
    pw = want['patches'][bad]
    inv = T.spt(pw)
    x = T.point_mlp_max(inv, W).reshape(-1, 16, 3, 7, 20)
    xg = got['x'][bad].cpu().reshape(-1, 16, 3, 7, 20)
    y = T.cylindrical_net(x, W)
    yg = pipe.desc.fused(got['x'][bad].contiguous()).cpu()
    w_ = torch.nn.functional.conv2d(y, W['Desc.pool_layer.0.weight'], W['Desc.pool_layer.0.bias'])
    f, e = T.desc_head(y, W)
    pooled = None
    for j, r in enumerate(bad):
        hits = int(((pw[j] ** 2).sum(1) > 0).sum())
        print(f'  row {r}: d desc {dd[r]:.3e} d equi {de[r]:.3e} d patch {dp[r]:.3e} | d voxel map {(x[j] - xg[j]).abs().max():.3e} (scale {x[j].abs().max():.2f})'
              f' d conv map {(y[j] - yg[j]).abs().max():.3e} (scale {y[j].abs().max():.3f}) | nonzero patch points {hits}'
              f' axis {ka[r].cpu().numpy()} |axis| {float(ka[r].norm()):.6f}')
