"""Development aid: furthest point sampling of two 14 000-point clouds (5000 samples), in voxel-key order and shuffled."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from buffer_amd import ops, synth

dev = torch.device('cuda:0')
d = synth.make_pair(3, n_raw=250_000)
for order in ('voxel_key', 'cells12', 'cells25', 'morton6', 'shuffled'):
    clouds = []
    for key in ('src_fds_pts', 'tgt_fds_pts'):
        p = d[key][:, :3].astype(np.float32)[:14000]
        if order == 'voxel_key':
            k = np.floor(p / 0.025).astype(np.int64)
            p = p[np.lexsort((k[:, 0], k[:, 1], k[:, 2]))]
        elif order in ('cells12', 'cells25'):
            k = np.floor(p / (0.12 if order == 'cells12' else 0.25)).astype(np.int64)
            p = p[np.lexsort((k[:, 0], k[:, 1], k[:, 2]))]
        elif order == 'morton6':
            k = np.floor((p - p.min(0)) / 0.06).astype(np.int64)
            code = np.zeros(len(p), np.int64)
            for bit in range(10):
                for a in range(3):
                    code |= ((k[:, a] >> bit) & 1) << (3 * bit + a)
            p = p[np.argsort(code, kind='stable')]
        else:
            p = p[np.random.default_rng(0).permutation(len(p))]
        clouds.append(p)
    flat = torch.from_numpy(np.concatenate(clouds)).to(dev)
    lens = np.asarray([len(c) for c in clouds])
    ops.furthest_point_sample_ragged(flat, lens, 5000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ops.furthest_point_sample_ragged(flat, lens, 5000)
    torch.cuda.synchronize()
    print(f'{order}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms ({lens.tolist()} points, 5000 samples)')
