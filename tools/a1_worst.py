"""Development aid: the worst case of A1 -- every point of a 16384-point element in ONE voxel (the rank inside a bucket is quadratic in its
population in both forms of the operator)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import ops
dev = torch.device('cuda:0')
P = torch.from_numpy(np.tile(np.array([[0.3, 0.4, 0.5]], np.float32), (16384, 1))).to(dev)
lens = np.array([16384], np.int32)
for form in ('1', '0'):
    os.environ['BUF_VOX_FUSED'] = form
    ops.grid_subsample_batch(P, lens, 0.1); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): ops.grid_subsample_batch(P, lens, 0.1)
    torch.cuda.synchronize()
    print('BUF_VOX_FUSED', form, '16384 points in ONE voxel:', (time.perf_counter() - t) / 3 * 1e3, 'ms per call')
