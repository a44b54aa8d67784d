"""Micro-benchmarks of single operators on the GPU (development aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from buffer_amd import ops, synth

dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'fps'
if which == 'fps':
    for n in (8000, 11000):
        rng = np.random.default_rng(0)
        pts = torch.from_numpy((rng.random((n, 3)) * 2 + 0.5).astype(np.float32)).to(dev)
        for m in (1500, 5000):
            ops.furthest_point_sample_ragged(pts, [n], m)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                ops.furthest_point_sample_ragged(pts, [n], m)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 3
            print(f'fps n={n} m={m}: {dt*1e3:.2f} ms  {dt/m*1e6:.2f} us/round')
