#!/usr/bin/env python3
"""Development aid: time of one mutual-matching 1-NN call (32 x 5000 x 5000, d = 32) and its kernels' share."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
b, n = 32, 5000
d = torch.nn.functional.normalize(torch.randn((2, b, n, 32), generator=g), dim=3).to(dev)
for _ in range(3):
    ops.knn(d[0], d[1], 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ops.knn(d[0], d[1], 1)
torch.cuda.synchronize()
print(f'knn k=1 d=32, {b} x {n} x {n}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call')
