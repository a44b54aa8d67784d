"""Development aid: k_patch_voxelize alone on 3DMatch-shaped input (surface-like radial density), for timing and PMC runs.
    python tools/vox_probe.py [npatch]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buffer_amd import ops
from buffer_amd.weights import load_weights
from oracle import torch_ref as T          # (tool only: voxel centre table)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device('cuda:0')
W = load_weights('3dmatch')
Wt = {k: torch.from_numpy(v) for k, v in W.items()}
s = (Wt['Desc.pnt_layer.1.weight'] / torch.sqrt(Wt['Desc.pnt_layer.1.running_var'] + 1e-5)).numpy()
t = (Wt['Desc.pnt_layer.1.bias'] - Wt['Desc.pnt_layer.1.running_mean'] * torch.from_numpy(s)).numpy()
centres = T.voxel_centres(3, 20, 7).float().to(dev)
ang = -torch.arange(20, dtype=torch.float64) * 2 * np.pi / 20
azi_cs = torch.stack([torch.cos(ang), torch.sin(ang)], 1).float().to(dev)
g = torch.Generator(device=dev).manual_seed(1)
u = torch.randn((n, 512, 3), generator=g, device=dev)
u = u / u.norm(dim=-1, keepdim=True) * torch.rand((n, 512, 1), generator=g, device=dev).sqrt() * 0.3   # density ~ rho, des_r 0.3
u[:, -1] = 0
axis = torch.nn.functional.normalize(torch.randn((n, 3), generator=g, device=dev), dim=1)
args = (u, axis, 0.3, centres, azi_cs, 0.8 / 3, 10, W['Desc.pnt_layer.0.weight'].reshape(16, 3), W['Desc.pnt_layer.0.bias'], s, t, 20, False)
ops.patch_voxelize(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ops.patch_voxelize(*args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f'{n} patches: {dt * 1e3:.3f} ms per call = {dt / n * 320000 * 1e3:.2f} ms per 320 000 patches')
