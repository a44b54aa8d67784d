"""Development probe: the fused cost-volume network (csrc/costnet.hip) against library convolutions (accuracy) and its speed.
   python tools/cost_probe.py [matches]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import registration
from buffer_amd.weights import load_weights
from oracle import torch_ref as T

dev = torch.device('cuda:0')
W = load_weights('3dmatch')
cv = registration.CostVolume(W, dev)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
g = torch.Generator(device='cpu').manual_seed(1)
a = torch.nn.functional.normalize(torch.rand((M, 32, 5, 20), generator=g), dim=1).to(dev)
b = torch.nn.functional.normalize(torch.rand((M, 32, 5, 20), generator=g), dim=1).to(dev)
Wd = {k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in W.items()}
want = T.cost_volume(a[:200], b[:200], Wd)
got = cv(a[:200], b[:200])
print(f'cost_net: max |ind - library convs| = {(got - want).abs().max().item():.3e}', flush=True)
for _ in range(2):
    cv(a, b)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    cv(a, b)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f'cost_net: {M} matches {dt*1e3:.2f} ms  {M*0.05191/dt/1e3:.1f} TFLOP/s executed ({M*0.160/dt/1e3:.1f} dense-equivalent)')
