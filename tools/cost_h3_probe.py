"""Development probe: the split-f16 cost net (csrc/costnet_h3.hip) against the fp32 kernel and the network in float64 (torch):
accuracy of the expected shift index, speed.   python tools/cost_h3_probe.py [matches]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import registration
from buffer_amd.weights import load_weights
from oracle import torch_ref as T

dev = torch.device('cuda:0')
W = load_weights('3dmatch')
nets = [('fp32 kernel', registration.CostVolume(W, dev)), ('split f16x3', registration.CostVolume(W, dev, arith='split'))]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
g = torch.Generator(device='cpu').manual_seed(1)
a = torch.nn.functional.normalize(torch.rand((256, 32, 5, 20), generator=g), dim=1).to(dev)
b = torch.nn.functional.normalize(torch.rand((256, 32, 5, 20), generator=g), dim=1).to(dev)
W64 = {k: torch.from_numpy(np.asarray(v)).double().to(dev) for k, v in W.items() if k.startswith('Inlier')}
ref = T.cost_volume(a.double(), b.double(), W64)
for name, net in nets:
    y = net(a, b).double()
    print(f'{name:12s}: max |ind - ind64| = {(y - ref).abs().max().item():.3e}   (ind in [0, 20))', flush=True)
nets[1][1].fused.check_range()
a = torch.nn.functional.normalize(torch.rand((M, 32, 5, 20), generator=g), dim=1).to(dev)
b = torch.nn.functional.normalize(torch.rand((M, 32, 5, 20), generator=g), dim=1).to(dev)
for name, net in nets:
    for _ in range(2):
        net(a, b)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        net(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f'{name}: {M} matches {dt*1e3:.2f} ms  ({dt/M*1e9:.0f} ns per match, {M*0.16/dt/1e3:.1f} dense-equivalent TFLOP/s)')
