"""Development probe: the split-f16 Cylindrical_Net kernel (csrc/convnet_h3.hip) against the fp32 Winograd kernel and the stack in
float64: accuracy, speed.   python tools/h3_probe.py [patches]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import ops
from buffer_amd.config import THREEDMATCH
from buffer_amd.patch_embedder import PatchEmbedder
from buffer_amd.weights import load_weights

dev = torch.device('cuda:0')
W = load_weights('3dmatch')
pe = PatchEmbedder(W, dev, THREEDMATCH)
layers = pe.layers
nets = [('winograd f32', ops.CylindricalNet(layers, dev)), ('split f16x3', ops.CylindricalNetSplit(layers, dev))]
if os.environ.get('H3_ONLY'):
    nets = nets[1:]
P = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
g = torch.Generator(device='cpu').manual_seed(0)


def ref64(x):
    h = x.double().reshape(-1, 48, 7, 20)
    for w, b, relu in layers:
        h = torch.cat([h[..., -1:], h, h[..., :1]], -1)
        h = torch.nn.functional.pad(h, (0, 0, 1, 1))
        h = torch.nn.functional.conv2d(h, torch.from_numpy(w).double().to(dev), torch.from_numpy(b).double().to(dev))
        if relu:
            h = torch.relu(h)
    return h


for kind in ('relu(randn)', 'rand', 'signed'):
    x = torch.randn((64, 48, 140), generator=g)
    x = torch.relu(x) if kind == 'relu(randn)' else (torch.rand((64, 48, 140), generator=g) if kind == 'rand' else x)
    x = x.to(dev)
    r = ref64(x)
    for name, net in nets:
        y = net(x).double()
        d = (y - r).abs()
        print(f'{kind:12s} {name:13s}: max err / max |y| = {d.max().item() / r.abs().max().item():.3e}   rms {d.pow(2).mean().sqrt().item() / r.abs().max().item():.3e}'
              f'   (|y|max {r.abs().max().item():.3f})', flush=True)
nets[-1][1].check_range()
x = torch.relu(torch.randn((P, 48, 140), generator=g)).to(dev)
for name, net in nets:
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        net(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f'{name}: {P} patches {dt*1e3:.2f} ms  {P*0.1187/dt/1e3:.1f} dense-equivalent TFLOP/s  -> {dt*1e3*320000/P:.1f} ms per 320000 patches')
