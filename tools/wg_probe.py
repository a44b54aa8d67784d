"""Development probe: the Winograd-form Cylindrical_Net kernel (csrc/convnet_wg.hip) against the direct-form one
(tests/native/convnet_direct.hip, test infrastructure): accuracy vs the stack in float64, speed.   python tools/wg_probe.py [patches]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import ops
from buffer_amd.config import THREEDMATCH
from buffer_amd.patch_embedder import PatchEmbedder, _fold_bn
from buffer_amd.weights import load_weights

dev = torch.device('cuda:0')
W = load_weights('3dmatch')
pe = PatchEmbedder(W, dev, THREEDMATCH)
layers = pe.layers
wino = ops.CylindricalNet(layers, dev)
if not os.environ.get('WG_ONLY'):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from util import DirectCylindricalNet
    direct = DirectCylindricalNet(layers, dev)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
names = [('winograd', wino)] if os.environ.get('WG_ONLY') else [('direct', direct), ('winograd', wino)]
g = torch.Generator(device='cpu').manual_seed(0)
x = torch.relu(torch.randn((P, 48, 140), generator=g)).to(dev)


def ref64(x):
    h = x.double().reshape(-1, 48, 7, 20)
    for w, b, relu in layers:
        h = torch.cat([h[..., -1:], h, h[..., :1]], -1)
        h = torch.nn.functional.pad(h, (0, 0, 1, 1))
        h = torch.nn.functional.conv2d(h, torch.from_numpy(w).double().to(dev), torch.from_numpy(b).double().to(dev))
        if relu:
            h = torch.relu(h)
    return h


n = min(P, 64)
r = ref64(x[:n])
for name, net in names:
    y = net(x[:n]).double()
    e = (y - r).abs().max().item() / r.abs().max().item()
    print(f'{name}: max err / max |y| = {e:.3e}   (|y|max {r.abs().max().item():.3f})', flush=True)
for name, net in names:
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        net(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f'{name}: {P} patches {dt*1e3:.2f} ms  {P*0.1187/dt/1e3:.1f} dense-equivalent TFLOP/s')

