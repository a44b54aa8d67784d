"""Development probe: the descriptor head kernel (csrc/convnet.hip k_desc_head) against torch (accuracy) and its speed.
   python tools/head_probe.py [patches]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd.config import THREEDMATCH
from buffer_amd.patch_embedder import PatchEmbedder
from buffer_amd.weights import load_weights

dev = torch.device('cuda:0')
pe = PatchEmbedder(load_weights('3dmatch'), dev, THREEDMATCH)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 320000
g = torch.Generator(device='cpu').manual_seed(2)
y = torch.randn((4096, 32, 7, 20), generator=g).to(dev).repeat((P + 4095) // 4096, 1, 1, 1)[:P].contiguous()
d, e = pe.head(y)
h = pe.fused_head
p = h.params.cpu().numpy()
w0, b0, w3, b3 = p[:512].reshape(16, 32), p[512:528], p[528:544], p[544]
ys = y[:256].double().cpu().numpy().reshape(256, 32, 140)
hid = np.maximum(np.einsum('jc,pcx->pjx', w0.astype(np.float64), ys) + b0[None, :, None], 0)
wgt = np.maximum(np.einsum('j,pjx->px', w3.astype(np.float64), hid) + b3, 0)
f = (ys * wgt[:, None, :]).mean(2)
f = f / np.maximum(np.linalg.norm(f, axis=1, keepdims=True), 1e-12)
print(f'desc_head: max |desc - float64| = {np.abs(d[:256].cpu().numpy() - f).max():.3e}', flush=True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    pe.head(y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print(f'desc_head: {P} patches {dt*1e3:.2f} ms  {P*35968/dt/1e9:.0f} GB/s algorithmic')
