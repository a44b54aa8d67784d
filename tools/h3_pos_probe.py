import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from buffer_amd import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
widths = [(64, 64)] * 6 + [(64, 32), (32, 32)]
layers = [((rng.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32), (rng.standard_normal(co) * 0.1).astype(np.float32), i < 7) for i, (ci, co) in enumerate(widths)]
net = ops.CylindricalNetSplit(layers, dev)
x = torch.relu(torch.randn((20000, 64, 140))).to(dev)
for _ in range(3):
    net(x)
torch.cuda.synchronize()
