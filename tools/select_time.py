"""Development aid: k_select_patches_grid (ops.select_patches_batched) at the bench's step size: 64 clouds x 5000 keypoints."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import ops, synth
dev = torch.device('cuda:0')
npairs, m = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 5000
samples = [synth.make_pair(3000 + i) for i in range(8)]
clouds = []
for i in range(npairs):
    s = samples[i % 8]
    clouds += [torch.from_numpy(s['src_fds_pts'].astype(np.float32)).to(dev), torch.from_numpy(s['tgt_fds_pts'].astype(np.float32)).to(dev)]
sup, lens = ops.permute_clouds(clouds, [ops.perm_key(i // 2, i % 2) for i in range(len(clouds))])
g = torch.Generator(device='cpu').manual_seed(0)
off = np.concatenate([[0], np.cumsum(lens)])
kp = torch.cat([sup[off[c]:off[c + 1]][torch.randint(0, int(lens[c]), (m,), generator=g).to(dev)] for c in range(len(clouds))]).contiguous()
out = ops.select_patches_batched(sup, lens, kp, m, 0.3, 512)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t = time.perf_counter(); ops.select_patches_batched(sup, lens, kp, m, 0.3, 512, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(f'select_patches_batched {len(clouds)} clouds x {m} keypoints: {np.median(ts):.3f} ms per call (grid build included), checksum {float(out.double().sum()):.6f}')
