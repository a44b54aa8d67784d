import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from buffer_amd import preprocess, stream, ops
from buffer_amd.config import THREEDMATCH as cfg
dev = torch.device('cuda:0')
raws = stream.generate(16, dev)
rl = [r[f'{s}_raw'] for r in raws for s in ('src', 'tgt')]
cat = torch.cat(rl)
fds, fl = preprocess.voxel_down_sample_batch(cat, [int(r.shape[0]) for r in rl], cfg.downsample)
sds, sl = preprocess.voxel_down_sample_batch(fds, fl, cfg.voxel_size_0)
pts = sds.float().contiguous()
n0 = int(sl[0])
g = torch.Generator(device=dev).manual_seed(0)
sel = torch.randperm(n0, generator=g, device=dev)[:256]
d = torch.cdist(pts[:n0][sel].double(), pts[:n0].double())
kth = torch.topk(d, 30, dim=1, largest=False).values[:, -1]
r = float(kth.median().item()) * 1.1
grid = ops.CellGrid(pts, sl, r)
out, cnt = grid.query(pts, sl, 40, counts=True)
c = cnt.float()
print('radius', r, 'n', pts.shape[0], 'count: median', c.median().item(), 'mean', c.mean().item(), 'frac > 64:', (c > 64).float().mean().item(), 'frac < 30:', (c < 30).float().mean().item(), 'max', c.max().item())
for scale in (0.9, 1.0, 1.1, 1.2):
    rr = float(kth.median().item()) * scale
    grid = ops.CellGrid(pts, sl, rr)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): out, cnt = grid.query(pts, sl, 40, counts=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    c = cnt.float()
    print(f'scale {scale}: query {dt*1e3:.2f} ms  >64: {(c > 64).float().mean().item():.4f}  <30: {(c < 30).float().mean().item():.4f}')
