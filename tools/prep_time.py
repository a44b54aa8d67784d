import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from buffer_amd import preprocess, stream, ops
from buffer_amd.config import THREEDMATCH as cfg
dev = torch.device('cuda:0')
raws = stream.generate(16, dev)
def T(f, reps=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, r
rl = [r[f'{s}_raw'] for r in raws for s in ('src', 'tgt')]
t0, cat = T(lambda: torch.cat(rl))
t1, (fds, fl) = T(lambda: preprocess.voxel_down_sample_batch(cat, [int(r.shape[0]) for r in rl], cfg.downsample))
t2, (sds, sl) = T(lambda: preprocess.voxel_down_sample_batch(fds, fl, cfg.voxel_size_0))
s32 = sds.float()
t3, _ = T(lambda: preprocess.estimate_normals(s32, lengths=sl))
t4, frs = T(lambda: preprocess.prepare_fragments(rl, cfg.downsample, cfg.voxel_size_0, cfg.max_num_pts, list(range(32))))
t5, _ = T(lambda: stream.prepare_batch(raws, cfg, list(range(16))))
t6, _ = T(lambda: [stream.upload(s) for s in stream.prepare_batch(raws, cfg, list(range(16)))])
print(f'alone on the GPU, 16 pairs: cat {t0:.1f}  voxel L1 {t1:.1f}  L2 {t2:.1f}  normals {t3:.1f}  prepare_fragments {t4:.1f}  prepare_batch {t5:.1f}  + upload {t6:.1f} ms')
