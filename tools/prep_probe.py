import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from buffer_amd import preprocess, synth, stream
from buffer_amd.config import THREEDMATCH as cfg
dev = torch.device('cuda:0')
raws = stream.generate(8, dev)
def T(f, reps=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, r
clouds = [r[k] for r in raws for k in ('src_raw', 'tgt_raw')]
t1, fds = T(lambda: [preprocess.voxel_down_sample(c, cfg.downsample) for c in clouds])
t2, sds = T(lambda: [preprocess.voxel_down_sample(f, cfg.voxel_size_0) for f in fds])
s32 = [s.float() for s in sds]
t3, nr = T(lambda: [preprocess.estimate_normals(s) for s in s32])
t4, _ = T(lambda: [stream.prepare(r, cfg, i) for i, r in enumerate(raws)])
print(f'16 fragments: voxel L1 {t1:.1f} ms, voxel L2 {t2:.1f} ms, normals {t3:.1f} ms, prepare (all) {t4:.1f} ms; raw {clouds[0].shape[0]} fds {fds[0].shape[0]} sds {sds[0].shape[0]}')
t5, _ = T(lambda: stream.prepare_batch(raws, cfg, list(range(len(raws)))))
print(f'prepare_batch (8 pairs, stacked normals): {t5:.1f} ms = {t5 / 8:.2f} ms per pair')
