"""Development aid: A1 (buf_grid_subsample_batch) and the A2 grid build per call at the bench's 32-pair step size."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import ops, synth
dev = torch.device('cuda:0')
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
samples = [synth.make_pair(3000 + i) for i in range(4)]
pts, lens = [], []
for i in range(npairs):
    s = samples[i % 4]
    pts += [s['src_sds_pts'][:, :3], s['tgt_sds_pts'][:, :3]]
    lens += [len(s['src_sds_pts']), len(s['tgt_sds_pts'])]
P = torch.from_numpy(np.concatenate(pts).astype(np.float32)).to(dev)
lens = np.array(lens, np.int32)


def timeit(f, reps=30):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e6)
    return float(np.median(ts))


for form in ('1', '0'):                # one workgroup per element, table in LDS | global-table counting sort
    os.environ['BUF_VOX_FUSED'] = form
    for dl in (0.07, 0.14):
        sub, sl = ops.grid_subsample_batch(P, lens, dl)
        print(f'A1 grid_subsample_batch (BUF_VOX_FUSED={form}) {npairs} pairs, {P.shape[0]} -> {sub.shape[0]} rows, dl {dl}: {timeit(lambda: ops.grid_subsample_batch(P, lens, dl)):.1f} us per call (host clock, incl. the row-count round trip)')
os.environ.pop('BUF_VOX_FUSED')
print(f'A2 grid build r=0.07: {timeit(lambda: ops.CellGrid(P, lens, 0.07)):.1f} us')
g = ops.CellGrid(P, lens, 0.07)
print(f'A2 self query K=17 in grid order: {timeit(lambda: g.query(P, lens, 17, q_order=g.order)):.1f} us')
