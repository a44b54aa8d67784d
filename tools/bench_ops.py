"""Micro-benchmarks of single operators on the GPU (development aid)."""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import ops, synth

dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'fps'
if which == 'fps':
    for n in (8000, 11000, 13300, 15000):
        rng = np.random.default_rng(0)
        pts = torch.from_numpy((rng.random((n, 3)) * 2 + 0.5).astype(np.float32)).to(dev)
        for m in (1500, 5000):
            ops.furthest_point_sample_ragged(pts, [n], m)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                ops.furthest_point_sample_ragged(pts, [n], m)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 3
            print(f'fps n={n} m={m}: {dt*1e3:.2f} ms  {dt/m*1e6:.2f} us/round')
    # fragment-shaped clouds (surfaces): what the pruned form is built for
    for seed in (3000, 3001):
        d = synth.make_pair(seed)
        for key in ('src_sds_pts', 'tgt_sds_pts'):
            c = torch.from_numpy(d[key][:, :3].astype(np.float32)).to(dev)
            n = c.shape[0]
            for m in (1500, 5000):
                ref = ops.furthest_point_sample_ragged(c, [n], m)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(3):
                    ops.furthest_point_sample_ragged(c, [n], m)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) / 3
                print(f'fps fragment {seed} {key} n={n} m={m}: {dt*1e3:.2f} ms  {dt/m*1e6:.2f} us/round  checksum {int(ref.long().sum())}')
    # a ragged batch the way the step launches it (one workgroup per cloud, the kernel tier picked by the largest)
    rng = np.random.default_rng(1)
    lens = [int(v) for v in rng.integers(6500, 13300, 32)]
    pts = torch.from_numpy((rng.random((sum(lens), 3)) * 2 + 0.5).astype(np.float32)).to(dev)
    for m in (1500, 5000):
        ref = ops.furthest_point_sample_ragged(pts, lens, m)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            ops.furthest_point_sample_ragged(pts, lens, m)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        print(f'fps ragged 32 clouds {min(lens)}..{max(lens)} m={m}: {dt*1e3:.2f} ms  checksum {int(ref.long().sum())}')
if which == 'cyl':
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    from buffer_amd.weights import load_weights
    pe = PatchEmbedder(load_weights('3dmatch'), dev, THREEDMATCH)
    x = torch.rand((5000, 16, 420), device=dev)
    for _ in range(2):
        pe.fused(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        pe.fused(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f'cyl_net_wg 5000 patches: {dt*1e3:.2f} ms  {5000*0.1187/dt/1e3:.1f} dense-equivalent TFLOP/s  (matrix pipe {5000*0.1187*0.5587/dt/1e3/157.3*100:.0f}% busy)')
if which == 'radius':
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    samples = [synth.make_pair(3000 + i) for i in range(min(npairs, int(os.environ.get('BENCH_OPS_SAMPLES', 4))))]      # distinct pairs, repeated
    pts, lens = [], []
    for i in range(npairs):
        s = samples[i % len(samples)]
        pts += [s['src_sds_pts'][:, :3], s['tgt_sds_pts'][:, :3]]
        lens += [len(s['src_sds_pts']), len(s['tgt_sds_pts'])]
    P = torch.from_numpy(np.concatenate(pts).astype(np.float32)).to(dev)
    lens = np.array(lens, np.int32)
    K = 17
    grid = ops.CellGrid(P, lens, 0.07)
    order = grid.order
    out = grid.query(P, lens, K, q_order=order)
    torch.cuda.synchronize()
    for use_order in (True, False):
        t = time.perf_counter()
        for _ in range(iters):
            grid.query(P, lens, K, q_order=order if use_order else None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / iters
        nq = P.shape[0]
        by = 12 * nq * 2 + 4 * nq * K
        print(f'radius self-query pairs={npairs} nq={nq} K={K} order={use_order}: {dt*1e6:.1f} us/launch (incl. host), '
              f'{by/dt/1e9:.1f} GB/s algorithmic')
    t = time.perf_counter()
    for _ in range(iters):
        g2 = ops.CellGrid(P, lens, 0.07)
    torch.cuda.synchronize()
    print(f'grid build: {(time.perf_counter()-t)/iters*1e6:.1f} us')
if which == 'vox':
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    from buffer_amd.weights import load_weights
    cfg = THREEDMATCH
    pe = PatchEmbedder(load_weights('3dmatch'), dev, cfg)
    s = synth.make_pair(3)
    raw = torch.from_numpy(s['src_fds_pts'].astype(np.float32)).to(dev)
    kp = raw[torch.randperm(raw.shape[0], device=dev)[:5000]].contiguous()
    ax = torch.nn.functional.normalize(torch.randn(5000, 3, device=dev), dim=1)
    patches = ops.select_patches(raw, kp, cfg.des_r, 512)
    def run():
        return ops.patch_voxelize(patches, ax, cfg.des_r, pe.centres, pe.azi_cs, cfg.delta / cfg.rad_n, cfg.voxel_sample,
                                  pe.mlp_w, pe.mlp_b, pe.mlp_s, pe.mlp_t, cfg.azi_n, False)
    run(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    print(f'voxelize 5000 patches: {(time.perf_counter()-t)/10*1e3:.3f} ms')
    t = time.perf_counter()
    for _ in range(10):
        ops.select_patches(raw, kp, cfg.des_r, 512)
    torch.cuda.synchronize()
    print(f'select_patches 5000 kpts x {raw.shape[0]} pts: {(time.perf_counter()-t)/10*1e3:.3f} ms')
if which == 'cost':
    from buffer_amd import registration
    from buffer_amd.weights import load_weights
    cv = registration.CostVolume(load_weights('3dmatch'), dev)
    g = torch.Generator(device='cpu').manual_seed(1)
    a = torch.nn.functional.normalize(torch.rand((2500, 32, 5, 20), generator=g), dim=1).to(dev)
    b = torch.nn.functional.normalize(torch.rand((2500, 32, 5, 20), generator=g), dim=1).to(dev)
    cv(a, b); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        cv(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f'cost_net 2500 matches: {dt*1e3:.2f} ms  {2500*0.05191/dt/1e3:.1f} TFLOP/s executed ({2500*0.160/dt/1e3:.1f} dense-equivalent)')
if which == 'prep':
    from buffer_amd import preprocess
    sample = synth.make_pair(seed=0)
    raw = torch.from_numpy(np.concatenate([sample['src_fds_pts']] * 12) + np.random.default_rng(0).normal(0, 0.004, (12 * sample['src_fds_pts'].shape[0], 3)).astype(np.float32)).to(dev)
    sds = torch.from_numpy(sample['src_sds_pts'][:, :3].copy()).to(dev)

    def timeit(f, reps=5):
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps
    dt = timeit(lambda: preprocess.voxel_down_sample(raw, 0.02))
    m = preprocess.voxel_down_sample(raw, 0.02).shape[0]
    print(f'voxel_down_sample {raw.shape[0]} -> {m} pts: {dt*1e3:.2f} ms')
    dt = timeit(lambda: preprocess.estimate_normals(sds))
    print(f'estimate_normals {sds.shape[0]} pts (knn 30): {dt*1e3:.2f} ms')
    dt = timeit(lambda: preprocess.prepare_fragment(raw, 0.02, 0.035))
    print(f'prepare_fragment (2 voxel levels + shuffles + normals): {dt*1e3:.2f} ms')
