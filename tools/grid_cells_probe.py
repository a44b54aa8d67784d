#!/usr/bin/env python3
"""Development aid (round 5): keypoint-stage time of a 32-pair step against the size of the dense cell tables."""
import os, sys, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dataclasses import replace
    from buffer_amd import synth, ops, pyramid
    from buffer_amd.config import THREEDMATCH, KITTI
    from buffer_amd.pipeline import BufferPipeline
    dev = torch.device('cuda:0')
    kitti = sys.argv[2] == 'kitti'
    cfg = replace(KITTI if kitti else THREEDMATCH, num_keypts=1500 if kitti else 5000)
    pipe = BufferPipeline(cfg, dev)
    mk = synth.make_kitti_pair if kitti else synth.make_pair
    pipe.calibrate([mk(1000)])
    nb = 8 if kitti else 16
    inps = [pipe.upload(mk(2000 + i)) for i in range(nb)]
    def med(fn, n=7):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))
    lens = np.concatenate([np.asarray(i['lengths'], np.int32) for i in inps]); pts = torch.cat([i['points'] for i in inps])
    pyr = med(lambda: pyramid.build_pyramid(pts, lens, pipe.limits, cfg))
    kp = med(lambda: pipe._keypoints(inps, list(range(nb)), None))
    st = pipe._keypoints(inps, list(range(nb)), None)
    raws = [r for i in inps for r in (i['src_raw'], i['tgt_raw'])]
    sup, sup_len = ops.permute_clouds(raws, [ops.perm_key(b, j) for b in range(nb) for j in range(2)])
    sel = med(lambda: ops.select_patches_batched(sup, sup_len, st['kp'], cfg.num_keypts, cfg.des_r, cfg.num_points_per_patch))
    print(f'{sys.argv[2]} cells/point {os.environ.get("BUF_GRID_CELLS_PER_POINT", "16 (default)")}, vox {os.environ.get("BUF_VOX_CELLS_PER_POINT", "4 (default)")}: '
          f'pyramid {pyr:.3f} ms, keypoint stage {kp:.3f} ms, select_patches {sel:.3f} ms  ({nb} pairs)')
else:
    for shape in ('3dmatch', 'kitti'):
        for f, v in (('', '64'), ('', ''), ('8', ''), ('4', ''), ('2', '')):
            env = dict(os.environ)
            if f: env['BUF_GRID_CELLS_PER_POINT'] = f
            if v: env['BUF_VOX_CELLS_PER_POINT'] = v
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child', shape], env=env)
