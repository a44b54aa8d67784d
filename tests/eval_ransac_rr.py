"""A15 evidence (run on the GPU box; writes profiles/<round>_ransac_rr.json): the product's deterministic
4096-hypothesis GPU RANSAC (csrc/registration.hip k_ransac) against the restated open3d 0.13 RANSAC the reference
calls (models/BUFFER.py:314-326, oracle/ransac_o3d.py: 50 000 iterations, confidence 0.999, edge 0.8 / distance 0.10
checkers), on the SAME correspondences of every pair, both followed by the same post-refinement, scored with the 3DMatch
protocol (DGR recall and Registration Recall through evaluate.evaluate_registration) over a synthetic stream with a
quarter of the pairs at 0.3 overlap; several seeds on both sides give the spread.

    python tests/eval_ransac_rr.py --pairs 240 --seeds 5 --out profiles/r02_ransac_rr.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from buffer_amd import ops, stream  # noqa: E402
from buffer_amd.config import THREEDMATCH  # noqa: E402
from buffer_amd.pipeline import BufferPipeline  # noqa: E402
from buffer_amd.threedmatch import upload  # noqa: E402
from oracle import ransac_o3d  # noqa: E402  (the checker; test infrastructure)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=240)
    ap.add_argument('--seeds', type=int, default=5)
    ap.add_argument('--overlaps', default='0.75,0.6,0.45,0.3')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'ransac_rr.json'))
    a = ap.parse_args()
    overlaps = tuple(float(x) for x in a.overlaps.split(','))
    dev = torch.device('cuda:0')
    cfg = THREEDMATCH
    pipe = BufferPipeline(cfg, dev)
    raws = stream.generate(a.pairs, dev, seed0=30000, overlaps=overlaps)
    first = stream.prepare(raws[0], cfg, 0)
    pipe.calibrate([{k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in first.items()}])
    variants = {f'gpu4096_seed{s}': [] for s in range(a.seeds)}
    variants.update({f'open3d_seed{s}': [] for s in range(a.seeds)})
    variants['product'] = []
    meta = []
    t0 = time.time()
    for k in range(a.pairs):
        inp = upload(stream.prepare(raws[k], cfg, k))
        pose, d = pipe.register(inp, seed=k, detail=True)
        variants['product'].append(pose.cpu().numpy())
        if 's_mids' not in d or 'inlier_mask' not in d:                 # no keypoints / < 3 matches: identity everywhere
            for name in variants:
                if name != 'product':
                    variants[name].append(np.eye(4))
            meta.append(dict(pair=k, overlap=overlaps[k % len(overlaps)], matches=0, inliers=0))
            continue
        ss = d['kpts'][0][d['s_mids']].contiguous()
        tt = d['kpts'][1][d['t_mids']].contiguous()
        mask = d['inlier_mask']
        ind = torch.nonzero(mask).flatten().cpu().numpy()
        ss_h, tt_h = ss.cpu().numpy().astype(np.float64), tt.cpu().numpy().astype(np.float64)
        its = []
        for s in range(a.seeds):
            T, _ = ops.ransac_kabsch_masked(ss, tt, mask, cfg.ransac_hypotheses, 1000 * s + k, cfg.dist_th, cfg.similar_th)
            T, _ = ops.post_refine(T, ss, tt, cfg.refine_threshold, 20)
            variants[f'gpu4096_seed{s}'].append(T.cpu().numpy())
            r = ransac_o3d.ransac_correspondence(ss_h, tt_h, np.stack([ind, ind], 1), cfg.dist_th, cfg.similar_th, cfg.dist_th,
                                                 cfg.iter_n, cfg.confidence, seed=1000 * s + k)
            T, _ = ops.post_refine(torch.from_numpy(r['T'].astype(np.float32)).to(dev), ss, tt, cfg.refine_threshold, 20)
            variants[f'open3d_seed{s}'].append(T.cpu().numpy())
            its.append(r['iterations'])
        meta.append(dict(pair=k, overlap=overlaps[k % len(overlaps)], matches=int(ss.shape[0]), inliers=int(len(ind)),
                         open3d_iterations=its))
        if k % 20 == 0:
            print(k, meta[-1], flush=True)
    res = {name: stream.evaluate_stream(raws, np.stack(p)) for name, p in variants.items()}
    gpu = [res[f'gpu4096_seed{s}'] for s in range(a.seeds)]
    o3d = [res[f'open3d_seed{s}'] for s in range(a.seeds)]
    summ = dict(
        pairs=a.pairs, overlaps=overlaps, seeds=a.seeds, seconds=time.time() - t0,
        product=dict(dgr_recall=res['product']['dgr_recall'], registration_recall=res['product']['registration_recall']),
        gpu4096=dict(dgr_recall=[r['dgr_recall'] for r in gpu], registration_recall=[r['registration_recall'] for r in gpu]),
        open3d_restated=dict(dgr_recall=[r['dgr_recall'] for r in o3d], registration_recall=[r['registration_recall'] for r in o3d]),
    )
    for key in ('dgr_recall', 'registration_recall'):
        g, o = np.array(summ['gpu4096'][key]), np.array(summ['open3d_restated'][key])
        summ[key + '_difference_points'] = dict(mean=float(100 * (g.mean() - o.mean())), gpu_spread=float(100 * (g.max() - g.min())),
                                                open3d_spread=float(100 * (o.max() - o.min())))
    # per-pair agreement of success flags between the two samplers (seed 0)
    ok = lambda name: np.array([np.linalg.norm(p[:3, 3] - raws[i]['relt_pose'][:3, 3]) < 0.3 for i, p in enumerate(variants[name])])
    summ['pairs_where_seed0_success_differs'] = int((ok('gpu4096_seed0') != ok('open3d_seed0')).sum())
    out = dict(summary=summ, results=res, pairs=meta)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, 'w'), indent=0)
    print(json.dumps(summ, indent=1))


if __name__ == '__main__':
    main()
