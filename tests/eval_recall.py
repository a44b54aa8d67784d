"""Registration quality on a stream of synthetic 3DMatch-shape pairs of varying overlap (BASELINE config #3 stand-in:
the real test set is not available offline).  --backend gpu runs buffer_amd (HIP), --backend oracle the CPU
restatement of the reference (oracle/pipeline_ref.py); both use the same pairs, permutations and seeds, so the two
JSON files can be compared pair by pair (profiles/recall_*.json)."""
import argparse
import json
import os
import sys
import time
from dataclasses import replace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from buffer_amd import synth  # noqa: E402
from buffer_amd.config import THREEDMATCH  # noqa: E402
from buffer_amd.evaluate import dgr_success  # noqa: E402


def make(i):
    overlap = [0.75, 0.6, 0.45, 0.3][i % 4]
    return synth.make_pair(9000 + i, overlap=overlap)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', choices=['gpu', 'oracle'], required=True)
    ap.add_argument('--pairs', type=int, default=24)
    ap.add_argument('--keypts', type=int, default=1500)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    cfg = replace(THREEDMATCH, num_keypts=a.keypts)
    calib = make(0)
    rows = []
    if a.backend == 'gpu':
        from buffer_amd.pipeline import BufferPipeline
        dev = torch.device('cuda:0')
        pipe = BufferPipeline(cfg, dev)
        limits = pipe.calibrate([calib])
    else:
        from oracle import cpu, pipeline_ref, torch_ref
        from buffer_amd.weights import load_weights
        cpu.build(ref=True)
        limits = [int(x) for x in torch_ref.calibrate_limits([calib])]
        W = {k: torch.from_numpy(v) for k, v in load_weights(cfg.weights).items()}
    t0 = time.time()
    for i in range(a.pairs):
        s = make(i)
        rng = np.random.default_rng(i)
        perms = [rng.permutation(len(s['src_fds_pts'])), rng.permutation(len(s['tgt_fds_pts']))]
        if a.backend == 'gpu':
            pose = pipe.register(pipe.upload(s), seed=i, perms=[torch.from_numpy(p).to(dev) for p in perms]).cpu().numpy()
        else:
            pose, _ = pipeline_ref.register_pair(s, W, limits, cfg, i, perms, use_ref=cpu.have_ref())
        ok, rte, rre = dgr_success(pose, s['relt_pose'])
        rows.append(dict(pair=i, overlap=[0.75, 0.6, 0.45, 0.3][i % 4], ok=ok, rte=rte, rre=rre,
                         pose=np.asarray(pose, np.float64).tolist()))
        print(i, ok, round(rte, 4), round(rre, 3), flush=True)
    out = dict(backend=a.backend, keypts=a.keypts, limits=[int(x) for x in limits], pairs=rows,
               recall=float(np.mean([r['ok'] for r in rows])), seconds=time.time() - t0)
    path = a.out or os.path.join(ROOT, 'profiles', f'recall_{a.backend}.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, 'w'), indent=0)
    print('recall', out['recall'], 'in', round(out['seconds'], 1), 's ->', path)


if __name__ == '__main__':
    main()
