"""TEST INFRASTRUCTURE: runs the CPU oracle (oracle/pipeline_ref.register_pair, the restated reference) on the synthetic
pairs of tests/eval_recall.py in a process of its own -- tests/test_rr_parity_gpu.py starts a few of these beside the GPU
run and compares pair by pair.  No GPU is touched here.

    python tests/oracle_worker.py --pairs 0,4 --keypts 1500 --threads 32 --out /tmp/oracle_0.npz
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', required=True)
    ap.add_argument('--keypts', type=int, default=1500)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--out', required=True)
    ap.add_argument('--dataset', choices=['3dmatch', 'kitti'], default='3dmatch',
                    help="kitti: synth.make_kitti_pair(2000 + i) (the pairs of bench.py --workload kitti), KITTI constants and weights")
    ap.add_argument('--limits', default=None, help='neighbour limits a,b,c (default: calibrated here on pair 0)')
    a = ap.parse_args()
    if a.threads:
        os.environ['OMP_NUM_THREADS'] = str(a.threads)
    os.environ['CUDA_VISIBLE_DEVICES'] = ''
    os.environ['HIP_VISIBLE_DEVICES'] = ''
    from dataclasses import replace
    import numpy as np
    import torch
    if a.threads:
        torch.set_num_threads(a.threads)
    from buffer_amd import synth
    from buffer_amd.config import KITTI, THREEDMATCH
    from buffer_amd.weights import load_weights
    from oracle import cpu, pipeline_ref, torch_ref
    from tests.eval_recall import make
    if a.dataset == 'kitti':
        make = lambda i: synth.make_kitti_pair(2000 + i)      # noqa: E731
    cfg = replace(KITTI if a.dataset == 'kitti' else THREEDMATCH, num_keypts=a.keypts)
    cpu.build(ref=False)
    if a.limits:
        limits = [int(x) for x in a.limits.split(',')]
    else:
        limits = [int(x) for x in torch_ref.calibrate_limits([make(0)], cfg.voxel_size_0, cfg.conv_radius)]
    W = {k: torch.from_numpy(v) for k, v in load_weights(cfg.weights).items()}
    out = dict(limits=np.array(limits))
    for i in [int(x) for x in a.pairs.split(',')]:
        s = make(i)
        rng = np.random.default_rng(i)
        perms = [rng.permutation(len(s['src_fds_pts'])), rng.permutation(len(s['tgt_fds_pts']))]
        pose, d = pipeline_ref.register_pair(s, W, limits, cfg, i, perms, use_ref=cpu.have_ref())
        out[f'pose_{i}'] = np.asarray(pose, np.float64)
        out[f'kp0_{i}'], out[f'kp1_{i}'] = d['kpts'][0].numpy(), d['kpts'][1].numpy()
        out[f'smids_{i}'], out[f'tmids_{i}'] = np.asarray(d['s_mids']), np.asarray(d['t_mids'])
        out[f'inliers_{i}'] = np.asarray(d['inlier_ind'])
        print('oracle pair', i, 'done', flush=True)
    np.savez(a.out, **out)


if __name__ == '__main__':
    main()
