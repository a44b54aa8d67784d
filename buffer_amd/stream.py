"""BASELINE config #3 on synthetic data: a stream of fragment pairs from RAW clouds through the test-split
pre-processing (ThreeDMatch/dataset.py:91-153 on device: two voxel levels, shuffles, 30-NN normals) and the batched
registration path, evaluated with the 3DMatch protocol (ThreeDMatch/test.py:114-173,263-308: DGR criterion per pair,
Registration Recall = mean over 8 scenes of the per-scene recall on non-consecutive fragment pairs through the
gt.info RMSE proxy).  The real test set (1623 pairs over 8 scenes) is not available offline: pair k of the stream goes to
scene k mod 8 as fragments (3q, 3q+2) with q = k div 8, and its gt.info matrix comes from the shared slab of the room."""
import time

import numpy as np
import torch

from . import evaluate, preprocess, synth
from .threedmatch import upload

OVERLAPS = (0.75, 0.6, 0.45, 0.3)            # a quarter of the stream at 0.3 (3DLoMatch territory)
N_SCENES = 8


def generate(n_pairs, device, seed0=20000, overlaps=OVERLAPS, n_raw=250_000):
    """-> list of make_raw_pair_device dicts, device resident (the untimed part: 'files already read')."""
    return [synth.make_raw_pair_device(seed0 + k, overlaps[k % len(overlaps)], device, n_raw=n_raw) for k in range(n_pairs)]


def prepare(raw, cfg, index):
    """one raw pair -> the reference's sample dict (device tensors), seeds as ThreeDMatchTestSet.item"""
    out = {'relt_pose': raw['relt_pose']}
    for j, side in enumerate(('src', 'tgt')):
        it = preprocess.prepare_fragment(raw[f'{side}_raw'], cfg.downsample, cfg.voxel_size_0, cfg.max_num_pts, seed=2 * index + j)
        out[f'{side}_fds_pts'], out[f'{side}_sds_pts'] = it['fds_pts'], it['sds_pts']
    return out


def prepare_batch(raws, cfg, indices):
    """prepare() for several raw pairs with the normals of all 2B fragments estimated in one stacked pass (same result)"""
    frs = preprocess.prepare_fragments([r[f'{side}_raw'] for r in raws for side in ('src', 'tgt')], cfg.downsample,
                                       cfg.voxel_size_0, cfg.max_num_pts, seeds=[2 * i + j for i in indices for j in range(2)])
    return [{'relt_pose': r['relt_pose'], 'src_fds_pts': frs[2 * k]['fds_pts'], 'src_sds_pts': frs[2 * k]['sds_pts'],
             'tgt_fds_pts': frs[2 * k + 1]['fds_pts'], 'tgt_sds_pts': frs[2 * k + 1]['sds_pts']} for k, r in enumerate(raws)]


def run(pipe, raws, batch=32, first_index=0):
    """Timed part: pre-processing + registration of every pair, `batch` pairs per set of stacked launches, batches
    software-pipelined (BufferPipeline.register_batches: pre-processing and keypoint stage of batch i+1 on the side stream
    beside the CNN kernels of batch i).  -> (poses f32[n,4,4] device, seconds)."""
    dev = pipe.device
    chunks = [list(range(lo, min(lo + batch, len(raws)))) for lo in range(0, len(raws), batch)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    makers = [(lambda ids=ids: [upload(s) for s in prepare_batch([raws[i] for i in ids], pipe.cfg, [first_index + i for i in ids])])
              for ids in chunks]
    out = pipe.register_batches(makers, seeds=[[first_index + i for i in ids] for ids in chunks])
    poses = [p for ps in out for p in ps]
    res = torch.stack(poses) if poses else torch.zeros((0, 4, 4), device=dev)
    torch.cuda.synchronize(dev)
    return res, time.perf_counter() - t0


def evaluate_stream(raws, poses, first_index=0):
    """3DMatch protocol over the synthetic stream -> dict(dgr_recall, registration_recall, per_scene, te, re, per-overlap)."""
    poses = np.asarray(poses, np.float64)
    n = len(raws)
    stats = [evaluate.dgr_success(poses[k], raws[k]['relt_pose']) for k in range(n)]
    scenes = [dict(pairs=[], gt=[], info=[], est=[]) for _ in range(N_SCENES)]
    for k in range(n):
        g = first_index + k
        sc, q = scenes[g % N_SCENES], g // N_SCENES
        # one leading consecutive pair per scene so that list index 0 is never a counted pair (test.py:119-123 skips it)
        if not sc['pairs']:
            sc['pairs'].append((0, 1)); sc['gt'].append(np.eye(4)); sc['info'].append(np.eye(6)); sc['est'].append(np.eye(4))
        sc['pairs'].append((3 * q + 2, 3 * q + 4))
        sc['gt'].append(np.linalg.inv(raws[k]['relt_pose']))              # gt.log: fragment j -> fragment i
        sc['info'].append(np.asarray(raws[k]['info'], np.float64) if 'info' in raws[k]
                          else synth.information_matrix(raws[k]['overlap_pts'].cpu().numpy()))
        sc['est'].append(np.linalg.inv(poses[k]))                          # the .log holds the inverse estimate (test.py:255)
    per_scene = []
    for sc in scenes:
        if len(sc['pairs']) <= 1:
            continue
        pr = np.array(sc['pairs'])
        nfrag = int(pr.max()) + 1
        _, recall, _, _ = evaluate.evaluate_registration(nfrag, np.array(sc['est']), pr, pr, np.array(sc['gt']), np.array(sc['info']))
        per_scene.append(float(recall))
    st = np.array([[float(a), b, c] for a, b, c in stats]).reshape(-1, 3)
    good = st[:, 0] == 1
    return dict(pairs=n, dgr_recall=float(good.mean()) if n else 0.0, registration_recall=float(np.mean(per_scene)) if per_scene else 0.0,
                per_scene=per_scene, te=float(st[good, 1].mean()) if good.any() else float('nan'),
                re=float(st[good, 2].mean()) if good.any() else float('nan'))
