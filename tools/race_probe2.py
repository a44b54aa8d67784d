#!/usr/bin/env python3
"""Development aid (round 5): which stage of the two-stream pipeline (register_batches, cnn_arith='split') differs run to run?"""
import os
import sys
from dataclasses import replace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from buffer_amd import synth  # noqa: E402
from buffer_amd.config import THREEDMATCH  # noqa: E402
from buffer_amd.pipeline import BufferPipeline  # noqa: E402

arith = sys.argv[1] if len(sys.argv) > 1 else 'split'
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kp = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
dev = torch.device('cuda:0')
samples = [synth.make_pair(2000 + i) for i in range(4)]
pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=kp, cnn_arith=arith), dev)
pipe.calibrate([synth.make_pair(1000)])
mix = os.environ.get('RACE_MIX', '')
if mix:                                                      # replace one of the two CNN stages by the other arithmetic's
    from buffer_amd import registration
    from buffer_amd.patch_embedder import PatchEmbedder
    other = 'f32' if arith == 'split' else 'split'
    if 'cost' in mix:
        pipe.inlier = registration.CostVolume(pipe.W, dev, pipe.cfg.azi_n, other)
    if 'desc' in mix:
        pipe.desc = PatchEmbedder(pipe.W, dev, replace(pipe.cfg, cnn_arith=other))
    print('mixed:', mix, '->', other)
inps = [pipe.upload(s) for s in samples]
log = []
orig_describe, orig_gathered, orig_kp = pipe._describe, pipe.inlier.gathered, pipe._keypoints


def spy_kp(*a, **k):
    st = orig_kp(*a, **k)
    log.append(('kp', st['kp'].clone()))
    log.append(('ka', st['ka'].clone()))
    return st


def spy_describe(st):
    st = orig_describe(st)
    log.append(('x', st['emb']['x'].clone()))
    log.append(('desc', st['emb']['desc'].clone()))
    log.append(('equi', st['emb']['equi'].clone()))
    log.append(('R', st['emb']['R'].clone()))
    log.append(('s_nn', st['s_nn'].clone()))
    log.append(('mutual', st['mutual'].clone()))
    return st


def spy_gathered(equi, a, b):
    out = orig_gathered(equi, a, b)
    log.append(('rows', torch.stack([a, b]).clone()))
    log.append(('ind', out.clone()))
    return out


from buffer_amd import pyramid as _pyr  # noqa: E402
orig_build, orig_efcnn, orig_det = _pyr.build_pyramid, pipe.point.efcnn, pipe.point.detnet


def spy_build(*a, **k):
    pyr = orig_build(*a, **k)
    for key in ('points', 'neighbors', 'pools', 'upsamples'):
        for l, t in enumerate(pyr[key]):
            if isinstance(t, torch.Tensor) and t.numel():
                log.append((f'pyr.{key}[{l}]', t.clone()))
    return pyr


from buffer_amd import ops as _ops  # noqa: E402
for _name in ('vn_gather_block', 'gather_max', 'vn_pointwise'):
    def _mk(name, fn):
        def spy(*a, **k):
            out = fn(*a, **k)
            if os.environ.get('RACE_OPS'):
                log.append((f'op.{name}', out.clone()))
            return out
        return spy
    setattr(_ops, _name, _mk(_name, getattr(_ops, _name)))


def spy_efcnn(*a, **k):
    out = orig_efcnn(*a, **k)
    log.append(('axis', out[0].clone())); log.append(('eps', out[1].clone())); log.append(('bottle', out[2].clone()))
    for i, sk in enumerate(out[3]):
        log.append((f'skip{i}', sk.clone()))
    return out


def spy_det(*a, **k):
    out = orig_det(*a, **k)
    log.append(('score', out.clone()))
    return out


_pyr.build_pyramid, pipe.point.efcnn, pipe.point.detnet = spy_build, spy_efcnn, spy_det
pipe._describe, pipe.inlier.gathered, pipe._keypoints = spy_describe, spy_gathered, spy_kp
ref = None
seen, nbad = set(), 0
for r in range(R):
    log.clear()
    out = pipe.register_batches([inps, inps, inps], seeds=[[0, 1, 2, 3]] * 3)
    torch.cuda.synchronize()
    cur = [(n, t.cpu()) for n, t in log] + [('pose', torch.stack([torch.stack(o) for o in out]).cpu())]
    if ref is None:
        ref = cur
        names = [n for n, _ in cur]
        print('stages logged per run:', len(names))
        continue
    bad = [(i, n) for i, ((n, a), (_, b)) in enumerate(zip(ref, cur)) if a.shape != b.shape or not torch.equal(a, b)]
    if bad and bad[0][1] not in seen:
        seen.add(bad[0][1])
        print(f'run {r}: differing (log position, stage):', bad[:10])
        i, n = bad[0]
        a, b = ref[i][1], cur[i][1]
        if a.shape == b.shape:
            d = (a.double() - b.double()).abs()
            rows = torch.nonzero(d.reshape(d.shape[0], -1).amax(1) > 0).flatten()
            print(f'    first: {n}: {rows.numel()} rows differ (first {rows[:10].tolist()}), max |d| {float(d.max()):.3e}, elements {int((d > 0).sum())}')
    nbad += bool(bad)
print(f'done: {nbad} of {R - 1} runs differ')
