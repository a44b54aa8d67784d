#!/usr/bin/env python3
"""Development aid (round 5): is the path bit-reproducible run to run?  One pair through register(detail=True) N times and a batch
through register_batch N times; every stage's tensors compared bitwise with the first run."""
import os
import sys
from dataclasses import replace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from buffer_amd import synth  # noqa: E402
from buffer_amd.config import THREEDMATCH  # noqa: E402
from buffer_amd.pipeline import BufferPipeline  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
samples = [synth.make_pair(2000 + i) for i in range(4)]


def flat(d, pre=''):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[pre + k] = v.detach().cpu()
        elif isinstance(v, dict):
            out.update(flat(v, pre + k + '.'))
        elif isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                if isinstance(x, torch.Tensor):
                    out[f'{pre}{k}[{i}]'] = x.detach().cpu()
                elif isinstance(x, dict):
                    out.update(flat(x, f'{pre}{k}[{i}].'))
    return out


for arith in ('f32', 'split'):
    pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=P, cnn_arith=arith), dev)
    pipe.calibrate([synth.make_pair(1000)])
    inps = [pipe.upload(s) for s in samples]
    first = None
    for r in range(N):
        pose, d = pipe.register(inps[0], seed=0, detail=True)
        cur = flat(d)
        cur['pose'] = pose.cpu()
        if first is None:
            first = cur
            continue
        bad = [k for k in first if k in cur and (first[k].shape != cur[k].shape or not torch.equal(first[k], cur[k]))]
        print(f'{arith} single run {r}: differing tensors: {bad if bad else "none"}')
        for k in bad[:6]:
            if first[k].shape == cur[k].shape and first[k].dtype.is_floating_point:
                print('    ', k, 'max |d|', float((first[k] - cur[k]).abs().max()))
    firstb = None
    for r in range(N):
        poses = torch.stack(pipe.register_batch(inps, seeds=[0, 1, 2, 3])).cpu()
        if firstb is None:
            firstb = poses
            continue
        print(f'{arith} batch run {r}: poses equal: {torch.equal(firstb, poses)}  max |d| {float((firstb - poses).abs().max()):.3e}')

# the two-stream software pipeline of bench.py (register_batches): the same three steps of 4 pairs, N times, both arithmetics
for arith in ('f32', 'split'):
    for kp in (600, P):
        pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=kp, cnn_arith=arith), dev)
        pipe.calibrate([synth.make_pair(1000)])
        inps = [pipe.upload(s) for s in samples]
        ref = torch.stack(pipe.register_batch(inps, seeds=[0, 1, 2, 3])).cpu()
        for r in range(3 * N):
            out = pipe.register_batches([inps, inps, inps], seeds=[[0, 1, 2, 3]] * 3)
            got = torch.stack([torch.stack(o) for o in out]).cpu()
            eq = [torch.equal(got[i], ref) for i in range(3)]
            if not all(eq):
                print(f'{arith} keypts {kp} pipelined run {r}: steps equal to the un-pipelined batch: {eq}  max |d| {float((got - ref[None]).abs().max()):.3e}')
        print(f'{arith} keypts {kp}: pipelined runs done')
