#!/usr/bin/env python3
"""Development aid: keypoint stage of one pair / of a 32-pair step with the fused score heads and with the layer-by-layer path."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from buffer_amd import synth
from buffer_amd.config import THREEDMATCH
from buffer_amd.pipeline import BufferPipeline
dev = torch.device('cuda:0')
pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=5000), dev)
pipe.calibrate([synth.make_pair(1000)])
inps = [pipe.upload(synth.make_pair(2000 + i)) for i in range(8)]
def med(fn, n=15):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
from buffer_amd import pyramid
lens1 = np.asarray(inps[0]['lengths'], np.int32)
pyr1 = pyramid.build_pyramid(inps[0]['points'], lens1, pipe.limits, pipe.cfg)
for fused in (True, False, True, False):
    for h in (pipe.point.eps_head, pipe.point.key_head):
        h.fused = fused
    one = med(lambda: pipe._keypoints([inps[0]], [0], None))
    pl = med(lambda: (lambda a: pipe.point.detnet(pyr1, a[2], a[3]))(pipe.point.efcnn(pyr1, inps[0]['features'])))
    step = med(lambda: pipe._keypoints([inps[k % 8] for k in range(32)], list(range(32)), None), 5)
    print(f'fused score heads {fused}: keypoint stage one pair {one:.3f} ms (point learner alone {pl:.3f} ms), 32-pair step {step:.2f} ms')
