#!/usr/bin/env python3
"""Development aid (round 5): the keypoint-stage VN kernels as VICTIMS beside one chip-filling kernel on another stream.
   python tools/race_probe3.py <aggressor: h3|h3head|wg|cost_h3|cost|nn1|none> [runs] [priority of the victim stream]"""
import os
import sys
from dataclasses import replace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from buffer_amd import ops, pyramid, registration, synth  # noqa: E402
from buffer_amd.config import THREEDMATCH  # noqa: E402
from buffer_amd.patch_embedder import PatchEmbedder  # noqa: E402
from buffer_amd.pipeline import BufferPipeline  # noqa: E402
from buffer_amd.weights import load_weights  # noqa: E402

agg = sys.argv[1] if len(sys.argv) > 1 else 'h3'
R = int(sys.argv[2]) if len(sys.argv) > 2 else 200
prio = int(sys.argv[3]) if len(sys.argv) > 3 else -1
dev = torch.device('cuda:0')
W = load_weights('3dmatch')
g = torch.Generator(device='cpu').manual_seed(0)
cfg = replace(THREEDMATCH, num_keypts=1500)
pipe = BufferPipeline(cfg, dev)
pipe.calibrate([synth.make_pair(1000)])
inps = [pipe.upload(synth.make_pair(2000 + i)) for i in range(4)]
lens = np.concatenate([np.asarray(i['lengths'], np.int32) for i in inps])
pts = torch.cat([i['points'] for i in inps])
feats = torch.cat([i['features'] for i in inps]).contiguous()
pyr = pyramid.build_pyramid(pts, lens, pipe.limits, cfg)
PL = pipe.point
P, N, PO = pyr['points'], pyr['neighbors'], pyr['pools']
x0 = ops.vn_gather_block(PL.b0, P[0], P[0], feats, N[0], 6, PL.scale)
victims = {
    'gather6 (block 0)': lambda: ops.vn_gather_block(PL.b0, P[0], P[0], feats, N[0], 6, PL.scale),
    'gather_pre (block 1)': lambda: ops.vn_gather_block(PL.res[0]['conv'], P[1], P[0], x0, PO[0], 1, PL.scale),
    'gather_max': lambda: ops.gather_max(x0, PO[0]),
    'pointwise': lambda: ops.vn_pointwise(PL.res[0]['short'], x0),
}
ref = {k: f().clone() for k, f in victims.items()}
torch.cuda.synchronize()
n = 12000
x = torch.relu(torch.randn((n, 48, 140), generator=g)).to(dev)
equi = torch.nn.functional.normalize(torch.randn((n, 32, 7, 20), generator=g), dim=1).to(dev)
srow = torch.randint(0, n, (n,), generator=g).to(dev)
trow = torch.randint(0, n, (n,), generator=g).to(dev)
desc = torch.nn.functional.normalize(torch.randn((2, n, 32), generator=g), dim=2).to(dev)
pe_f = PatchEmbedder(W, dev, THREEDMATCH)
pe_s = PatchEmbedder(W, dev, replace(THREEDMATCH, cnn_arith='split'))
cv_f = registration.CostVolume(W, dev, 20, 'f32')
cv_s = registration.CostVolume(W, dev, 20, 'split')
aggressors = {
    'h3': lambda: pe_s.fused(x), 'h3head': lambda: pe_s.fused.with_head(x, pe_s.fused_head), 'wg': lambda: pe_f.fused(x),
    'cost_h3': lambda: cv_s.fused.gathered(equi, srow, trow), 'cost': lambda: cv_f.fused.gathered(equi, srow, trow),
    'nn1': lambda: ops.knn(desc[:1], desc[1:], 1), 'none': lambda: None,
}
side = torch.cuda.Stream(device=dev, priority=prio)
main = torch.cuda.current_stream(dev)
bad = {k: 0 for k in victims}
shown = 0
for r in range(R):
    aggressors[agg]()                                     # ~10-30 ms of chip-filling work on the current stream
    outs = {}
    with torch.cuda.stream(side):
        for _ in range(3):
            for k, f in victims.items():
                outs.setdefault(k, []).append(f())
    torch.cuda.synchronize()
    for k, lst in outs.items():
        for o in lst:
            if not torch.equal(o, ref[k]):
                bad[k] += 1
                if shown < 4:
                    shown += 1
                    d = (o - ref[k]).abs()
                    rows = torch.nonzero(d.amax(1) > 0).flatten()
                    cols = torch.nonzero(d[rows[0]] > 0).flatten().tolist()
                    print(f'   {k} run {r}: rows {rows[:12].tolist()} ({rows.numel()}), elements {int((d > 0).sum())}, max |d| {float(d.max()):.3e}; columns of the first row: {cols[:40]}')
print(f'aggressor {agg}, victim stream priority {prio}: mismatching victim launches of {3 * R}:', bad)
