"""Development aid: where one un-batched pair at the reference's operating point (1500 keypoints) spends its 7 ms: stage times (host clock,
synchronised) and the GPU-busy share from a kernel trace (run under tools/prof.sh <name> stats for the trace)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from buffer_amd import synth
from buffer_amd.config import THREEDMATCH
from buffer_amd.pipeline import BufferPipeline
dev = torch.device('cuda:0')
arith = sys.argv[1] if len(sys.argv) > 1 else 'f32'
pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=1500, cnn_arith=arith), dev)
pipe.calibrate([synth.make_pair(1000)])
inp = pipe.upload(synth.make_pair(2000))


def med(fn, n=15):
    fn(); ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts)), r


t_all, _ = med(lambda: pipe.register_batch([inp], seeds=[0]))
t_kp, st = med(lambda: pipe._keypoints([inp], [0], None))
t_desc, st2 = med(lambda: pipe._describe(dict(st)))
t_match, _ = med(lambda: pipe._match(dict(st2)))
print(f'{arith}: register_batch {t_all:.2f} ms = keypoints {t_kp:.2f} + describe {t_desc:.2f} + match {t_match:.2f} (sum {t_kp + t_desc + t_match:.2f})')
from buffer_amd import ops, pyramid
t_pyr, pyr = med(lambda: pyramid.build_pyramid(inp['points'], inp['lengths'], pipe.limits, pipe.cfg))
t_pl, _ = med(lambda: pipe.point.detnet(pyr, *pipe.point.efcnn(pyr, inp['features'])[2:4]))
print(f'   keypoints: pyramid {t_pyr:.2f}, point learner (efcnn + detnet) {t_pl:.2f}, rest (threshold, FPS of 1500, gathers) {t_kp - t_pyr - t_pl:.2f}')
