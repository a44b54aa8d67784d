"""Development probe: time of the point learner (pyramid given) with the hoisted VN gather against the direct one
(BUF_VN_GATHER_DIRECT=1).   python tools/vn_probe.py [pairs]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import synth, pyramid, _lib
from buffer_amd.config import THREEDMATCH
from buffer_amd.pipeline import BufferPipeline
import ctypes as C
dev = torch.device('cuda:0')
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pipe = BufferPipeline(THREEDMATCH, dev)
samples = [synth.make_pair(3000 + i) for i in range(4)]
pipe.calibrate([samples[0]])
inps = [pipe.upload(samples[i % 4]) for i in range(npairs)]
lens = np.concatenate([np.asarray(i['lengths'], np.int32) for i in inps])
pts = torch.cat([i['points'] for i in inps]); feats = torch.cat([i['features'] for i in inps])
pyr = pyramid.build_pyramid(pts, lens, pipe.limits, THREEDMATCH)
seg = lens.reshape(npairs, 2).sum(1).astype(np.int32)
L = _lib.lib()
for _ in range(2):
    pipe.point.efcnn(pyr, feats, seg)
torch.cuda.synchronize()
L.buf_timing_enable(1)
t = time.perf_counter()
for _ in range(5):
    pipe.point.efcnn(pyr, feats, seg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
L.buf_timing_enable(0)
ms, work = C.c_double(0), C.c_double(0)
n = L.buf_timing_collect_kernel(7, C.byref(ms), C.byref(work))
print(f'efcnn {npairs} pairs: {dt*1e3:.2f} ms per call; VN gather blocks: {ms.value/5:.3f} ms per call ({n} launches, {work.value/ms.value/1e6:.1f} GB/s algorithmic)')
