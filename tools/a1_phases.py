"""Development aid: phase times of k_vox_fused for the first element of a call (library built with -DVOXF_TIMING:
BUF_EXTRA_HIPCC_FLAGS=-DVOXF_TIMING python -c "from buffer_amd import build; build.build(force=True, out='build/ab/voxf_timing.so')";
BUF_LIB_PATH=build/ab/voxf_timing.so python tools/a1_phases.py)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import ops, synth, _lib
dev = torch.device('cuda:0')
s = synth.make_pair(3000)
P = torch.from_numpy(np.concatenate([s['src_sds_pts'][:, :3], s['tgt_sds_pts'][:, :3]]).astype(np.float32)).to(dev)
lens = np.array([len(s['src_sds_pts']), len(s['tgt_sds_pts'])], np.int32)
names = ['load', 'bbox', 'grid+zero', 'count', 'overflow?', 'scan', 'scatter', 'rank', 'write', 'head scan', 'emit']
L = _lib.lib()
L.buf_debug_voxf_ticks.argtypes = [C.c_void_p]
for dl in (0.07, 0.14):
    for _ in range(3):
        ops.grid_subsample_batch(P, lens, dl)
    torch.cuda.synchronize()
    t = (C.c_longlong * 16)()
    assert L.buf_debug_voxf_ticks(t) == 0
    t = np.array(t[:11], np.int64)
    d = np.diff(t) * 0.01
    print(f'dl {dl}, element of {lens[0]} points: total {0.01 * (t[10] - t[0]):.1f} us: ' + ', '.join(f'{n} {v:.1f}' for n, v in zip(names[1:], d)))
