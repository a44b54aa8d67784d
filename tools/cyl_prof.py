"""Development aid: cycle accounting of one k_cyl_net wavefront.  Needs a library built with
BUF_EXTRA_HIPCC_FLAGS=-DCYL_PROF (python buffer_amd/build.py --force)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import _lib
from buffer_amd.config import THREEDMATCH
from buffer_amd.patch_embedder import PatchEmbedder
from buffer_amd.weights import load_weights

dev = torch.device('cuda:0')
pe = PatchEmbedder(load_weights('3dmatch'), dev, THREEDMATCH)
x = torch.rand((5000, 16, 420), device=dev)
pe.fused(x)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 8)()
raw.buf_debug_cyl_prof(buf)
v = list(buf)
names = ['tap loops', 'MFMAs issued', 'barrier before epilogue', 'epilogue', 'barrier after layer', 'kernel total', 'tap set-up']
for n, c in zip(names, v):
    print(f'{n:26s} {c:10d}  {100.0 * c / max(v[5], 1):5.1f}%' if n != 'MFMAs issued' else f'{n:26s} {c:10d}')
print('own MFMA cycles (32 per MFMA): %d = %.1f%% of the kernel' % (v[1] * 32, 100.0 * v[1] * 32 / max(v[5], 1)))
