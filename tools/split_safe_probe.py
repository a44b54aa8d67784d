"""What 'safe by construction' costs when nothing trips (csrc/split_safe.hip): the split-f16 CNN kernels with the flag memset and the
masked fp32 re-run launches (every workgroup exits at once) against the bare split kernels (BUF_SPLIT_UNSAFE=1), same inputs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from buffer_amd import registration
from buffer_amd.config import THREEDMATCH
from buffer_amd.patch_embedder import PatchEmbedder
from buffer_amd.weights import load_weights

dev = torch.device('cuda:0')
W = load_weights('3dmatch')
cfg = replace(THREEDMATCH, cnn_arith='split')


def build(unsafe):
    if unsafe:
        os.environ['BUF_SPLIT_UNSAFE'] = '1'
    else:
        os.environ.pop('BUF_SPLIT_UNSAFE', None)
    return PatchEmbedder(W, dev, cfg), registration.CostVolume(W, dev, arith='split')


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


pe_s, cv_s = build(False)
pe_u, cv_u = build(True)
assert pe_s.fused.safe and not pe_u.fused.safe and cv_s.fused.safe and not cv_u.fused.safe
g = torch.Generator(device='cpu').manual_seed(0)
for npatch in (3000, 40000, 320000):
    x = torch.rand((min(npatch, 40000), 16, 420), generator=g).to(dev)
    if npatch > x.shape[0]:
        x = x.repeat(npatch // x.shape[0], 1, 1)
    a = timeit(lambda: pe_s.fused.with_head(x, pe_s.fused_head), 3)
    b = timeit(lambda: pe_u.fused.with_head(x, pe_u.fused_head), 3)
    d1, e1 = pe_s.fused.with_head(x, pe_s.fused_head)
    d2, e2 = pe_u.fused.with_head(x, pe_u.fused_head)
    print(f'descriptor CNN + head, {npatch} patches: safe {a:.3f} ms, bare {b:.3f} ms ({(a / b - 1) * 100:+.2f} %), identical {torch.equal(d1, d2) and torch.equal(e1, e2)}, fallbacks {pe_s.fused.range_fallbacks()}')
    del x, d1, e1, d2, e2
for m in (600, 2500, 64000):
    sa = torch.nn.functional.normalize(torch.rand((min(m, 8000), 32, 5, 20), generator=g), dim=1).to(dev)
    sb = torch.nn.functional.normalize(torch.rand((min(m, 8000), 32, 5, 20), generator=g), dim=1).to(dev)
    if m > sa.shape[0]:
        sa, sb = sa.repeat(m // sa.shape[0], 1, 1, 1), sb.repeat(m // sb.shape[0], 1, 1, 1)
    a = timeit(lambda: cv_s(sa, sb), 3)
    b = timeit(lambda: cv_u(sa, sb), 3)
    print(f'cost net, {m} matches: safe {a:.3f} ms, bare {b:.3f} ms ({(a / b - 1) * 100:+.2f} %), identical {torch.equal(cv_s(sa, sb), cv_u(sa, sb))}, fallbacks {cv_s.fused.range_fallbacks()}')
