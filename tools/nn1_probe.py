"""Development aid: the k = 1, d = 32 nearest-neighbour search alone (b = 32 elements of 5000 x 5000 unit descriptors)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buffer_amd import ops

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
b, n = 32, 5000
ref = torch.nn.functional.normalize(torch.randn((b, n, 32), generator=g, device=dev), dim=-1)
# queries = noisy copies of a third of the references (true matches) + unrelated ones
qry = torch.nn.functional.normalize(torch.randn((b, n, 32), generator=g, device=dev), dim=-1)
qry[:, ::3] = torch.nn.functional.normalize(ref[:, ::3] + 0.2 * torch.randn((b, (n + 2) // 3, 32), generator=g, device=dev), dim=-1)
ops.knn(ref, qry, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    d, i = ops.knn(ref, qry, 1)
torch.cuda.synchronize()
print(f'knn k=1: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call (b={b}, {n} x {n})')
