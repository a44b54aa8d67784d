#!/usr/bin/env python3
"""profiles/instr.json: wave-level instruction counts per unit of work of the kernels whose roofline is priced on ISSUED instructions,
from one rocprofv3 --pmc "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" pass of the bench command (full-step launches only, BUF_NO_TRAFFIC=1).

    python tools/make_instr.py <counter_collection.csv> <bench --detail-json file> <out json>

bench.py prices k_patch_voxelize (vector-ALU issue bound) with `valu_per_patch` from this file; the library version it was measured
on is recorded, and tests/test_bench_contract_gpu.py fails when buf_version() has moved on (a kernel changed, the count is stale)."""
import collections
import csv
import ctypes
import hashlib
import json
import os
import sys

path, bench_file, out = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = json.load(open(bench_file))
cfg = bench['config']
patches = 2 * cfg['keypoints_per_fragment'] * cfg['pairs_per_step_per_gpu']
rows = collections.defaultdict(dict)                  # kernel -> dispatch -> {grid, counters}
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
    e = rows[k].setdefault(r['Dispatch_Id'], {'grid': int(r['Grid_Size'])})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
kernels = {}
for k in ('k_patch_voxelize', 'k_cyl_net_wg', 'k_cyl_net_h3', 'k_desc_head', 'k_select_patches_grid'):
    d = list(rows.get(k, {}).values())
    if not d:
        continue
    g = max(x['grid'] for x in d)
    d = [x for x in d if x['grid'] == g]               # full-step launches only
    kernels[k] = {'launches': len(d), 'patches_per_launch': patches}
    for c, name in (('SQ_INSTS_VALU', 'valu_per_patch'), ('SQ_INSTS_MFMA', 'mfma_per_patch'), ('SQ_WAVES', 'waves_per_patch')):
        if all(c in x for x in d):
            kernels[k][name] = sum(x[c] for x in d) / len(d) / patches
lib = ctypes.CDLL(os.path.join(ROOT, 'buffer_amd', 'libbuffer_hip.so'))
src = {f: hashlib.sha256(open(os.path.join(ROOT, 'buffer_amd', 'csrc', f), 'rb').read()).hexdigest()[:16]
       for f in ('voxelize.hip', 'convnet_wg.hip', 'convnet_h3.hip')}
json.dump(dict(source='rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES on the bench.py command of the line in `bench` '
                      '(wave-level instruction counts; SQ_INSTS_VALU includes the MFMAs)',
               lib_version=int(lib.buf_version()), csrc_sha256_16=src, bench=dict(value=bench['value'], config=cfg), kernels=kernels),
          open(out, 'w'), indent=1)
for k, v in kernels.items():
    print(k, {a: round(b, 1) if isinstance(b, float) else b for a, b in v.items()})
