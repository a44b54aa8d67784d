#!/bin/bash
# Development aid (GPU box): HBM traffic and time of the Cylindrical_Net kernels on the tools/wg_probe.py workload.
tools/prof.sh wgt_f pmc "FETCH_SIZE" -- python3 tools/wg_probe.py 20000 > /dev/null 2>&1
tools/prof.sh wgt_w pmc "WRITE_SIZE" -- python3 tools/wg_probe.py 20000 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag, ctr in (('wgt_f', 'FETCH_SIZE'), ('wgt_w', 'WRITE_SIZE')):
    acc = collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == ctr and 'k_cyl_net' in r['Kernel_Name']:
                acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    for k, v in acc.items():
        big = max(v)
        print(f'{k:14s} {ctr}: {big * 1024 / 20000 * (2 if ctr == "FETCH_SIZE" else 1) / 1024:.1f} KB per patch (largest launch, FETCH doubled)')
PY
python3 tools/wg_probe.py 20000 2>/dev/null | grep patches
