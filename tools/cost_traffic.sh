#!/bin/bash
# Development aid (GPU box): HBM-side traffic of k_cost_net on the tools/cost_probe.py workload (25 600 matches, dense inputs).
tools/prof.sh cvt_f pmc "FETCH_SIZE" -- python3 tools/cost_probe.py 25600 > /dev/null 2>&1
tools/prof.sh cvt_w pmc "WRITE_SIZE" -- python3 tools/cost_probe.py 25600 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag, ctr in (('cvt_f', 'FETCH_SIZE'), ('cvt_w', 'WRITE_SIZE')):
    v = []
    for f in glob.glob(f'gpurun_out/{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == ctr and 'k_cost_net' in r['Kernel_Name']:
                v.append(float(r['Counter_Value']))
    if v:
        print(f'k_cost_net {ctr}: {max(v) * 1024 / 25600 * (2 if ctr == "FETCH_SIZE" else 1) / 1024:.1f} KB per match (largest launch, FETCH doubled; algorithmic 25.0 in, 0.004 out)')
PY
