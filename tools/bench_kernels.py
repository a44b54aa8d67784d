import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line)
        print('pairs/s', round(d['value'],2), {k: round(v,3) for k,v in d['timed_kernel_ms_per_step'].items()})
        for o in d['roofline_other']:
            if 'vn_gather' in o['kernel'] or 'grid_query' in o['kernel'] or 'select' in o['kernel']: print('  ', o['kernel'][:30], round(o['avg_us'],1), 'us frac', round(o['frac'],4))
