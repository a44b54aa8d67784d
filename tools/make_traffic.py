"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as
MI355X_MICROARCH.md prescribes: they do not fit one pass) of the same bench.py command.

    python tools/make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <bench --detail-json file> <out json>

Units and gfx950 correction (MI355X_MICROARCH.md, HBM): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB (1024 B);
FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> doubled; WRITE_SIZE taken as is."""
import collections
import csv
import json
import sys

fetch_csv, write_csv, bench_file, out = sys.argv[1:5]
bench = json.load(open(bench_file))            # the uncompacted record (bench.py --detail-json)


# kernels whose launch covers a FIXED number of units (one workgroup per patch): only the dispatches with the full-step grid count.
# (Round 4 averaged one 64-patch accuracy-probe launch of each CNN kernel into the mean: traffic under-reported by 1/6 and 1/4.)
FULL_GRID_ONLY = ('k_cyl_net_wg', 'k_cyl_net_h3', 'k_desc_head', 'k_patch_voxelize', 'k_select_patches_grid')
dropped = collections.defaultdict(int)


def per_launch(path, counter):
    """mean counter value per dispatch and kernel over the FULL-STEP launches (BUF_NO_TRAFFIC=1 turns the single-pair latency
    measurements and the 64-patch float64 probe of bench.py off; a dispatch of a fixed-unit kernel with a smaller grid than the
    largest one seen is dropped anyway and counted in `launches_dropped`)"""
    rows = collections.defaultdict(dict)              # kernel -> dispatch id -> [grid, value]
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        e = rows[k].setdefault(r['Dispatch_Id'], [int(r['Grid_Size']), 0.0])
        e[1] += float(r['Counter_Value'])
    mean, n = {}, {}
    for k, d in rows.items():
        vals = list(d.values())
        if k in FULL_GRID_ONLY:
            g = max(v[0] for v in vals)
            dropped[k] = max(dropped[k], sum(1 for v in vals if v[0] != g))
            vals = [v for v in vals if v[0] == g]
        mean[k] = sum(v[1] for v in vals) / len(vals) * 1024.0
        n[k] = len(vals)
    return mean, n


fetch, n_f = per_launch(fetch_csv, 'FETCH_SIZE')
write, n_w = per_launch(write_csv, 'WRITE_SIZE')
# the VN gather blocks as ONE operator (round 4): block 0 on k_vn_gather6_lds, the four resnet blocks on k_vn_linear_pre + k_vn_gather_pre;
# per launch of the OPERATOR = bytes of all three kernels / number of block calls (gather6 + gather_pre dispatches)
for d, n in ((fetch, n_f), (write, n_w)):
    ks = [k for k in ('k_vn_gather6_lds', 'k_vn_gather_pre', 'k_vn_linear_pre') if k in d]
    if ks and 'k_vn_gather' not in d:
        calls = n.get('k_vn_gather6_lds', 0) + n.get('k_vn_gather_pre', 0)
        d['k_vn_gather'] = sum(d[k] * n[k] for k in ks) / max(calls, 1)
        n['k_vn_gather'] = calls
# A2 as ONE operator: the cell-centric kernel (self queries) and the query-centric one (pool / upsample queries) of a pyramid
for d, n in ((fetch, n_f), (write, n_w)):
    ks = [k for k in ('k_grid_query_cell', 'k_grid_query_wave') if k in d]
    if ks:
        tot = sum(n[k] for k in ks)
        d['k_grid_query'] = sum(d[k] * n[k] for k in ks) / tot
        n['k_grid_query'] = tot
cfg = bench['config']
pairs = cfg['pairs_per_step_per_gpu']
patches = 2 * cfg['keypoints_per_fragment'] * pairs
cost = [o for o in bench['roofline_other'] if o['kernel'].startswith('k_cost_net')][0]
matches = cost['avg_algorithmic_flops'] / 51905536.0
# unit of work per launch and the algorithmic bytes per unit (SURVEY 8d formulas)
units = {
    'k_cyl_net_wg': (patches, 'patch', 48 * 140 * 4 + 32 * 140 * 4),
    'k_desc_head': (patches, 'patch', 2 * 32 * 140 * 4 + 128),
    'k_patch_voxelize': (patches, 'patch', 12 * 512 + 4 * 16 * 420),
    'k_select_patches_grid': (patches, 'patch', 12 * 512 + 12),
    'k_cost_net': (matches, 'match', 2 * 3200 * 4 + 4),
    'k_cyl_net_h3': (patches, 'patch', 48 * 140 * 4 + 32 * 140 * 4),
    'k_cost_net_h3': (matches, 'match', 2 * 3200 * 4 + 4),
    'k_grid_query': (pairs, 'pair (mean over the 7 query shapes: 3 cell-centric self queries, 4 query-centric)', None),
    'k_grid_query_cell': (pairs, 'pair (mean over the 3 self queries)', None),
    'k_grid_query_wave': (pairs, 'pair (mean over the 4 pool / upsample queries)', None),
    'k_vn_gather': (pairs, 'pair (mean over the 5 blocks)', None),
    'k_nn1': (pairs, 'pair', None),
    'k_fps': (pairs, 'pair', None),
}
kernels = {}
for k, (u, unit, alg) in units.items():
    if k not in fetch or k not in write:
        continue
    hbm = 2.0 * fetch[k] + write[k]
    kernels[k] = dict(fetch_size_bytes_per_launch_raw=fetch[k], write_size_bytes_per_launch=write[k], hbm_bytes_per_launch=hbm,
                      launches_profiled=n_f[k], launches_dropped=dropped.get(k, 0), units_per_launch=u, unit=unit, hbm_bytes_per_unit=hbm / u,
                      algorithmic_bytes_per_unit=alg)
json.dump(dict(source='rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on the bench.py command of '
                      'the line in `bench`; per-dispatch means over the full-step launches',
               correction='gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is; '
                          'counters are in KB (1024 B)',
               bench=dict(value=bench['value'], config=cfg), kernels=kernels), open(out, 'w'), indent=1)
for k, v in kernels.items():
    print(f"{k:26s} {v['hbm_bytes_per_unit']:12.1f} B per {v['unit']}" + (f"   (algorithmic {v['algorithmic_bytes_per_unit']})" if v['algorithmic_bytes_per_unit'] else ''))
