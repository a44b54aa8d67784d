#!/bin/bash
# kernel-trace of a short bench run; prints which kernels run beside the cost-volume CNN (overlap of the matching stage)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export BUF_BENCH_NO_FORK=1 BUF_NO_TRAFFIC=1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovtrace -o ov -- python3 bench.py --steps 4 --warmup 1 --no-split > gpurun_out/ovtrace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ovtrace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40], r.get('Queue_Id'), r.get('Stream_Id')) for r in rows]
ev.sort()
cost = [e for e in ev if 'k_cost_net' in e[2]]
print('cost dispatches', len(cost), 'queues', collections.Counter((e[2], e[3], e[4]) for e in ev if e[1]-e[0] > 1_000_000))
for c in cost[-6:]:
    ov = collections.Counter()
    for e in ev:
        if e is c: continue
        lo, hi = max(e[0], c[0]), min(e[1], c[1])
        if hi > lo: ov[e[2]] += (hi - lo) / 1e6
    print('cost %.2f ms; beside it (ms):' % ((c[1]-c[0])/1e6), {k: round(v, 2) for k, v in ov.most_common(6)})
PY
