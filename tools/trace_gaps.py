"""GPU busy/idle analysis of a rocprofv3 kernel_trace.csv: share of the wall time with at least one kernel running, and the
largest idle gaps (with the kernels before/after), over the last `frac` of the trace (steady state)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t1 - (t1 - t0) * frac
ev = [e for e in ev if e[0] >= lo]
busy, gaps, cur_end, last = 0, [], ev[0][0], ev[0][2]
start = ev[0][0]
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, last, n))
        busy += 0
        cur_start = s
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end, last = e, n
wall = cur_end - start
print(f'window {wall/1e6:.1f} ms, busy {busy/1e6:.1f} ms = {100*busy/wall:.2f} %, idle {100*(1-busy/wall):.2f} % in {len(gaps)} gaps')
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f'  gap {g/1e3:8.1f} us   after {a:40s} before {b}')
tot = {}
for s, e, n in ev:
    tot[n] = tot.get(n, 0) + e - s
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:8]:
    print(f'  {n:40s} {v/1e6:9.1f} ms ({100*v/wall:.1f} % of the window)')
