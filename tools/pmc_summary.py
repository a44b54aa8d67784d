"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (sum over dispatches, and per-dispatch mean)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if pat and pat not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(disp[k])
    print(k[:70], "dispatches", n)
    for c, x in sorted(v.items()):
        print("   %-28s total %16.0f   per dispatch %14.1f" % (c, x, x / n))
