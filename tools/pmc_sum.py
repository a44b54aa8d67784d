"""Sum rocprofv3 counter_collection CSV rows per kernel and counter (development aid)."""
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
for k in acc:
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in acc[k].items():
        print(f'   {c:32s} {v:.4g}  ({n[(k, c)]} rows, {v / n[(k, c)]:.4g} per launch)')
