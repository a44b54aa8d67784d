"""Print the top rows of a rocprofv3 kernel_stats.csv (per-kernel totals)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:n]:
    print("%-64s calls %6s total %9.2f ms avg %10.1f us %6s%%" % (r["Name"][:64], r["Calls"], int(r["TotalDurationNs"]) / 1e6,
                                                                float(r["AverageNs"]) / 1e3, r["Percentage"]))
