#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as per-iteration microseconds.  usage: stats_table.py <dir> [iterations] [rows]"""
import csv, glob, sys
d = sys.argv[1]
it = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = sorted(glob.glob(d + "/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print("%-90s %4d %8.1f us/it  avg %7.1f" % (r["Name"][:90], int(r["Calls"]), int(r["TotalDurationNs"]) / it / 1e3, float(r["AverageNs"]) / 1e3))
print("total %.1f us per iteration" % (tot / it / 1e3))
