#!/usr/bin/env python3
"""per-kernel totals of a rocprofv3 --kernel-trace --stats run: kstat.py <dir> [top]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.2f" % (tot / 1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(7), "%10.1f us tot" % (float(r["TotalDurationNs"]) / 1e3), "%8.2f us avg" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
