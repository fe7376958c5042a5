#!/usr/bin/env python3
"""Print the rocprofv3 --kernel-trace --stats summary (top kernels) found under a directory."""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for f in glob.glob(d + "/**/*_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f)
    for r in rows[:top]:
        print(f"{r['Name'][:56]:56s} calls={int(r['Calls']):6d} total_ms={float(r['TotalDurationNs'])/1e6:10.3f} "
              f"avg_us={float(r['AverageNs'])/1e3:10.2f} pct={float(r['Percentage']):6.2f}")
