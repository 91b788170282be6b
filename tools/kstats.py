#!/usr/bin/env python3
"""Print the head of a rocprofv3 --kernel-trace --stats CSV summary (kernel, calls, average us, share)."""
import csv
import glob
import sys

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
files = glob.glob(root + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no kernel_stats.csv under " + root)
for r in list(csv.DictReader(open(files[0])))[:top]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1e3:10.1f} us {float(r["Percentage"]):6.2f} %')
