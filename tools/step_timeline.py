#!/usr/bin/env python3
"""Timeline of one time step from a rocprofv3 --kernel-trace CSV: the kernels between two consecutive launches of
the main step kernel (the n-th from the end), with start / end relative to the first and the queue (stream) they
ran on.   tools/step_timeline.py <dir or csv> [which-from-the-end=5] [main kernel substring=step_kernel]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sub = sys.argv[3] if len(sys.argv) > 3 else "step_kernel"
if os.path.isdir(path):
    path = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(path))]
rows.sort()
# main launches: the longest kernel family matching `sub`
mains = [k for k, r in enumerate(rows) if sub in r[2] and (r[1] - r[0]) > 300e3]
a, b = mains[-back - 1], mains[-back]
t0 = rows[a][0]
print(f"step of {(rows[b][0] - t0) / 1e3:.1f} us between two main launches")
for s, e, name, q in rows[a:b + 1]:
    short = name.replace("mphip::", "").split("(")[0][:44]
    print(f"  {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:7.1f})  q{q}  {short}")
