#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (whole device, all streams):
  tools/trace_gaps.py <dir or csv> [kernel substring]
Prints, for the kernels whose name contains the substring (default: step_kernel), the distribution of
start(k) - end(previous kernel on the device), and the busy fraction of the traced span."""
import csv
import glob
import os
import sys

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "step_kernel"
if os.path.isdir(path):
    path = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
gaps, busy, last_end = [], 0, None
for s, e, name in rows:
    if last_end is not None and sub in name:
        gaps.append((s - last_end) / 1e3)
    busy += e - s
    last_end = e if last_end is None else max(last_end, e)
span = rows[-1][1] - rows[0][0]
gaps.sort()
if gaps:
    n = len(gaps)
    print(f"{n} launches of *{sub}*: gap before the launch  median {gaps[n // 2]:.1f} us  p10 {gaps[n // 10]:.1f}  "
          f"p90 {gaps[9 * n // 10]:.1f}  max {gaps[-1]:.1f}")
print(f"kernels {len(rows)}, span {span / 1e6:.2f} ms, summed kernel time {busy / 1e6:.2f} ms")
