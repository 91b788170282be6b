#!/usr/bin/env python3
"""Kernels of the last gridded output of a bench run from a rocprofv3 --kernel-trace CSV: everything between the end of
the timed multi-step launch (or the last step kernel before the output) and the end of the output's last kernel.
  tools/grid_timeline.py <dir or csv>"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(path))]
rows.sort()
ends = [k for k, r in enumerate(rows) if "cell_sum_chains_kernel" in r[2] or "grid_accumulate_kernel" in r[2]
        or "cell_sum_groups_kernel<mphip::GridVals" in r[2]]
b = ends[-1]
a = b
while a > 0 and "step_kernel" not in rows[a][2]:
    a -= 1
t0 = rows[a][1]
print(f"gridded output: {(rows[b][1] - t0) / 1e3:.1f} us from the end of the step kernel to the end of its last kernel")
for s, e, name, q in rows[a + 1:b + 1]:
    print(f"  {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:7.1f})  q{q}  {name.replace('mphip::', '').split('(')[0][:60]}")
