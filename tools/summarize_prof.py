#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ tree (tools/profile.sh) into the text
summary that is committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1]
out = []


def find(pattern):
    return sorted(glob.glob(os.path.join(src, pattern), recursive=True))


for f in find("stats/**/*kernel_stats.csv"):
    out.append(f"== kernel stats ({os.path.relpath(f, src)}) ==")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        out.append("  {Name:60.60s} calls={Calls:>6s} total_ns={TotalDurationNs:>14s} avg_ns={AverageNs:>14s} pct={Percentage}".format(**r))
bj = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(bj):
    out.append("== bench line under the profiler ==")
    out.append("  " + open(bj).read().strip())

for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.append(f"== PMC {os.path.basename(d)} (step_kernel dispatches, per-dispatch mean) ==")
        for k, v in sorted(acc.items()):
            # rocprofv3 emits one row per dispatch (and per dimension instance); average per dispatch
            out.append(f"  {k:34s} n={len(v):4d} mean={sum(v) / len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
print("\n".join(out))
