#!/usr/bin/env python3
"""HBM bytes per time step of a whole bench workload from a tools/profile_traffic.sh run: every dispatch of the
TIMED region counts (the step kernels, module_sort's passes, module_mixing's kernels, the deposition launch, the
gridded output ...), FETCH_SIZE x the calibration factor of profiles/pmc_traffic.json (pack_kernel's known reads: the
gfx950 half-count) + WRITE_SIZE, KiB -> bytes, divided by the steps of the region.

The run is `bench.py --steps 10 --warmup 2`: dispatches are taken in order; the timed region is the ten steps
between the gridded output that ends the warm-up and the one that ends the timed steps (grid_records_kernel or
grid_accumulate marks an output).  Prints per-kernel means and writes <dir>/traffic.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, workload = sys.argv[1], sys.argv[2]
STEPS = 10
calib = 1.98
try:
    calib = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["_raw"]["calib_fetch_factor"]
except Exception:
    pass


def dispatches(kind, counter):
    rows = []
    for f in glob.glob(os.path.join(src, kind, "**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0))
    acc = defaultdict(float)
    name = {}
    for d, k, v in rows:            # (one row per dimension instance: sum per dispatch)
        acc[d] += v
        name[d] = k
    return [(d, name[d], acc[d]) for d in sorted(acc)]


def timed(rows):
    """dispatches behind the gridded output that ends the warm-up, up to and including the output that ends the
    timed steps (an output starts with grid_records_kernel / grid_accumulate_kernel and ends with its last sum kernel)"""
    starts = [i for i, (_, k, _) in enumerate(rows) if "grid_records_kernel" in k or "grid_accumulate_kernel" in k]
    if len(starts) < 2:
        return rows

    def end_of(i):
        j = i
        while j < len(rows) and not ("cell_sum_chains_kernel" in rows[j][1] or "cell_sum_groups_kernel<mphip::GridVals" in rows[j][1]
                                     or "grid_accumulate_kernel" in rows[j][1]):
            j += 1
        return min(j, len(rows) - 1)
    return rows[end_of(starts[-2]) + 1:end_of(starts[-1]) + 1]


fetch, write = timed(dispatches("fetch", "FETCH_SIZE")), timed(dispatches("write", "WRITE_SIZE"))
per = defaultdict(lambda: [0, 0.0, 0.0])
for _, k, v in fetch:
    per[k.split("(")[0][:70]][0] += 1
    per[k.split("(")[0][:70]][1] += v * calib
for _, k, v in write:
    per[k.split("(")[0][:70]][2] += v
total = sum(v[1] + v[2] for v in per.values())
print(f"workload {workload}: {len(fetch)} dispatches in the timed region ({STEPS} steps + one gridded output); FETCH_SIZE x {calib:.3f} + WRITE_SIZE")
for k, (n, f, w) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"  {k:70s} {n:5d} dispatches  read {f / STEPS / 1e6:9.1f} MB/step  written {w / STEPS / 1e6:9.1f} MB/step")
print(f"HBM traffic per time step: {total / STEPS / 1e9:.3f} GB")
json.dump({"workload": workload, "traffic_bytes_per_step": total / STEPS, "calib_fetch_factor": calib,
           "dispatches": len(fetch)}, open(os.path.join(src, "traffic.json"), "w"), indent=1)
