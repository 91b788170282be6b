#!/usr/bin/env python3
"""Gather micro-benchmark (GPU box): lane-by-lane vs quad-cooperative stencil
fetch on the C3 workload, unsorted and cell-sorted."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, 8)
for sorted_ in (0, 1):
    s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    s.set_option("locality_sort_interval", 1 if sorted_ else 0)
    s.timesteps_init(0.0, 0.0)
    s.run_timestep(0.0)
    s.run_timestep(180.0)
    for reps in (1, 4):
        for mode in (0, 1):
            ms, chk = s.test_gather(mode, reps)
            print(f"sorted={sorted_} reps={reps} mode={mode}: {ms:.3f} ms  checksum {chk!r}", flush=True)
    s.close()
