#!/usr/bin/env python3
"""Cost of the modules after module_mixing on the C5 workload (GPU box): wet / dry deposition alone and together."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C5", 0, 1, 12)
s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
s.timesteps_init(0.0, 0.0)
dt = s.ctl.dt_mod
for k in range(4):
    s.run_timestep(k * dt)
s.synchronize()
for name in ("wet_depo", "dry_depo", "decay", "timesteps"):
    s.profile_begin()
    for r in range(5):
        s.module(name, 4 * dt)
    nl, ms = s.profile_end()
    print(f"{name:10s}: {ms / nl:.3f} ms per launch", flush=True)
s.close()
