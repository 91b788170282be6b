#!/usr/bin/env python3
"""Cost of the modules after module_mixing on the C5 workload (GPU box): wet / dry deposition alone and together,
for the workload's particles and for the same particles lifted above every cloud top / surface layer (where the
modules return before they gather anything: DevMet::ps_skip / pct_skip)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C5", 0, 1, 12)
print("surface pressure %.2f ... %.2f hPa, cloud top %.2f ... %.2f hPa, particles %.1f ... %.1f hPa"
      % (np.nanmin(m0.f2["ps"]), np.nanmax(m0.f2["ps"]), np.nanmin(m0.f2["pct"]), np.nanmax(m0.f2["pct"]), atm["p"].min(), atm["p"].max()))
for label, lift in (("as the workload has them", False), ("all at 20 hPa", True)):
    a = dict(atm)
    if lift:
        a["p"] = np.full_like(atm["p"], 20.0)
    s = hip.Simulation(ctl, clim, m0, m1, a, n_total=n_total, shard=(0, n_local))
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    for k in range(2):
        s.run_timestep(k * dt)
    s.synchronize()
    print(label)
    for name in ("wet_depo", "dry_depo", "decay", "timesteps"):
        s.profile_begin()
        for r in range(5):
            s.module(name, 2 * dt)
        nl, ms = s.profile_end()
        print(f"  {name:10s}: {ms / nl:.3f} ms per launch", flush=True)
    s.close()
