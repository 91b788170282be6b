#!/usr/bin/env python3
"""Step-kernel time of workload C3 with the specialised and with the generic instantiation (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

for generic in (0, 1):
    ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, 30)
    s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    s.set_option("generic_kernel", generic)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    for k in range(3):
        s.run_timestep(k * dt)
    s.synchronize()
    s.profile_begin()
    for k in range(3, 23):
        s.run_timestep(k * dt)
    s.synchronize()
    nl, ms = s.profile_end()
    print("generic" if generic else "specialised", "step_kernel %.3f ms" % (ms / nl), flush=True)
    s.close()
