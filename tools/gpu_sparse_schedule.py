#!/usr/bin/env python3
"""C5's particles, grid and modules (decay, wet / dry deposition, mixing, module_sort) with the sort and the mixing due
once an hour instead of in every step (GPU box): wall time per step of 120 steps, one mphip_run_timestep call per step
against twenty steps per mphip_run_timesteps call (launches shared between the steps at which something is due)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C5", 0, 1, 300)
ctl = dict(ctl, sort_dt=3600.0, mixing_dt=3600.0)
for batch in (1, 20):
    s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    k = 0
    for _ in range(41):          # (past the first sort / mixing step and the clock ramp)
        s.run_timestep(k * dt)
        k += 1
    s.synchronize()
    s.profile_begin()
    t0 = time.perf_counter()
    if batch == 1:
        for _ in range(120):
            s.run_timestep(k * dt)
            k += 1
    else:
        for _ in range(6):
            s.run_timesteps(k * dt, 20)
            k += 20
    s.synchronize()
    wall = time.perf_counter() - t0
    n, ms = s.profile_end()
    print(f"steps per call {batch:3d}: {n_local * 120 / wall:.3e} particle-steps/s  wall {wall / 120 * 1e3:.3f} ms per step  "
          f"step kernels {ms / 120:.3f} ms per step in {n} launches", flush=True)
    s.close()
