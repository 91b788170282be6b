#!/usr/bin/env python3
"""Step-kernel time of workload C3 with the three integrators of module_advect (ADVECT 1 Euler, 2 midpoint -- the
reference's default --, 4 Runge-Kutta), GPU box; median of launches 35-70 (past the clock ramp)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, 80)
for advect in (4, 2, 1):
    s = hip.Simulation(dict(ctl, advect=advect), clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    for kv in sys.argv[1:]:
        name, value = kv.split("=")
        s.set_option(name, float(value))
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    s.run_timestep(0.0)
    ms = []
    for k in range(1, 71):
        s.profile_begin()
        s.run_timestep(k * dt)
        n, t = s.profile_end()
        ms.append(t / max(n, 1))
    tail = sorted(ms[35:])
    print(f"ADVECT {advect}: step kernel {tail[len(tail) // 2]:.4f} ms", flush=True)
    s.close()
