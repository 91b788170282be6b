#!/usr/bin/env python3
"""Option lds_tile (SURVEY x1 in the real kernel: traj_tile_kernel) on pure trajectories, C3's grid and particles (GPU
box): step-kernel time per step of 20-step launches without a tile and with tiles of 384 / 1024 / 1700 cells, RK4 and
the midpoint scheme; positions compared bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
ctl, clim, met0, met1, atm, _, _ = bench.build_inputs("C3", 0, 1, 90, particles=n)
ctl = dict(ctl, diffusion=0, conv_cape=-999.0, qnt_rp=-1, qnt_rhop=-1)      # trajectories only
print(f"{n} particles, 721 x 361 x 137 grid, trajectories only (module_timesteps, module_position, module_advect)")
for advect in (4, 2):
    ref = None
    for cells in (0, 384, 1024, 1700):
        s = hip.Simulation(dict(ctl, advect=advect), clim, met0, met1, atm)
        s.set_option("lds_tile", cells)
        s.timesteps_init(0.0, 0.0)
        dt = s.ctl.dt_mod
        s.run_timestep(0.0)
        s.run_timesteps(dt, 20)
        s.synchronize()
        times = []
        for rep in range(3):
            s.profile_begin()
            s.run_timesteps((21 + 20 * rep) * dt, 20)
            launches, ms = s.profile_end()
            times.append(ms / 20)
        g = s.get_atm()
        same = "" if ref is None else ("   identical bits" if all(np.array_equal(g[k], ref[k]) for k in ("time", "lon", "lat", "p")) else "   DIFFERENT")
        if ref is None:
            ref = g
        print(f"ADVECT {advect}  tile {cells:5d} cells ({cells * 24 / 1024:5.1f} KB)  {min(times):.4f} ms per step (three launches: "
              + ", ".join(f"{t:.4f}" for t in times) + f"; {launches} launch per 20 steps){same}")
        s.close()
