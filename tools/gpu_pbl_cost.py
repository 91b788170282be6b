#!/usr/bin/env python3
"""What the boundary-layer closure (TURB_PBL_SCHEME 1, module_diff_pbl) costs on workload C3p (GPU box): the same 10^7
particles and modules (mesoscale diffusion horizontal only)
  a  without the closure (exact lean instantiation),
  b  with it (gated instantiation + kPblClosure),
  c  with it, but every particle lifted above 3 km (no particle inside any boundary layer: what the instantiation
     itself costs -- registers, switches -- against a),
each as 20 steps in one call and as 10 single steps; step-kernel time per step from HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
ctl, clim, met0, met1, atm, _, _ = bench.build_inputs("C3p", 0, 1, 64, particles=n)
inside = int(np.count_nonzero(atm["p"] > 880.0))
print(f"{n} particles, {inside} of them below 880 hPa ({100.0 * inside / n:.1f} %)")
lifted = dict(atm, p=np.minimum(atm["p"], 1013.25 * np.exp(-3.0 / 7.0)))
for name, over, particles in (("a  no closure", dict(turb_pbl_scheme=0), atm), ("b  closure", {}, atm),
                              ("c  closure, all particles above 3 km", {}, lifted)):
    s = hip.Simulation(dict(ctl, **over), clim, met0, met1, particles)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    s.run_timestep(0.0)
    s.run_timesteps(dt, 20)        # (first steps: locality order, clocks)
    s.synchronize()
    s.profile_begin()
    s.run_timesteps(21 * dt, 20)
    l1, ms1 = s.profile_end()
    s.profile_begin()
    for k in range(41, 51):
        s.run_timestep(k * dt)
    l2, ms2 = s.profile_end()
    g = s.get_atm()
    assert np.all(np.isfinite(g["p"])) and np.all(g["time"] == 50 * dt)
    print(f"{name:40s} 20 steps per call: {ms1 / 20:.4f} ms per step ({l1} launch)   single steps: {ms2 / 10:.4f} ms")
    s.close()
