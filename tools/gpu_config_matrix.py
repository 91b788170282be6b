#!/usr/bin/env python3
"""Step-kernel time of a few everyday control sets on the C3 particles and grid (GPU box; one launch per step: median
of launches 35-70; then twenty steps per mphip_run_timesteps call): which instantiation of the step kernel they take
shows in the time.
  python tools/gpu_config_matrix.py [NAME=VALUE options]   e.g. generic_kernel=1 for the general instantiation"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402
from mptrac_amd.ctl import ctl_from_quantities  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, 200)
gas = ctl_from_quantities(("m",))
gas.update(qnt_rp=-1, qnt_rhop=-1)
SETS = {
    "C3 (RK4, diffusion, convection, sedimentation)": {},
    "default integrator (midpoint), same modules": dict(advect=2),
    "gas tracer: midpoint, diffusion, convection": dict(advect=2, **gas),
    "gas tracer + boundary condition + decay": dict(advect=2, bound_lat0=-90.0, bound_lat1=90.0, bound_p0=1e10, bound_p1=-1e10,
                                                     bound_dps=100.0, bound_mass=0.0, tdec_trop=259200.0, tdec_strat=259200.0,
                                                     **gas),
    "trajectories only (midpoint)": dict(advect=2, diffusion=0, conv_cape=-999.0, **gas),
}
for name, over in SETS.items():
    s = hip.Simulation(dict(ctl, **over), clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        s.set_option(k, float(v))
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
    single = tail[len(tail) // 2]
    # ... and twenty steps per mphip_run_timesteps call (steps with nothing scheduled between them share a launch)
    k = 71
    s.run_timesteps(k * dt, 20)
    k += 20
    s.synchronize()
    s.profile_begin()
    for _ in range(3):
        s.run_timesteps(k * dt, 20)
        k += 20
    n, t = s.profile_end()
    print(f"{name:52s} {single:.4f} ms   20 steps per call: {t / 60:.4f} ms per step ({n} launches for 60 steps)", flush=True)
    s.close()
