#!/usr/bin/env python3
"""Step-kernel time of everyday control sets with winds from the model levels (C3z particles and grid, GPU box):
one launch per step (median of launches 35-70) and twenty steps per mphip_run_timesteps call.  Which instantiation
of the step kernel a set takes shows in the time.
  python tools/gpu_ml_subsets.py [NAME=VALUE options]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3z", 0, 1, 200)
SETS = {
    "C3z (RK4, diffusion, convection, sedimentation)": {},
    "default integrator (midpoint), same modules": dict(advect=2),
    "midpoint, diffusion, no convection": dict(advect=2, conv_cape=-999.0),
    "trajectories only (midpoint)": dict(advect=2, diffusion=0, conv_cape=-999.0, qnt_rp=-1, qnt_rhop=-1),
    "trajectories only (RK4)": dict(diffusion=0, conv_cape=-999.0, qnt_rp=-1, qnt_rhop=-1),
    "C3z modules + boundary condition (mass)": dict(bound_lat0=-90.0, bound_lat1=90.0, bound_p0=1e10, bound_p1=-1e10,
                                                    bound_dps=100.0, bound_mass=0.0),
}
print(f"{'':52s} {'one launch per step':>20s} {'20 steps per call':>20s}   (ms per step)")
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
    k = 71
    s.run_timesteps(k * dt, 20)       # (warm: the first multi-step launch)
    k += 20
    s.synchronize()
    s.profile_begin()
    for _ in range(3):
        s.run_timesteps(k * dt, 20)
        k += 20
    n, t = s.profile_end()
    print(f"{name:52s} {single:20.4f} {t / 60:20.4f}   ({n} launches for 60 steps)", flush=True)
    s.close()
