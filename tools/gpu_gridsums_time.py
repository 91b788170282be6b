#!/usr/bin/env python3
"""Wall time of one gridded output (mphip_grid_sums, ordered sums) of workload C3 after 30 steps, GPU box.
  python tools/gpu_gridsums_time.py [NAME=VALUE options]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, 60)
s = hip.Simulation(ctl, clim, m0, m1, atm)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    s.set_option(k, float(v))
s.timesteps_init(0.0, 0.0)
dt = s.ctl.dt_mod
for k in range(31):
    s.run_timestep(k * dt)
s.grid_sums(30 * dt)
s.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    cnt, mean, sig = s.grid_sums(30 * dt)
    ts.append((time.perf_counter() - t0) * 1e3)
print("grid_sums wall ms:", " ".join("%.3f" % t for t in ts), " particles binned", int(cnt.sum()), " checksum %.17g" % mean.sum())
s.close()
