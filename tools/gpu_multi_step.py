#!/usr/bin/env python3
"""Time per step of mphip_run_timesteps batches against one launch per step (workload C3 or --particles):
  python tools/gpu_multi_step.py [particles] [batch ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
opts = [a for a in sys.argv[2:] if "=" in a]
batches = [int(x) for x in sys.argv[2:] if "=" not in x] or [1, 2, 5, 10, 20, 60]
ctl, clim, met0, met1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, 400, particles=n)
sim = hip.Simulation(ctl, clim, met0, met1, atm)
for kv in opts:
    name, value = kv.split("=")
    sim.set_option(name, float(value))
sim.timesteps_init(0.0, 0.0)
dt = sim.ctl.dt_mod
sim.run_timestep(0.0)
k = 1
for _ in range(40):         # clocks
    sim.run_timestep(k * dt)
    k += 1
sim.synchronize()
for b in batches:
    sim.set_option("multi_step", b if b > 1 else 0)
    reps = max(1, 60 // b)
    sim.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sim.run_timesteps(k * dt, b)
        k += b
    sim.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / (reps * b)
    print(f"batch {b:3d}: {ms:.4f} ms per step  ({n_local / ms * 1e3:.3e} particle-steps/s)", flush=True)
sim.close()
