#!/usr/bin/env python3
"""Throughput of the fused step for a few workloads / options (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402


def run(workload, interval, steps=20, warm=3):
    ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs(workload, 0, 1, steps + warm + 1)
    s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    s.set_option("locality_sort_interval", interval)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    k = 0
    for _ in range(warm + 1):
        s.run_timestep(k * dt)
        k += 1
    s.synchronize()
    s.profile_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.run_timestep(k * dt)
        k += 1
    s.synchronize()
    wall = time.perf_counter() - t0
    nl, ms = s.profile_end()
    print(f"{workload} interval={interval:3d}: {n_local * steps / wall:.3e} p-steps/s  wall/step {wall / steps * 1e3:.3f} ms  "
          f"step_kernel {ms / nl:.3f} ms x {nl}", flush=True)
    s.close()


if __name__ == "__main__":
    for wl in sys.argv[1:] or ["C3"]:
        for interval in (30, 60, 120, 240, 480):     # (warm = 60: past the clock ramp of a device that was idle)
            run(wl, interval, steps=480, warm=60)
