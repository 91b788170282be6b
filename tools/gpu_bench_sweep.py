#!/usr/bin/env python3
"""Throughput of the fused step for a few workloads / options (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402


def run(workload, interval, steps=20, warm=3, batch=1):
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
    if batch > 1:      # the time loop of a run with an output every `batch` steps: one mphip_run_timesteps call each
        for _ in range(steps // batch):
            s.run_timesteps(k * dt, batch)
            k += batch
    else:
        for _ in range(steps):
            s.run_timestep(k * dt)
            k += 1
    s.synchronize()
    wall = time.perf_counter() - t0
    nl, ms = s.profile_end()
    print(f"{workload} interval={interval:3d} steps per call={batch:3d}: {n_local * steps / wall:.3e} p-steps/s  wall/step "
          f"{wall / steps * 1e3:.3f} ms  step_kernel {ms / steps:.3f} ms per step, {nl} launches", flush=True)
    s.close()


if __name__ == "__main__":
    for wl in sys.argv[1:] or ["C3"]:
        for interval in (60, 120, 240):     # (warm = 60: past the clock ramp of a device that was idle)
            for batch in (1, 20, 60):
                run(wl, interval, steps=480, warm=60, batch=batch)
