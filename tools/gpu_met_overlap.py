#!/usr/bin/env python3
"""PCIe-inclusive stepping rate on workload C3 (GPU box): a new meteo snapshot every 20 steps
(DT_MET 3600 / DT_MOD 180), handed over (a) synchronously -- mphip_swap_met + mphip_update_met, the
reference's mptrac_get_met -- and (b) through mphip_prefetch_met / mphip_commit_met with the upload
running beside the time steps.  The resident rate (no hand-over) is printed for comparison."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402
from mptrac_amd.synth import synthetic_met  # noqa: E402

INTERVALS, PER = 4, 20


def run(mode):
    ctl, clim, m0, m1, atm, n_local, n_total = bench.build_inputs("C3", 0, 1, PER - 1)
    fields = tuple(m0.f3) + tuple(m0.f2)
    ctl.update(t_stop=3600.0 * (INTERVALS + 2))
    # the reference re-uses two met_t buffers; three here so that a prefetch never overwrites a resident one
    spare = [synthetic_met("C3", 3600.0 * (k + 2), 1.0 + 0.1 * k, fields=fields) for k in range(3)]
    s = hip.Simulation(ctl, clim, m0, m1, atm, n_total=n_total, shard=(0, n_local))
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    k = 0
    for _ in range(PER):                      # first interval: warm-up (first sort, page-locking of the arrays)
        s.run_timestep(k * dt)
        k += 1
    if mode == "prefetch":
        # untimed: page-lock the three host buffers once (a production run re-uses its met_t buffers)
        for w in range(3):
            spare[w].time = 3600.0 * (w + 2)
            s.prefetch_met(spare[w])
            s.commit_met()
        s.run_timestep(k * dt)
        spare[0].time = 3600.0 * 2
        s.prefetch_met(spare[0])
        while not s.prefetch_done():      # steady state: this upload ran beside the previous interval's steps
            time.sleep(0.001)
    s.synchronize()
    host_ms = []
    t0 = time.perf_counter()
    for it in range(INTERVALS):
        nxt = spare[it % 3]
        nxt.time = 3600.0 * (it + 2)
        if mode == "sync":
            s.swap_met(nxt)
        elif mode == "prefetch":
            s.commit_met()
        for j in range(PER):
            if mode == "prefetch" and j == 4:
                # a few steps are queued first: the call itself keeps the host busy for some ms
                after = spare[(it + 1) % 3]
                after.time = 3600.0 * (it + 3)
                h0 = time.perf_counter()
                s.prefetch_met(after)
                host_ms.append((time.perf_counter() - h0) * 1e3)
            s.run_timestep(k * dt)
            k += 1
    s.synchronize()
    wall = time.perf_counter() - t0
    print(f"{mode:9s}: {n_local * INTERVALS * PER / wall:.3e} particle-steps/s  ({wall / (INTERVALS * PER) * 1e3:.3f} ms per step, "
          f"{INTERVALS} hand-overs)" + (f"  host time of mphip_prefetch_met: {np.mean(host_ms):.2f} ms" if host_ms else ""), flush=True)
    s.close()


if __name__ == "__main__":
    for mode in ("resident", "sync", "prefetch"):
        run(mode)
