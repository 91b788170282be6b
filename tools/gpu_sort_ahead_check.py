#!/usr/bin/env python3
"""module_sort ahead of time (option sort_ahead) at a size where its kernels really overlap the rest of the
step: C5's module set on 3e6 particles, 18 steps with a gridded output in between, option on / off / on --
every array must come out with identical bits."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mptrac_amd import hip  # noqa: E402

ctl, clim, met0, met1, atm, n_local, n_total = bench.build_inputs("C5", 0, 1, 30, particles=3e6)
res = []
for ahead in (1, 0, 1):
    s = hip.Simulation(ctl, clim, met0, met1, atm)
    s.set_option("sort_ahead", ahead)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    for k in range(18):
        s.run_timestep(k * dt)
        if k == 7:
            s.grid_sums(k * dt)
    res.append((s.get_atm(), s.get_cache()))
    s.close()
for a in (1, 2):
    for key in ("time", "lon", "lat", "p", "q"):
        assert np.array_equal(res[0][0][key], res[a][0][key]), (a, key)
    assert np.array_equal(res[0][1]["uvwp"], res[a][1]["uvwp"]) and np.array_equal(res[0][1]["dt"], res[a][1]["dt"])
print("sort ahead on / off / on: identical bits, %d particles x 18 steps" % n_total)
