#!/usr/bin/env python3
"""Write tests/golden/oracle_<case>.npz: final state of the CPU oracle after
the 21 calls of mptrac_run_timestep of each parity case (1000 particles on the
C1 grid; inputs are regenerated analytically, only outputs are stored).

Provenance: produced by oracle/ (this repo's restatement, pinned to the
reference's own goldens by tests/test_oracle_pins.py), NOT by the reference
binary -- the reference cannot be built in this image.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import cases  # noqa: E402
from oracle import binding as B  # noqa: E402

N = 1000


def run_case(name):
    ctl, clim, m0, m1, atm = cases.make_case(name, n=N)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    cases.prepare(o)
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
    s = o.state()
    s["rng_ctr"] = np.array([o.cache.rng_ctr], dtype=np.uint64)
    return s


if __name__ == "__main__":
    for name in cases.CASES:
        out = os.path.join(HERE, "golden", f"oracle_{name}.npz")
        np.savez_compressed(out, **run_case(name))
        print("wrote", out, os.path.getsize(out), "bytes")
