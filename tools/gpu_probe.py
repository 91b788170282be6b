#!/usr/bin/env python3
"""First-contact probe on the GPU box: parity numbers per case + a quick
throughput figure.  Prints, never asserts (tests/ hold the assertions)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from mptrac_amd import hip  # noqa: E402
from oracle import binding as B  # noqa: E402


def run_case(name, n=10000, grid="C1"):
    ctl, clim, m0, m1, atm = cases.make_case(name, n=n, grid=grid)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(atm["time"].min(), atm["time"].max())
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
        s.run_timestep(t)
    g = s.state()
    r = o.state()
    errs = {k: cases.rel_err(g[k], r[k]) for k in ("time", "lon", "lat", "p")}
    errs["q"] = cases.rel_err(g["q"], r["q"])
    errs["uvwp"] = cases.rel_err(g["uvwp"], r["uvwp"])
    print(f"case {name:16s} n={n} " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()), flush=True)
    s.close()


def quick_bench(name, n, grid, steps=20):
    fields = ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel")
    ctl, clim, m0, m1, atm = cases.make_case(name, n=n, grid=grid, fields=fields, quantities=("m", "rp", "rhop"))
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    ts = cases.step_times(s.ctl)
    s.run_timestep(ts[0])
    s.run_timestep(ts[1])
    s.synchronize()
    s.profile_begin()
    t0 = time.time()
    for t in ts[2:2 + steps]:
        s.run_timestep(t)
    s.synchronize()
    wall = time.time() - t0
    nl, ms = s.profile_end()
    print(f"bench {name} n={n} grid={grid}: {n * steps / wall:.3e} p-steps/s wall, "
          f"kernel {ms / max(nl, 1):.3f} ms/launch over {nl} launches", flush=True)
    s.close()


if __name__ == "__main__":
    for name in cases.CASES:
        try:
            run_case(name)
        except Exception as e:   # keep going: this is a probe
            print(f"case {name}: FAILED {type(e).__name__}: {e}", flush=True)
    for name, n, grid in (("advect", 10 ** 6, "C2"), ("conv_sedi", 10 ** 6, "C2"), ("advect", 10 ** 7, "C3"),
                          ("conv_sedi", 10 ** 7, "C3")):
        try:
            quick_bench(name, n, grid)
        except Exception as e:
            print(f"bench {name}: FAILED {type(e).__name__}: {e}", flush=True)
