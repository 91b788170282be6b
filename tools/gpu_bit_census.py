"""How many values of the final state are NOT the oracle's bits, per named case of tests/cases.py after its steps:
the census behind DESIGN.md section 2 (MPTRAC_AMD_EXACT=1 in front: the reference-rounding build; MPHIP_LIB=...: any).
    python tools/gpu_bit_census.py [--json] [case ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases                       # noqa: E402
from mptrac_amd import hip         # noqa: E402
from oracle import binding as B    # noqa: E402


def census(name, n=10000, steps=None):
    ctl, clim, m0, m1, atm = cases.make_case(name, n=n)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    cases.prepare(o)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    cases.prepare(s)
    s.timesteps_init(atm["time"].min(), atm["time"].max())
    times = cases.step_times(o.ctl)[:steps]
    for t in times:
        o.run_timestep(t)
        s.run_timestep(t)
    g, r = s.state(), o.state()
    s.close()
    out = {k: int(np.count_nonzero(g[k] != r[k])) for k in ("lon", "lat", "p")}
    same = (g["q"] == r["q"]) | (np.isnan(g["q"]) & np.isnan(r["q"]))
    out["q"] = int(np.count_nonzero(~same))
    if out["q"]:
        ml = cases.CASES[name].get("advect_vert_coord", 0) in (1, 3)
        rows = cases.CASE_QUANTITIES.get(name, cases.QUANTITIES_ML if ml else cases.QUANTITIES)
        out["q rows"] = {rows[k]: int(np.count_nonzero(~same[k])) for k in range(same.shape[0]) if not same[k].all()}
    out["uvwp"] = int(np.count_nonzero(g["uvwp"] != r["uvwp"]))
    worst = max(cases.rel_err(g[k], r[k]) for k in ("lon", "lat", "p"))
    return len(times), out, worst


if __name__ == "__main__":
    import json
    args = sys.argv[1:]
    as_json = "--json" in args       # (tests/test_gpu_exact_library.py reads this)
    names = [a for a in args if not a.startswith("--")] or list(cases.CASES)
    print("library:", os.path.basename(hip.lib_path()))
    for name in names:
        k, out, worst = census(name)
        if as_json:
            print("JSON " + json.dumps(dict(case=name, steps=k, worst=worst, **out)))
        else:
            print(f"{name:24s} {k:3d} steps, of 10000 particles not the oracle's bits: {out}  worst rel {worst:.1e}")
