#!/usr/bin/env python3
"""A grid beyond 32-bit byte offsets, for real (GPU box): 1441 x 721 columns x 180 levels = 187e6 cells, 4.49 GB of packed
wind records (the limit of the lean kernels' 32-bit offsets is 4.29 GB).  C3's modules (RK4, turbulent + mesoscale diffusion,
convection, sedimentation), 10^5 particles, one single step + six steps that share a launch + three single steps, against
the oracle.  The library takes the kBigGrid instantiations by itself (no option).  About 10 GB of host memory.
  python tools/gpu_big_grid.py zeta     the same with winds from the model levels (zeta coordinate; about 25 GB)
  python tools/gpu_big_grid.py zeta27   27 pressure levels (0.63 GiB of records: 32-bit offsets would do) under 180 model
                                        levels (4.18 GiB): the size tests must follow the model levels (met_t::npl > met_t::np)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import cases  # noqa: E402
from mptrac_amd import hip  # noqa: E402
from mptrac_amd.ctl import ctl_from_quantities  # noqa: E402
from mptrac_amd.synth import synthetic_met, synthetic_particles  # noqa: E402
from oracle import binding as B  # noqa: E402

few = "zeta27" in sys.argv[1:]
GRID = (1440, 721, 27 if few else 180)
ML = 180
cells = (GRID[0] + 1) * GRID[1] * (ML if few else GRID[2])
print(f"grid {GRID[0] + 1} x {GRID[1]} x {GRID[2]} = {cells / 1e6:.1f}e6 cells, wind records {24 * cells / 2 ** 30:.2f} GiB "
      f"(32-bit offsets reach 4.00 GiB)", flush=True)
assert 24 * cells >= 2 ** 32
zeta = few or "zeta" in sys.argv[1:]
names = ("m", "rp", "rhop", "zeta") if zeta else ("m", "rp", "rhop")
ctl = dict(cases.CASES["zeta_full" if zeta else "conv_sedi"])
ctl.update(ctl_from_quantities(names))
fields = ("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel") + (("pl", "ul", "vl", "zetal", "zeta_dotl") if zeta else ())
t0 = time.time()
m0 = synthetic_met(GRID, 0.0, 1.0, fields=fields, model_levels=ML if few else None)
m1 = synthetic_met(GRID, 3600.0, 1.25, fields=fields, model_levels=ML if few else None)
if few:
    assert m0.npl == ML and m0.np == 27 and 24 * m0.nx * m0.ny * m0.np < 2 ** 32
print(f"two synthetic snapshots in {time.time() - t0:.0f} s", flush=True)
atm = synthetic_particles(100000, seed=11, quantities=names)
if zeta:      # a vertical coordinate inside the range of the synthetic zetal field (as cases.make_case)
    atm["q"][list(names).index("zeta")] = 320.0 + 1680.0 * ((atm["lat"] + 85.0) / 170.0)
clim = cases.load_clim_tropo()
B.lib().orc_set_num_threads(B.usable_cores())
o = B.Oracle(ctl, clim, m0, m1, atm)
o.timesteps_init()
s = hip.Simulation(ctl, clim, m0, m1, atm)
s.timesteps_init(0.0, 0.0)
times = cases.step_times(o.ctl)[:10]
s.run_timestep(times[0])
s.synchronize()
s.profile_begin()
s.run_timesteps(times[1], 6)
n, ms = s.profile_end()
for t in times[7:10]:
    s.run_timestep(t)
for t in times:
    o.run_timestep(t)
g, r = s.state(), o.state()
errs = {k: cases.rel_err(g[k], r[k]) for k in ("lon", "lat", "p", "q")}
print(f"six steps in {n} launch(es); 10 steps against the oracle: " + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()), flush=True)
assert n == 1 and np.array_equal(g["time"], r["time"]) and all(v <= 1e-10 for v in errs.values())
assert s.get_cache()["rng_ctr"] == o.cache.rng_ctr
print("BIG GRID OK")
