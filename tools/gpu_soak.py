#!/usr/bin/env python3
"""Long run against the oracle (GPU box): 10^5 particles, every module of the `full` case (sort, mixing, decay,
wet / dry deposition, convection, sedimentation, diffusion) plus module_meteo quantities, 400 time steps over
20 h with three meteo hand-overs through mphip_prefetch_met / mphip_commit_met and downloads every 97 steps.
Prints the relative deviations on the way; fails above the 1e-10 bar.
  python tools/gpu_soak.py batched     the C3 module set (`conv_sedi`: nothing scheduled between the steps) handed to
                                       the device twenty steps at a time (mphip_run_timesteps: shared launches)
  python tools/gpu_soak.py zeta        the same hand-over with winds from the model levels (`zeta_full`: zeta
                                       advection, diffusion, convection, sedimentation -- the lean model-level kernels);
                                       the level search of a step starts from the index the step before stored, which
                                       after a meteo hand-over points into fields that have changed
  python tools/gpu_soak.py fullbatched the full module set (module_sort every 1800 s, mixing every 900 s, module_meteo
                                       quantities) handed over twenty steps at a time: launches are shared between
                                       the steps at which the sort or the mixing is due
  python tools/gpu_soak.py everystep   the full module set with module_sort and module_mixing in EVERY step (the schedule of
                                       BASELINE configs[4]): the sort that runs ahead repairs the previous order 400 times
                                       in a row, across the meteo hand-overs"""
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
from mptrac_amd.synth import FIELDS_METEO_ONLY, synthetic_met, synthetic_particles  # noqa: E402
from oracle import binding as B  # noqa: E402

zeta = "zeta" in sys.argv[1:]
fullbatched = "fullbatched" in sys.argv[1:]
everystep = "everystep" in sys.argv[1:]
batched = "batched" in sys.argv[1:] or zeta or fullbatched
if fullbatched:
    names = ("m", "vmr", "rp", "rhop", "loss_rate", "mloss_decay", "mloss_wet", "mloss_dry", "aoa", "t", "u", "ps", "theta")
    ctl = dict(cases.CASES["full"])
    ctl.update(ctl_from_quantities(names))
    ctl.update(t_stop=4 * 18000.0, dt_met=18000.0, met_dt_out=0.1, sort_dt=1800.0, mixing_dt=900.0)
elif zeta:
    names = tuple(cases.QUANTITIES_ML)
    ctl = dict(cases.CASES["zeta_full"])
    ctl.update(ctl_from_quantities(names))
    ctl.update(t_stop=4 * 18000.0, dt_met=18000.0, met_dt_out=0.0)
elif batched:
    names = ("m", "rp", "rhop")
    ctl = dict(cases.CASES["conv_sedi"])
    ctl.update(ctl_from_quantities(names))
    ctl.update(t_stop=4 * 18000.0, dt_met=18000.0, met_dt_out=0.0)
else:
    names = ("m", "vmr", "rp", "rhop", "loss_rate", "mloss_decay", "mloss_wet", "mloss_dry", "aoa", "t", "u", "ps", "theta")
    ctl = dict(cases.CASES["full"])
    ctl.update(ctl_from_quantities(names))
    ctl.update(t_stop=4 * 18000.0, dt_met=18000.0, met_dt_out=0.1, sort_dt=1800.0, mixing_dt=900.0)
    if everystep:
        ctl.update(sort_dt=ctl["dt_mod"], mixing_dt=ctl["dt_mod"])
fields = None if zeta else tuple(cases.PRESSURE_LEVEL_FIELDS) + tuple(FIELDS_METEO_ONLY)   # (None: model-level fields too)
mets = [synthetic_met("C1", 18000.0 * k, 1.0 + 0.1 * k, fields=fields) for k in range(6)]
atm = synthetic_particles(100000, seed=7, quantities=names)
if zeta:      # a vertical coordinate inside the range of the synthetic zetal field (as cases.make_case)
    atm["q"][list(names).index("zeta")] = 320.0 + 1680.0 * ((atm["lat"] + 85.0) / 170.0)
clim = cases.load_clim_tropo()
B.lib().orc_set_num_threads(B.usable_cores())
o = B.Oracle(ctl, clim, mets[0], mets[1], atm)
o.timesteps_init()
s = hip.Simulation(ctl, clim, mets[0], mets[1], atm)
s.timesteps_init(0.0, 0.0)
s.prefetch_met(mets[2])
imet, t0 = 0, time.time()
times = cases.step_times(o.ctl)
pending = []          # (batched) steps the device has not been given yet
for k, t in enumerate(times):
    if t > mets[imet + 1].time:
        if pending:
            s.run_timesteps(pending[0], len(pending))
            pending = []
        imet += 1
        o.swap_met(mets[imet + 1])
        s.commit_met()
        if imet + 2 < len(mets):
            s.prefetch_met(mets[imet + 2])
    o.run_timestep(t)
    if batched:
        pending.append(t)
        if len(pending) == 20 or k % 97 == 0 or k == len(times) - 1 or (k + 1 < len(times) and times[k + 1] - t != o.ctl.dt_mod):
            s.run_timesteps(pending[0], len(pending))
            pending = []
    else:
        s.run_timestep(t)
    if k % 97 == 0:
        g, r = s.state(), o.state()
        print(k, "t = %.0f s" % t, {kk: "%.1e" % cases.rel_err(g[kk], r[kk]) for kk in ("lon", "lat", "p", "q")}, flush=True)
g, r = s.state(), o.state()
errs = {kk: cases.rel_err(g[kk], r[kk]) for kk in ("lon", "lat", "p", "q")}
print("steps", len(times), "hand-overs", imet, "final", errs, "%.0f s" % (time.time() - t0))
assert np.array_equal(g["time"], r["time"]) and all(v <= 1e-10 for v in errs.values())
assert s.get_cache()["rng_ctr"] == o.cache.rng_ctr
print("SOAK OK")
