"""Randomised module combinations against the oracle (GPU): every seed draws a control set (integrator,
stochastic modules, convection, sedimentation, sort, mixing, decay, wet / dry deposition, boundary conditions,
isosurface mode, meteo quantities (climatology ones included), trace gases with surface time series, direction, grid orientation, vertical coordinate of the advection) and runs 12 steps through
mphip_run_timestep (even seeds) or mphip_run_timesteps in pieces of one to five steps (odd seeds).  Catches interactions between modules and between the kernel instantiations that the
named cases do not cover."""
import os

import numpy as np
import pytest

import cases
from mptrac_amd import build as _build
from mptrac_amd import hip
from mptrac_amd.ctl import ctl_from_quantities
from mptrac_amd.synth import FIELDS_METEO_ONLY, FIELDS_ML, synthetic_met, synthetic_particles
from oracle import binding as B

pytestmark = pytest.mark.gpu


def draw(seed):
    r = np.random.default_rng(seed)
    pick = lambda *a: a[int(r.integers(len(a)))]
    names = ["m", "vmr", "rp", "rhop", "loss_rate", "mloss_decay", "mloss_wet", "mloss_dry", "aoa"]
    ctl = dict(advect=pick(1, 2, 4, 4), dt_mod=pick(180.0, 600.0), dt_met=7200.0, rng_type=1, direction=pick(1, 1, -1))
    if r.random() < 0.7:
        ctl.update(diffusion=1, turb_dz_trop=pick(0.0, 0.1), turb_dx_trop=pick(0.0, 50.0), turb_dz_pbl=pick(0.0, 0.5),
                   turb_mesox=pick(0.0, 0.16), turb_mesoz=pick(0.0, 0.16), turb_pbl_trans=pick(0.0, 0.2))
        if r.random() < 0.25:
            ctl.update(turb_pbl_scheme=1, turb_mesoz=0.0)
    if r.random() < 0.6:
        ctl.update(conv_cape=pick(0.0, 150.0), conv_cin=pick(-999.0, 5.0), conv_mix_pbl=pick(0, 1), conv_pbl_trans=pick(0.0, 0.1),
                   conv_dt=pick(-999.0, 360.0))
    if r.random() < 0.4:
        ctl.update(sort_dt=pick(360.0, 1200.0))
    if r.random() < 0.4:
        ctl.update(mixing_trop=1e-3, mixing_strat=1e-6, mixing_dt=pick(360.0, 1200.0), mixing_nx=36, mixing_ny=18, mixing_nz=20)
    if r.random() < 0.5:
        ctl.update(tdec_trop=259200.0, tdec_strat=432000.0)
    if r.random() < 0.4:
        ctl.update(wet_depo_ic_a=1e-4, wet_depo_ic_b=0.8, wet_depo_bc_a=5e-5, wet_depo_bc_b=0.6)
    if r.random() < 0.4:
        ctl.update(dry_depo_vdep=0.15)
    if r.random() < 0.3:
        ctl.update(bound_lat0=-60.0, bound_lat1=60.0, bound_p0=1100.0, bound_p1=200.0, bound_mass=2.0, bound_vmr=1e-9,
                   bound_dps=pick(-999.0, 300.0), bound_pbl=pick(0, 1))
    if r.random() < 0.25:
        ctl.update(isosurf=pick(1, 2, 3))
    sedi = r.random() < 0.6
    if r.random() < 0.4:
        extra = list(r.choice(["t", "u", "ps", "pv", "theta", "rh", "zg", "sst", "pbl", "cape", "lwc", "h2o", "o3c", "vh"],
                              size=4, replace=False))
        names += extra
        ctl.update(met_dt_out=pick(0.1, 1200.0))
    vert = pick(0, 0, 0, 1, 2, 3)         # winds from the model levels: zeta, pressure, eta coordinate
    if vert:
        names += ["zeta", "eta"]
        ctl.pop("isosurf", None)
    # a second stream for what came later (the draws above stay what they were for every seed): trace gases of
    # module_bound_cond / module_mixing with and without a surface time series, climatology quantities of module_meteo
    r2 = np.random.default_rng(seed + 1000003)
    tables = {}
    if r2.random() < 0.35:
        gases = list(r2.choice(["Cccl4", "Cccl3f", "Cccl2f2", "Cn2o", "Csf6"], size=int(r2.integers(1, 4)), replace=False))
        for gname in gases[:max(0, 16 - len(names))]:
            names.append(gname)
            if r2.random() < 0.75:
                tt = np.sort(r2.uniform(-8000.0, 8000.0, int(r2.integers(2, 40))))
                tables[{"Cccl4": "ccl4", "Cccl3f": "ccl3f", "Cccl2f2": "ccl2f2", "Cn2o": "n2o", "Csf6": "sf6"}[gname]] = \
                    (tt + np.arange(len(tt)) * 1e-3, r2.uniform(1e-12, 4e-10, len(tt)))
    if r2.random() < 0.3 and len(names) <= 12:
        import refclim
        want = list(r2.choice(["hno3", "oh", "ho2", "tnat"], size=2, replace=False))
        if "tnat" in want and len(names) <= 10 and r2.random() < 0.5:
            want += ["tice", "tsts"]
        for q in want:
            if q not in names:
                names.append(q)
        if "met_dt_out" not in ctl:
            ctl.update(met_dt_out=pick(0.1, 1200.0))
        tables["hno3"] = refclim.load_zonal_mean()
        tables["oh"] = refclim.synthetic_zonal_mean(seed, scale=1e-13)
        tables["ho2"] = refclim.synthetic_zonal_mean(seed + 1, np_=7, nlat=13, scale=1e-12)
        ctl.update(oh_chem_beta=float(r2.choice([0.0, 0.6])))
    # (round 5, drawn last so that every earlier draw of a seed stays what it was) module_sort / module_mixing in EVERY
    # step: the sort that runs ahead then repairs the previous order
    if r2.random() < 0.3:
        if "sort_dt" in ctl:
            ctl["sort_dt"] = ctl["dt_mod"]
        if "mixing_dt" in ctl and r2.random() < 0.5:
            ctl["mixing_dt"] = ctl["dt_mod"]
    ctl["_tables"] = tables
    ctl.update(ctl_from_quantities(names), advect_vert_coord=vert)
    if not sedi:
        ctl["qnt_rp"] = ctl["qnt_rhop"] = -1
    n_steps = 12
    ctl["t_stop"] = ctl["direction"] * n_steps * ctl["dt_mod"]
    geom = dict(grid=pick("tiny", "C1"), lon0=pick(-180.0, 0.0), lat_reverse=bool(pick(0, 1)))
    return ctl, tuple(names), geom


def _seeds():
    """48 seeds in the suite; MPTRAC_FUZZ_SEEDS=first:last runs a longer campaign (tools: a few minutes of GPU)."""
    import os
    span = os.environ.get("MPTRAC_FUZZ_SEEDS")
    if span:
        a, b = (int(x) for x in span.split(":"))
        only = os.environ.get("MPTRAC_FUZZ_ONLY")       # e.g. turb_pbl_scheme: the seeds whose draw sets that parameter
        if only:
            return [s for s in range(a, b) if draw(s)[0].get(only)]
        return range(a, b)
    return range(48)


def _contexts(seed):
    """The drawn case as an oracle and a device context at their initial state."""
    ctl, names, geom = draw(seed)
    tables = ctl.pop("_tables")
    fields = cases.PRESSURE_LEVEL_FIELDS + FIELDS_METEO_ONLY
    if ctl["advect_vert_coord"]:
        fields = fields + FIELDS_ML
        geom["lat_reverse"] = False
    t0, t1 = (0.0, 7200.0) if ctl["direction"] == 1 else (-7200.0, 0.0)
    m0 = synthetic_met(geom["grid"], t0, 1.0, fields=fields, lon0=geom["lon0"], lat_reverse=geom["lat_reverse"])
    m1 = synthetic_met(geom["grid"], t1, 1.25, fields=fields, lon0=geom["lon0"], lat_reverse=geom["lat_reverse"])
    # (MPTRAC_FUZZ_PARTICLES: campaigns with more particles per seed -- more tiles per sort, fuller mixing boxes)
    n = int(os.environ.get("MPTRAC_FUZZ_PARTICLES", "6000"))
    atm = synthetic_particles(n, seed=100 + seed, quantities=names, lon=(geom["lon0"], geom["lon0"] + 360.0))
    for nq in ("zeta", "eta"):
        if nq in names:      # a vertical coordinate inside the range of the synthetic zetal field
            atm["q"][list(names).index(nq)] = 320.0 + 1680.0 * ((atm["lat"] + 85.0) / 170.0)
    if ctl.get("turb_pbl_scheme", 0):
        atm["p"][::2] = 1013.25 * np.exp(-(0.02 + 0.9 * (atm["lon"][::2] - geom["lon0"]) / 360.0) / 7.0)
    clim = cases.load_clim_tropo() + (tables,)
    for k, name in enumerate(names):
        if name.startswith("C"):     # trace gases: mixing ratios of their own
            atm["q"][k] = np.random.default_rng(seed + k).uniform(1e-12, 4e-10, len(atm["time"]))
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(atm["time"].min(), atm["time"].max())
    return ctl, names, o, s, cases.step_times(o.ctl)


def _first_divergence(seed):
    """For the message of a failing seed: the first step at which the single-precision perturbations cache->uvwp
    differ, in float ulps, and where the positions stood then.  (cache->uvwp is what the reference stores in single
    precision, mptrac.h:3265; a last-bit difference of the double in front of that rounding flips the float with
    probability 2^-29, and the positions inherit one float ulp of a wind x DT_MOD -- DESIGN.md section 4.)"""
    _, _, o, s, times = _contexts(seed)
    try:
        for k, t in enumerate(times):
            o.run_timestep(t)
            s.run_timestep(t)
            g, r = s.state(), o.state()
            d = np.abs(g["uvwp"].astype(np.float64) - r["uvwp"])
            if d.max() > 0:
                i, c = np.unravel_index(int(np.argmax(d)), d.shape)
                ulps = float(d[i, c] / np.spacing(np.float32(max(abs(float(r["uvwp"][i, c])), 1e-30))))
                pos = {q: cases.rel_err(g[q], r[q]) for q in ("lon", "lat", "p")}
                return (f"first difference of cache->uvwp: step {k} of {len(times)}, particle {i}, component {c}, "
                        f"{ulps:.1f} float ulp ({int((d > 0).sum())} values differ); positions at that step {pos}")
        return "cache->uvwp identical in every step (one step at a time)"
    finally:
        s.close()


@pytest.mark.parametrize("seed", _seeds())
def test_random_module_combination(seed):
    ctl, names, o, s, times = _contexts(seed)
    for t in times:
        o.run_timestep(t)
    if seed % 2 == 0:
        for t in times:
            s.run_timestep(t)
    else:
        # odd seeds: the same steps handed over in pieces of one to five (mphip_run_timesteps -- pieces with
        # nothing scheduled inside share a kernel launch; whatever the module set, the result is the loop's)
        rr = np.random.default_rng(7000 + seed)
        stride = o.ctl.direction * o.ctl.dt_mod
        i = 0
        while i < len(times):
            k = 1
            want = int(rr.integers(1, 6))
            while k < want and i + k < len(times) and times[i + k] == times[i + k - 1] + stride:
                k += 1
            s.run_timesteps(times[i], k)
            i += k
    g, r = s.state(), o.state()
    ctr = s.get_cache()["rng_ctr"]
    s.close()
    assert np.array_equal(g["time"], r["time"])
    worst = {k: cases.rel_err(g[k], r[k]) for k in ("lon", "lat", "p")}
    err, row = cases.q_rows_err(o.ctl, g["q"], r["q"])      # every quantity row on its own scale
    worst[names[row] if len(names) else "q"] = err
    if _build.exact_requested():
        # MPTRAC_AMD_EXACT=1, the reference-rounding build: positions and perturbations are the oracle's bits; the
        # quantities too, but for sums whose order the device does not fix (module_mixing's atomic cell sums)
        bits = {k: int(np.count_nonzero(g[k] != r[k])) for k in ("lon", "lat", "p", "uvwp")}
        if any(bits.values()) or max(worst.values()) > 1e-13:
            pytest.fail(f"seed {seed} (reference-rounding build): values that are not the oracle's bits {bits}, {worst}; {ctl}")
    if max(worst.values()) > 1e-10:
        pytest.fail(f"seed {seed}: {worst} (bar 1e-10); {_first_divergence(seed)}; {ctl}")
    assert cases.rel_err(g["uvwp"], r["uvwp"]) <= 1e-6
    assert ctr == o.cache.rng_ctr
