"""Pin the CPU oracle against everything the reference's own tests hold for
the hot path: the `sedi` known-answer table, the dd_test
golden trajectories, and the micro-KATs captured from the compiled reference
during the survey.  CPU only."""
import ctypes as C
import json
import os
import re

import numpy as np

import refcases
from mptrac_amd.clim import load_clim_tropo
from mptrac_amd.synth import synthetic_met, synthetic_particles
from oracle import binding as B

GOLD = refcases.GOLD


def test_sedi_matches_reference_tools_test_table():
    """tests/tools_test/data.ref/sedi.tab: 144 cases printed with %g."""
    txt = open(os.path.join(GOLD, "ref_tools_test", "sedi.tab")).read()
    blocks = [b for b in txt.split("\n\n") if "v_s=" in b]
    assert len(blocks) == 144
    L = B.lib()
    for b in blocks:
        val = {k.strip(): v.split()[0] for k, v in (ln.split("=") for ln in b.strip().splitlines())}
        vs = L.orc_sedi(float(val["p"]), float(val["T"]), float(val["r_p"]), float(val["rho_p"]))
        assert "%g" % vs == val["v_s"], (val, vs)


def _survey():
    return json.load(open(os.path.join(GOLD, "survey_kats.json")))


def test_survey_kats_scalar_helpers():
    k = _survey()
    L = B.lib()
    for p, T, rp, rhop, want in k["sedi"]:
        assert L.orc_sedi(p, T, rp, rhop) == want
    xx = np.array(k["locate_axis"], dtype=np.float64)
    yy = np.ascontiguousarray(xx[::-1])
    dp = C.POINTER(C.c_double)
    for x, want in k["locate_irr_asc"]:
        assert L.orc_locate_irr(xx.ctypes.data_as(dp), len(xx), x) == want
    for x, want in k["locate_irr_desc"]:
        assert L.orc_locate_irr(yy.ctypes.data_as(dp), len(yy), x) == want
    for x, want in k["locate_reg_asc"]:
        assert L.orc_locate_reg(xx.ctypes.data_as(dp), len(xx), x) == want


def _tiny_oracle(**ctl):
    m0 = synthetic_met("tiny", 0.0, 1.0)
    m1 = synthetic_met("tiny", 3600.0, 1.2)
    return B.Oracle(ctl, load_clim_tropo(), m0, m1, synthetic_particles(8))


def test_survey_kats_clim_tropo():
    k = _survey()
    o = _tiny_oracle()
    for t, lat, want in k["clim_tropo"]:
        assert o.lib.orc_clim_tropo(C.byref(o.clim), t, lat) == want


def test_survey_kats_squares_rng():
    k = _survey()
    o = _tiny_oracle()
    got = [o.lib.orc_squares(i) / 2.0 ** 64 for i in range(7)]
    assert got == k["squares_uniform_ctr0_6"]
    # the same through module_rng (uniform): ctr 0..6 consumed by n = 6 (+1)
    o.lib.orc_module_rng(C.byref(o.ctl), C.byref(o.cache), 6, 0)
    assert list(o.rs[:7]) == k["squares_uniform_ctr0_6"] and o.cache.rng_ctr == 7
    # next call, normal: ctr 7..13, Box-Muller on flat pairs
    o.lib.orc_module_rng(C.byref(o.ctl), C.byref(o.cache), 6, 1)
    assert list(o.rs[:6]) == k["squares_normal_ctr7_n6"] and o.cache.rng_ctr == 14


def test_dd_test_golden_trajectories():
    """Midpoint advection through the wind-tool field reproduces the 6-digit
    golden positions of tests/dd_test at every hourly output."""
    ctl, atm, t0, gold = refcases.dd_test_case()
    mets = [refcases.wind_tool_met(t0 + 3600.0 * i) for i in range(8)]
    o = B.Oracle(ctl, load_clim_tropo(), mets[0], mets[1], atm)
    o.timesteps_init()
    assert o.ctl.t_start == t0
    imet = 0
    t = t0
    nbad = 0
    worst = 0.0
    while True:
        # mptrac_get_met: advance the bracketing pair when t passes met1
        while t > o.met[1].time:
            imet += 1
            o.swap_met(mets[imet + 1])
        o.run_timestep(t)
        hours = (t - t0) / 3600.0
        if hours == int(hours):
            g = gold[int(hours)]
            idx = o.q[0].astype(int)
            order = np.argsort(idx)
            assert np.array_equal(idx[order], g[:, 4].astype(int))
            for col, arr in ((2, o.lon), (3, o.lat)):
                for a, b in zip(arr[order], g[:, col]):
                    if refcases.fmt_g(a) != b:
                        nbad += 1
                        worst = max(worst, abs(a - b))
            z = 7.0 * np.log(1013.25 / o.p[order])
            assert all(refcases.fmt_g(a) == b for a, b in zip(z, g[:, 1]))
            assert np.all(o.time == t)
        if t >= ctl["t_stop"]:
            break
        t += ctl["dt_mod"]
    assert nbad == 0, (nbad, worst)


def test_humidity_macros_match_reference_met_sample_table():
    """tests/met_test/data.ref/sample_ei_2011_06_05_00.tab (met_sample tool): RH, RHICE, TDEW, TICE
    recomputed from the printed p, T, H2O columns.  Inputs carry six digits, so the bar is what that
    allows (1e-3 on the humidities, 1e-4 on the temperatures), tight enough to catch a wrong constant
    or formula of PW / PSAT / PSICE / TDEW / TICE (mptrac.h:1808-2102)."""
    a = np.loadtxt(os.path.join(GOLD, "ref_met_test", "sample_thermo.tab"))
    assert a.shape == (3000, 7)
    L = B.lib()
    for p, t, h2o, rh, rhice, tdew, tice in a:
        assert abs(L.orc_rh(p, t, h2o) - rh) <= 1e-3 * abs(rh)
        assert abs(L.orc_rhice(p, t, h2o) - rhice) <= 1e-3 * abs(rhice)
        assert abs(L.orc_tdew(p, h2o) - tdew) <= 1e-4 * tdew
        assert abs(L.orc_tice(p, h2o) - tice) <= 1e-4 * tice


def test_module_meteo_is_the_interpolation_plus_the_macros():
    """orc_module_meteo against the separately pinned pieces: orc_intpol_met_time_3d/2d at the particle
    and the scalar helpers."""
    from mptrac_amd.ctl import ctl_from_quantities
    from mptrac_amd.synth import FIELDS_2D, FIELDS_3D, FIELDS_METEO_ONLY
    import cases
    names = ("t", "h2o", "ps", "theta", "zeta_d", "rh", "tice", "lapse", "sst", "p", "pv", "vh", "u", "v")
    fields = cases.PRESSURE_LEVEL_FIELDS + FIELDS_METEO_ONLY
    m0 = synthetic_met("tiny", 0.0, 1.0, fields=fields)
    m1 = synthetic_met("tiny", 3600.0, 1.3, fields=fields)
    atm = synthetic_particles(300, quantities=names, time=900.0)
    o = B.Oracle(dict(ctl_from_quantities(names)), load_clim_tropo(), m0, m1, atm)
    o.module("meteo")
    q = dict(zip(names, o.q))
    L = o.lib
    v = C.c_double()
    for ip in range(o.n):
        tm, p, lon, lat = o.time[ip], o.p[ip], o.lon[ip], o.lat[ip]
        L.orc_intpol_met_time_3d(C.byref(o.met[0]), C.byref(o.met[1]), FIELDS_3D.index("t"), tm, p, lon, lat, C.byref(v))
        t = v.value
        L.orc_intpol_met_time_3d(C.byref(o.met[0]), C.byref(o.met[1]), FIELDS_3D.index("h2o"), tm, p, lon, lat, C.byref(v))
        h2o = v.value
        L.orc_intpol_met_time_2d(C.byref(o.met[0]), C.byref(o.met[1]), FIELDS_2D.index("ps"), tm, lon, lat, C.byref(v))
        ps = v.value
        L.orc_intpol_met_time_2d(C.byref(o.met[0]), C.byref(o.met[1]), FIELDS_2D.index("sst"), tm, lon, lat, C.byref(v))
        sst = v.value
        assert (q["t"][ip], q["h2o"][ip], q["ps"][ip], q["p"][ip]) == (t, h2o, ps, p)
        assert q["sst"][ip] == sst or (np.isnan(sst) and np.isnan(q["sst"][ip]))
        assert q["theta"][ip] == L.orc_theta(p, t)
        assert q["zeta_d"][ip] == L.orc_zeta(ps, p, t)
        assert q["rh"][ip] == L.orc_rh(p, t, h2o)
        assert q["tice"][ip] == L.orc_tice(p, h2o)
        assert q["lapse"][ip] == L.orc_lapse_rate(t, h2o)
        assert q["vh"][ip] == np.sqrt(q["u"][ip] ** 2 + q["v"][ip] ** 2)
    assert np.isnan(q["sst"]).any() and np.isfinite(q["sst"]).any()


def test_coord_test_golden_of_the_reference():
    """The reference's tests/coord_test end to end: its netCDF meteo files (UTM grid), 1000 parcels, two hours of
    midpoint advection + turbulent + mesoscale diffusion (Squares generator, single-precision sine / cosine),
    the meteo hand-over after one hour and module_meteo's t, u, v, w -- the oracle reproduces every printed digit
    of the reference's own golden particle files."""
    import ref_coord as R
    from mptrac_amd.ctl import ctl_from_quantities
    mets = [R.load_met(h) for h in range(3)]
    ctl = dict(R.CTL, **ctl_from_quantities(R.QUANTITIES))
    o = B.Oracle(ctl, load_clim_tropo(), mets[0], mets[1], R.initial_particles())
    o.timesteps_init()
    worst = R.run_against_golden(o, mets)
    assert worst["x"] <= R.TOL_XY and worst["y"] <= R.TOL_XY, worst
    assert worst["z"] <= R.TOL_REL and worst["q"] <= R.TOL_REL, worst


def test_nat_temperature_matches_reference_tools_test_table():
    """tests/tools_test of the reference (run.sh:39-47): 36 calls of its `tnat` tool -- dew point, frost point and
    NAT existence temperature at (p, H2O, HNO3); every printed digit."""
    L = B.lib()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tools_test", "tnat.tab")
    vals = {}
    seen = 0
    for ln in open(path):
        if "=" not in ln:
            continue
        key, rest = ln.split("=")
        vals[key.strip()] = rest.split()[0]
        if key.strip() == "T_NAT":
            p, h2o, hno3 = float(vals["p"]), float(vals["q_H2O"]), float(vals["q_HNO3"])
            assert "%g" % L.orc_tdew(p, h2o) == vals["T_dew"]
            assert "%g" % L.orc_tice(p, h2o) == vals["T_ice"]
            assert "%g" % L.orc_nat_temperature(p, h2o, hno3) == vals["T_NAT"]
            seen += 1
    assert seen == 36


def test_zonal_mean_climatology_against_an_independent_restatement():
    """orc_clim_zm on the reference's HNO3 climatology (gap-filled as read_clim_zm does) and on a synthetic table,
    against the numpy statement of the same interpolation in tests/refclim.py -- inside the table, beyond its
    edges (clamped), across the turn of the year's last node."""
    import refclim
    L = B.lib()
    rng = np.random.default_rng(11)
    for table in (refclim.load_zonal_mean(), refclim.synthetic_zonal_mean(3)):
        time, p, lat, vmr = (np.ascontiguousarray(a) for a in table)
        assert (vmr >= 0).all()
        z = B.OrcZm()
        z.ntime, z.np, z.nlat = len(time), len(p), len(lat)
        z.time, z.p, z.lat, z.vmr = (a.ctypes.data_as(C.POINTER(C.c_double)) for a in (time, p, lat, vmr))
        worst = 0.0
        for _ in range(4000):
            t = rng.uniform(-4e7, 9e8)
            la = rng.uniform(-95.0, 95.0)
            pr = float(np.exp(rng.uniform(np.log(0.05), np.log(1100.0))))
            want = refclim.clim_zm(table, t, la, pr)
            got = L.orc_clim_zm(C.byref(z), t, la, pr)
            worst = max(worst, abs(got - want) / max(abs(want), 1e-30))
        assert worst <= 1e-12, worst


def test_trace_gas_time_series_interpolation():
    """orc_clim_ts (clim_ts, mptrac.c:394-410) on the reference's SF6 series against numpy's clamped linear
    interpolation: inside, at the nodes, before the first and after the last entry."""
    L = B.lib()
    raw = np.loadtxt(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data", "noaa_gml_sf6.tab"))
    time = np.ascontiguousarray((raw[:, 0] - 2000.0) * 365.25 * 86400.0)
    vmr = np.ascontiguousarray(raw[:, 1])
    ts = B.OrcTs()
    ts.ntime = len(time)
    ts.time, ts.vmr = (a.ctypes.data_as(C.POINTER(C.c_double)) for a in (time, vmr))
    rng = np.random.default_rng(2)
    probes = np.concatenate([rng.uniform(time[0] - 1e8, time[-1] + 1e8, 3000), time[::7], [time[0], time[-1]]])
    got = np.array([L.orc_clim_ts(C.byref(ts), float(t)) for t in probes])
    want = np.interp(probes, time, vmr)
    assert np.allclose(got, want, rtol=1e-13, atol=0)
    assert got[-2] == vmr[0] and got[-1] == vmr[-1]
