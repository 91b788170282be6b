"""CPU tests of the host-side logic: synthetic inputs, control defaults,
sharding arithmetic, oracle self-consistency and drift against the committed
oracle vectors."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
from mptrac_amd import ctl as ctlmod
from mptrac_amd import hip
from mptrac_amd.clim import load_clim_tropo
from mptrac_amd.synth import lcg_uniform, synthetic_met, synthetic_particles
from oracle import binding as B

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lcg_jump_ahead_equals_the_scalar_recurrence():
    s = 12345
    want = []
    for _ in range(1000):
        s = (s * 6364136223846793005 + 1442695040888963407) % 2 ** 64
        want.append((s >> 11) / 2.0 ** 53)
    assert np.array_equal(lcg_uniform(1000, 12345), np.array(want))


def test_synthetic_met_layout():
    m = synthetic_met("tiny", 0.0)
    assert m.lon[0] == -180.0 and m.lon[-1] == 180.0 and m.nx == 37
    assert m.p[0] == 1013.25 and np.all(np.diff(m.p) < 0)
    for a in list(m.f3.values()) + list(m.f2.values()):
        assert a.dtype == np.float32 and a.flags.c_contiguous
        assert np.array_equal(a[-1], a[0])          # periodic column
    assert np.all(m.f2["cl"] > 0)


def test_ctl_defaults_follow_the_reference():
    c = ctlmod.fill_ctl(hip.MphipCtl())
    assert (c.advect, c.rng_type, c.diffusion, c.dt_mod, c.dt_met, c.direction) == (2, 1, 0, 180.0, 3600.0, 1)
    assert (c.turb_dx_pbl, c.turb_dx_trop, c.turb_dz_strat, c.turb_mesox) == (50.0, 50.0, 0.1, 0.16)
    assert (c.conv_cape, c.sort_dt, c.mixing_trop, c.t_stop) == (-999.0, -999.0, -999.0, 1e100)
    assert (c.mixing_nx, c.mixing_ny, c.mixing_nz, c.grid_nx, c.grid_ny, c.grid_nz) == (360, 180, 90, 360, 180, 1)
    assert tuple(c.wet_depo_pre) == (0.5, 0.36) and c.dry_depo_dp == 30.0
    with pytest.raises(KeyError):
        ctlmod.fill_ctl(hip.MphipCtl(), no_such_key=1)


def test_shard_ranges_partition_the_index_space():
    for n in (0, 1, 7, 10 ** 7 + 3):
        for w in (1, 2, 3, 8):
            r = [hip.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))


def test_step_times_follow_the_driver_loop():
    o = B.Oracle(dict(cases.BASE), load_clim_tropo(), synthetic_met("tiny", 0.0), synthetic_met("tiny", 3600.0),
                 synthetic_particles(4))
    o.timesteps_init()
    ts = cases.step_times(o.ctl)
    assert len(ts) == 21 and ts[0] == 0.0 and ts[-1] == 3600.0      # 20 steps + the dt = 0 first call
    o.ctl.t_stop = 3500.0
    assert cases.step_times(o.ctl)[-1] == 3500.0                    # last step clamped (trac.c:136-137)


def test_oracle_is_thread_count_independent():
    """The Squares stream is counter-based: 1 thread and all threads agree bit
    for bit (the property the reference's tests rely on)."""
    code = ("import sys; sys.path[:0]=[%r, %r]; import cases, numpy as np; from oracle import binding as B\n"
            "c,cl,m0,m1,a = cases.make_case('conv_sedi', n=3000, grid='tiny')\n"
            "o = B.Oracle(c,cl,m0,m1,a); o.timesteps_init()\n"
            "[o.run_timestep(t) for t in cases.step_times(o.ctl)[:6]]\n"
            "print(repr(float(o.lon.sum())), repr(float(o.p.sum())), repr(float(o.uvwp.sum())))\n"
            ) % (os.path.dirname(GOLD), os.path.dirname(os.path.dirname(GOLD)))
    outs = []
    for nt in ("1", "4"):
        env = dict(os.environ, OMP_NUM_THREADS=nt)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env).decode())
    assert outs[0] == outs[1] and "nan" not in outs[0]


def test_oracle_sort_is_stable_and_a_permutation():
    ctl, clim, m0, m1, atm = cases.make_case("full", n=5000, grid="tiny")
    o = B.Oracle(ctl, clim, m0, m1, atm)
    before = o.lon.copy()
    keys, perm = o.sort()
    ks = keys[perm]
    assert np.all(np.diff(ks) >= 0) and sorted(perm) == list(range(5000))
    ties = ks[1:] == ks[:-1]
    assert ties.any() and np.all(perm[1:][ties] > perm[:-1][ties])
    assert np.array_equal(o.lon, before[perm])


def test_oracle_scheduler_equals_module_sequence():
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=2000, grid="tiny")
    a = B.Oracle(ctl, clim, m0, m1, atm)
    b = B.Oracle(ctl, clim, m0, m1, atm)
    a.timesteps_init()
    b.timesteps_init()
    for t in cases.step_times(a.ctl)[:4]:
        a.run_timestep(t)
        for m in ("timesteps", "position", "advect", "diff_turb", "diff_meso", "convection", "sedi", "position"):
            b.module(m, t)
    for k, v in a.state().items():
        assert np.array_equal(v, b.state()[k]), k


def test_grid_sums_are_additive_over_index_shards():
    """What the RCCL all-reduce relies on: per-shard sums add up to the global
    sums (counts exactly)."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=6000, grid="tiny")
    o = B.Oracle(ctl, clim, m0, m1, atm)
    cnt, mean, sig = o.grid_sums(0.0)
    acc_c, acc_m = 0, 0
    for r in range(3):
        lo, hi = hip.shard_range(6000, r, 3)
        sub = {k: (v[lo:hi] if k != "q" else v[:, lo:hi]) for k, v in atm.items()}
        c, m, _ = B.Oracle(ctl, clim, m0, m1, sub).grid_sums(0.0)
        acc_c, acc_m = acc_c + c, acc_m + m
    assert cnt.sum() == 6000 and np.array_equal(acc_c, cnt)
    assert cases.rel_err(acc_m, mean) < 1e-13


@pytest.mark.parametrize("case", list(cases.CASES))
def test_oracle_reproduces_committed_vectors(case):
    """Drift check: tests/golden/oracle_<case>.npz were written by
    tests/make_golden.py from this oracle; the GPU suite checks the HIP path
    against the same files."""
    path = os.path.join(GOLD, f"oracle_{case}.npz")
    if not os.path.exists(path):
        pytest.skip("vector not generated")
    import make_golden
    ref = np.load(path)
    got = make_golden.run_case(case)
    for k in ref.files:
        assert np.array_equal(got[k], ref[k], equal_nan=got[k].dtype.kind == "f"), (case, k)


def test_oracle_isosurface_modes_conserve_their_quantity():
    """module_isosurf (mptrac.c:4956-5005): after every step the particle is back on its surface --
    pressure (mode 1), density p / T (mode 2), potential temperature (mode 3)."""
    import ctypes as C
    from mptrac_amd.synth import FIELDS_3D
    for case, mode in (("isosurf_p", 1), ("isosurf_rho", 2), ("isosurf_theta", 3)):
        ctl, clim, m0, m1, atm = cases.make_case(case, n=400)
        ctl["sort_dt"] = -999.0      # (module_sort leaves cache->iso_var with the slot, not the particle)
        o = B.Oracle(ctl, clim, m0, m1, atm)
        o.timesteps_init()
        p0 = o.p.copy()
        for t in cases.step_times(o.ctl)[:6]:
            o.run_timestep(t)
        if mode == 1:
            assert np.array_equal(o.p, p0)
            continue
        v = C.c_double()
        for ip in range(0, o.n, 7):
            o.lib.orc_intpol_met_time_3d(C.byref(o.met[0]), C.byref(o.met[1]), FIELDS_3D.index("t"), o.time[ip], o.p[ip],
                                         o.lon[ip], o.lat[ip], C.byref(v))
            now = o.p[ip] / v.value if mode == 2 else o.lib.orc_theta(o.p[ip], v.value)
            # the temperature used to restore p was taken at the pressure before the restore: first order only
            assert abs(now - o.iso_var[ip]) <= 5e-2 * abs(o.iso_var[ip]), (case, ip, now, o.iso_var[ip])


def test_oracle_boundary_conditions_only_touch_the_region():
    """module_bound_cond (mptrac.c:3789-3881) as a single call: mass / vmr with trend and age of air inside
    the latitude-pressure window and surface layer, everything else untouched."""
    ctl, clim, m0, m1, atm = cases.make_case("bound", n=3000)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    o.module("timesteps", 180.0)
    q0 = o.q.copy()
    o.module("bound_cond")
    names = cases.CASE_QUANTITIES["bound"]
    im, iv, ia = names.index("m"), names.index("vmr"), names.index("aoa")
    hit = o.q[ia] != q0[ia]                      # aoa = time marks the particles that were set (time = 0 -> all 0!)
    inside = (o.lat >= -60) & (o.lat <= 60) & (o.p <= 1100) & (o.p >= 300)
    changed = o.q[im] != q0[im]
    assert changed.any() and not changed[~inside].any()
    assert np.all(o.q[im][changed] == 2.5 + 1e-4 * o.time[changed])
    assert np.all(o.q[iv][changed] == 3e-9 + 1e-13 * o.time[changed])
    assert not hit[~changed].any()
    for k in range(len(names)):
        if k not in (im, iv, ia):
            assert np.array_equal(o.q[k], q0[k])


def test_host_layer_calendar_functions_match_the_reference_time_table():
    """time2jsec / jsec2time of the host layer (they name the meteo and output files) against the reference's
    tests/tools_test/data.ref/time.tab: 135 dates between 1900 and 2100, incl. hour 24 and day 31 of short
    months (tests/tools_test/run.sh:13-25); the day-of-year lines of the table belong to tools outside this
    repository's scope and are skipped."""
    import ctypes as C
    from mptrac_amd import build
    lib, _ = build.build_host()
    L = C.CDLL(lib)
    L.time2jsec.argtypes = [C.c_int] * 6 + [C.c_double, C.POINTER(C.c_double)]
    L.jsec2time.argtypes = [C.c_double] + [C.POINTER(C.c_int)] * 6 + [C.POINTER(C.c_double)]
    lines = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tools_test", "time.tab")).read().splitlines()
    k = 0
    for year in (1900, 1980, 2000, 2020, 2100):
        for mon in (1, 7, 12):
            for day in (1, 15, 31):
                for hour in (0, 12, 24):
                    t = C.c_double()
                    L.time2jsec(year, mon, day, hour, 0, 0, 0.0, C.byref(t))
                    v = [C.c_int() for _ in range(6)]
                    r = C.c_double()
                    L.jsec2time(t.value, *[C.byref(x) for x in v], C.byref(r))
                    got = "%d %d %d %d %d %d %g = %.2f" % (*[x.value for x in v], r.value, t.value)
                    assert got == lines[k], (got, lines[k])
                    k += 2
    assert k == len(lines) == 270
