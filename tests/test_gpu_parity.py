"""GPU parity tests: the HIP path (through the C ABI in include/mptrac_hip.h)
against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): indices and sort order bit-exact; positions
and quantities within 1e-10 relative, |d| / max(|ref|, 1), under the fixed
Squares stream; the random numbers and the single-precision perturbations
cache->uvwp bit-exact.  The tolerances are written at each assert; measured errors
are ~1e-15 (reciprocal-multiply weights and fused multiply-adds of the device code;
exp / log / pow are the C library's bits since round 6).
"""
import ctypes as C

import numpy as np
import pytest

import cases
from mptrac_amd import hip
from mptrac_amd.ctl import ctl_from_quantities
from mptrac_amd.synth import synthetic_met, synthetic_particles
from oracle import binding as B

pytestmark = pytest.mark.gpu

TOL = 1e-10          # north_star tolerance for positions / quantities


def _pair(name, n=10000, grid="C1", **kw):
    ctl, clim, m0, m1, atm = cases.make_case(name, n=n, grid=grid, **kw)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(atm["time"].min(), atm["time"].max())
    assert s.ctl.t_start == o.ctl.t_start and s.ctl.t_stop == o.ctl.t_stop
    cases.prepare(o)
    cases.prepare(s)
    return o, s


def _compare(o, s, tol=TOL):
    g, r = s.state(), o.state()
    assert np.array_equal(g["time"], r["time"])
    for k in ("lon", "lat", "p"):
        assert cases.rel_err(g[k], r[k]) <= tol, (k, cases.rel_err(g[k], r[k]))
    if r["q"].size:
        # every quantity row on its own scale (vmr ~ 3e-9, loss_rate ~ 4e-6 ... are far below 1)
        err, row = cases.q_rows_err(o.ctl, g["q"], r["q"])
        assert err <= tol, ("q", row, err)
    assert np.array_equal(g["uvwp"], r["uvwp"])      # cache->uvwp (single precision): the oracle's bits
    assert s.get_cache()["rng_ctr"] == o.cache.rng_ctr


# ---------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------

def test_sincosf_is_bit_identical_to_libm():
    """The Box-Muller angle goes through single-precision cosf/sinf
    (mptrac.c:5824-5825); the device restatement must give glibc's bits."""
    _, s = _pair("advect", n=16)
    L = B.lib()
    fp = C.POINTER(C.c_float)
    two_pi_bits = 0x40c90fdb
    # every 61st float in [0, 2 pi], and one dense window around pi/4 and 2 pi
    for first, count, stride in ((0, two_pi_bits // 61, 61), (0x3f490000, 1 << 16, 1), (two_pi_bits - 70000, 70000, 1)):
        bits = (first + stride * np.arange(count, dtype=np.uint64)).astype(np.uint32)
        x = bits.view(np.float32)
        c_ref = np.empty_like(x)
        s_ref = np.empty_like(x)
        L.orc_libm_sincosf(x.ctypes.data_as(fp), C.c_size_t(len(x)), c_ref.ctypes.data_as(fp), s_ref.ctypes.data_as(fp))
        if stride == 1:
            c_dev, s_dev = s.test_sincosf(int(first), int(count))
        else:
            # strided sample: evaluate the covering range in chunks and pick
            c_dev = np.empty_like(x)
            s_dev = np.empty_like(x)
            chunk = 1 << 24
            lo = 0
            while lo < len(bits):
                hi = min(len(bits), lo + chunk // stride)
                b0, b1 = int(bits[lo]), int(bits[hi - 1])
                cc, ss = s.test_sincosf(b0, b1 - b0 + 1)
                c_dev[lo:hi] = cc[::stride]
                s_dev[lo:hi] = ss[::stride]
                lo = hi
        assert np.array_equal(c_dev.view(np.uint32), c_ref.view(np.uint32))
        assert np.array_equal(s_dev.view(np.uint32), s_ref.view(np.uint32))
    s.close()


@pytest.mark.parametrize("ctr,n", [(0, 7), (7, 6), (123456789012, 30001), (2 ** 40 + 3, 4096), (987654321, 3000000)])
def test_rng_stream_matches_module_rng(ctr, n):
    o, s = _pair("advect", n=max(n // 3 + 1, 16))
    for method in (0, 1, 2):      # 2: the normals as the modules draw them (a particle's triple at once: libm_log_unit_pair)
        o.cache.rng_ctr = ctr
        o.lib.orc_module_rng(C.byref(o.ctl), C.byref(o.cache), C.c_size_t(n), min(method, 1))
        ref = o.rs[:n].copy()
        dev = s.test_rng(ctr, n, method)
        # uniforms and normals: the reference's bits (Box-Muller through the C library's log, the IEEE square root
        # and the C library's cosf / sinf, all restated on the device: mphip_libm.h, sqrt_rn, libm_sincosf)
        assert np.array_equal(dev.view(np.uint64), ref.view(np.uint64)), method
    s.close()


def test_exp_log_pow_are_bit_identical_to_libm():
    """exp, log and pow of the kernels (csrc/mphip_libm.h) against the C library the oracle links, the one the
    reference's CPU build calls (src/mptrac.c:4531-4546, 5822-5825 and every Z(), P(), THETA() macro): identical bits
    on > 10^8 arguments -- wide ranges, the ranges the kernels use, every branch of the algorithms, special values --
    with the tables read from device memory and from the LDS copy; and the square root of the Box-Muller radius
    against sqrt()."""
    import libm_args
    _, s = _pair("advect", n=16)
    L = B.lib()
    dp = C.POINTER(C.c_double)

    def ref(op, x, y=None):
        out = np.empty_like(x)
        L.orc_libm_f64(op, x.ctypes.data_as(dp), y.ctypes.data_as(dp) if y is not None else None, C.c_size_t(len(x)),
                       out.ctypes.data_as(dp))
        return out

    rng = np.random.default_rng(20260930)
    n = 5_000_000
    total = 0
    for fn, op, sets in (("exp", 0, libm_args.exp_sets), ("log", 1, libm_args.log_sets)):
        for k, (name, x) in enumerate(sets(rng, n)):
            x = np.ascontiguousarray(x, dtype=np.float64)
            same = libm_args.same_bits(s.test_libm(fn, x, lds=bool(k & 1)), ref(op, x))
            assert same.all(), (fn, name, int((~same).sum()), float(x[~same][0]).hex())
            total += len(x)
    for k, (name, (x, y)) in enumerate(libm_args.pow_sets(rng, n)):
        x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        same = libm_args.same_bits(s.test_libm("pow", x, y, lds=bool(k & 1)), ref(2, x, y))
        assert same.all(), ("pow", name, int((~same).sum()), float(x[~same][0]).hex(), float(y[~same][0]).hex())
        total += len(x)
    assert total > 1e8, total
    # the square root of sqrt(-2 log u) and sqrt(1 - r^2): IEEE on its documented domain
    for x in (-2.0 * np.log(rng.uniform(0.0, 1.0, n)), rng.uniform(0.0, 1.0, n), 10.0 ** rng.uniform(-300.0, 300.0, n),
              np.array([0.0, -0.0, np.inf, np.nan, 1.0, 4.0, 2.0, 1e-300])):
        same = libm_args.same_bits(s.test_libm("sqrt", x), ref(3, x))
        assert same.all(), ("sqrt", int((~same).sum()), float(x[~same][0]).hex())
    # cos / sin of the C library (DX2DEG's cos(latitude) in the reference-rounding build, ZETA's sin): its bits on
    # the restated range |x| < 2.426 -- every latitude --, the device library's values beyond (not compared)
    for name, x in libm_args.sincos_sets(rng, n):
        for fn, op in (("cos", 4), ("sin", 5)):
            same = libm_args.same_bits(s.test_libm(fn, x), ref(op, x))
            assert same.all(), (fn, name, int((~same).sum()), float(x[~same][0]).hex())
    far = rng.uniform(2.5, 1e6, 1000)
    assert np.max(np.abs(s.test_libm("cos", far) - np.cos(far))) < 1e-15
    s.close()


# ---------------------------------------------------------------------------
# one module at a time, from identical inputs
# ---------------------------------------------------------------------------

MODULE_CASES = [("position", "advect"), ("advect", "advect"), ("advect", "advect_midpoint"),
                ("advect", "advect_zeta"), ("advect", "advect_eta"), ("advect", "advect_mlp"),
                ("advect", "advect_mlp_midpoint"), ("diff_pbl", "pbl"), ("diff_pbl", "pbl_meso"),
                ("advect", "advect_euler"), ("diff_turb", "turb"), ("diff_meso", "diff"),
                ("convection", "conv_sedi"), ("convection", "conv_thresh"), ("sedi", "conv_sedi"),
                ("decay", "full"), ("wet_depo", "full"), ("wet_depo", "wet_henry"), ("dry_depo", "full"),
                ("isosurf", "isosurf_rho"), ("isosurf", "isosurf_theta"), ("isosurf", "isosurf_balloon"),
                ("bound_cond", "bound"), ("bound_cond", "bound_pbl_zeta")]


@pytest.mark.parametrize("module,case", MODULE_CASES)
def test_single_module(module, case):
    o, s = _pair(case, n=5000)
    ts = cases.step_times(o.ctl)
    # a few full steps first so that the state is generic (uvwp != 0, q changed)
    for t in ts[:3]:
        o.run_timestep(t)
        s.run_timestep(t)
    t = ts[3]
    for eng in (o, s):
        eng.module("timesteps", t)
    assert np.array_equal(s.get_cache()["dt"], o.dt)
    o.module(module, t)
    s.module(module, t)
    _compare(o, s)
    s.close()


# ---------------------------------------------------------------------------
# whole time steps
# ---------------------------------------------------------------------------

TOL_DELIVERED = 2e-14   # what the named cases deliver after their steps: four orders inside the bar


@pytest.mark.parametrize("case", list(cases.CASES))
def test_run_timestep_20_steps(case):
    """Every named case against the oracle -- and, beyond the bar, how close the kernels are where no float of the
    reference flips: positions within 2e-14 relative (measured: 4e-16 .. 4e-15, i.e. 97 % of the longitudes and 84 % ..
    99 % of the pressures are the oracle's bits after 21 steps: tools/gpu_bit_census.py, profiles/r06_bit_census.txt)."""
    o, s = _pair(case, n=10000)
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
        s.run_timestep(t)
    _compare(o, s)
    g, r = s.state(), o.state()
    for k in ("lon", "lat", "p"):
        assert cases.rel_err(g[k], r[k]) <= TOL_DELIVERED, (k, cases.rel_err(g[k], r[k]))
    s.close()


def _tracer_series(t_mid):
    """Surface time series around the run (the reference's SF6 and N2O files shifted to it, a synthetic third)."""
    import os
    ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")
    out = {}
    for name, fn in (("sf6", "noaa_gml_sf6.tab"), ("n2o", "noaa_gml_n2o.tab")):
        raw = np.loadtxt(os.path.join(ref, fn))
        time = (raw[:, 0] - 2000.0) * 365.25 * 86400.0
        # one node per 100 s around the run instead of one per month: every step sees another segment
        out[name] = (t_mid + (time - time[len(time) // 2]) / 26298.0, raw[:, 1])
    out["ccl3f"] = (np.array([t_mid - 500.0, t_mid, t_mid + 700.0]), np.array([2.4e-10, 2.6e-10, 2.5e-10]))
    return out


@pytest.mark.parametrize("with_mass", [True, False], ids=["with_mass", "tracers_only"])
def test_trace_gas_boundary_conditions_and_mixing(with_mass):
    """module_bound_cond sets Csf6, Cn2o, Cccl3f from their surface time series at the particle's time inside the
    boundary region (mptrac.c:3857-3875; Cccl2f2 is carried without a series -- CLIM_CCL2F2_TIMESERIES "-" -- and
    keeps its values there) and module_mixing relaxes all of them towards the cell means (mptrac.c:5223-5230): 20
    steps of the boundary-condition case with mixing (and decay, when mass is carried) against the oracle."""
    quantities = (("m", "vmr") if with_mass else ()) + ("Csf6", "Cn2o", "Cccl3f", "Cccl2f2", "aoa")
    ctl, clim, m0, m1, atm = cases.make_case("bound", n=10000, quantities=quantities)
    if not with_mass:      # module_decay needs mass or volume mixing ratio (mptrac.c:4235)
        ctl = dict(ctl, tdec_trop=0.0, tdec_strat=0.0)
    rng = np.random.default_rng(3)
    for k, name in enumerate(quantities):
        if name.startswith("C"):
            atm["q"][k][:] = rng.uniform(1e-12, 5e-10, len(atm["time"]))
    before = [q.copy() for q in atm["q"]]
    clim = clim + (_tracer_series(1800.0),)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(atm["time"].min(), atm["time"].max())
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
        s.run_timestep(t)
    _compare(o, s)
    r = o.state()
    for name in ("Csf6", "Cn2o", "Cccl3f", "Cccl2f2"):
        k = quantities.index(name)
        assert np.abs(r["q"][k] - before[k]).max() > 1e-13, name        # the boundary condition and / or the mixing acted
    # inside the boundary region the three gases carry values of their series, the fourth does not
    sf6 = r["q"][quantities.index("Csf6")]
    lo, hi = clim[3]["sf6"][1].min(), clim[3]["sf6"][1].max()
    assert ((sf6 >= lo) & (sf6 <= hi)).sum() > 100
    s.close()


def test_bound_cond_cfc10_quirk_of_the_reference():
    """mptrac.c:3800-3804 tests the CFC-10 index for "non-zero" instead of "absent": with Cccl4 as the only
    quantity of the module's list, the module runs if and only if it is quantity 0.  Reproduced, not repaired."""
    for quantities, acts in ((("Cccl4", "rp"), True), (("rp", "Cccl4"), False)):
        ctl, clim, m0, m1, atm = cases.make_case("bound", n=3000, quantities=quantities)
        ctl = dict(ctl, mixing_dt=0.0, tdec_trop=0.0, tdec_strat=0.0)
        k = quantities.index("Cccl4")
        atm["q"][k][:] = 7e-11
        clim = clim + ({"ccl4": (np.array([0.0, 4000.0]), np.array([9e-11, 9.5e-11]))},)
        o = B.Oracle(ctl, clim, m0, m1, atm)
        o.timesteps_init()
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.timesteps_init(atm["time"].min(), atm["time"].max())
        for t in cases.step_times(o.ctl)[:4]:
            o.run_timestep(t)
            s.run_timestep(t)
        g, r = s.state(), o.state()
        assert np.array_equal(g["q"][k], r["q"][k])
        assert (np.abs(r["q"][k] - 7e-11).max() > 1e-11) == acts
        s.close()


def test_fused_step_equals_module_sequence():
    """mphip_run_timestep (one launch) and the module-by-module sequence
    (one launch each, cache->dt handed over in memory) give identical bits."""
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=4097)
    a = hip.Simulation(ctl, clim, m0, m1, atm)
    b = hip.Simulation(ctl, clim, m0, m1, atm)
    for sim in (a, b):
        sim.timesteps_init(0.0, 0.0)
    for t in cases.step_times(a.ctl)[:4]:
        a.run_timestep(t)
        for m in ("timesteps", "position", "advect", "diff_turb", "diff_meso", "convection", "sedi", "position2"):
            b.module(m, t)
    ga, gb = a.state(), b.state()
    for k in ("time", "lon", "lat", "p", "q", "uvwp"):
        assert np.array_equal(ga[k], gb[k]), k
    assert a.get_cache()["rng_ctr"] == b.get_cache()["rng_ctr"]
    a.close()
    b.close()


@pytest.mark.parametrize("over", [dict(diffusion=0, conv_cape=-999.0, conv_mix_pbl=0, qnt_rp=-1, qnt_rhop=-1),
                                  dict(turb_mesox=0.0, turb_mesoz=0.0, conv_cape=-999.0, conv_mix_pbl=0, qnt_rp=-1, qnt_rhop=-1),
                                  dict(),
                                  dict(tdec_trop=259200.0, tdec_strat=259200.0, dry_depo_vdep=0.15, wet_depo_ic_a=1e-4,
                                       wet_depo_ic_b=0.8, wet_depo_bc_a=5e-5, wet_depo_bc_b=0.6),
                                  # subsets of an instantiation's modules, switched at run time inside it: convection
                                  # without sedimentation (a gas tracer), sedimentation alone, mesoscale diffusion alone
                                  dict(qnt_rp=-1, qnt_rhop=-1),
                                  dict(diffusion=0, conv_cape=-999.0, conv_mix_pbl=0),
                                  dict(diffusion=0),
                                  dict(turb_dx_trop=0.0, turb_dx_pbl=0.0, turb_dz_trop=0.0, turb_dz_strat=0.0, turb_dz_pbl=0.0,
                                       conv_cape=-999.0, conv_mix_pbl=0, qnt_rp=-1, qnt_rhop=-1)],
                         ids=["advect", "advect_turb", "c3_set", "c3_set_decay_deposition", "no_sedimentation",
                              "sedimentation_only", "convection_sedimentation", "mesoscale_only"])
@pytest.mark.parametrize("advect", [4, 2, 1], ids=["rk4", "midpoint", "euler"])
def test_lean_instantiations_equal_the_general_code(over, advect):
    """The specialised (lean) instantiations of the step kernel -- straight-line stencil set-up, packed corner
    differences, one reciprocal per latitude; four Runge-Kutta stages, or the two of the midpoint scheme (the
    reference's default) of which the Euler scheme runs the first -- and the general instantiation compute the
    same bits."""
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=6001)
    ctl.update(over, advect=advect)
    runs = []
    for generic in (0, 1):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("generic_kernel", generic)
        s.timesteps_init(0.0, 0.0)
        for t in cases.step_times(s.ctl)[:6]:
            s.run_timestep(t)
        runs.append(s.state())
        s.close()
    for k in ("time", "lon", "lat", "p", "q", "uvwp"):
        assert np.array_equal(runs[0][k], runs[1][k]), k


@pytest.mark.parametrize("case", ["pbl", "pbl_meso"])
@pytest.mark.parametrize("advect", [4, 2], ids=["rk4", "midpoint"])
def test_lean_boundary_layer_closure_equals_the_general_code(case, advect):
    """module_diff_pbl inside the gated lean instantiations (kPblClosure) and inside the general kernel: same bits;
    and the closure moved particles (half of the case's particles start inside the boundary layer)."""
    ctl, clim, m0, m1, atm = cases.make_case(case, n=6001)
    ctl.update(advect=advect)
    runs = []
    for generic in (0, 1):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("generic_kernel", generic)
        s.timesteps_init(0.0, 0.0)
        for t in cases.step_times(s.ctl)[:6]:
            s.run_timestep(t)
        runs.append(s.state())
        s.close()
    for k in ("time", "lon", "lat", "p", "q", "uvwp"):
        assert np.array_equal(runs[0][k], runs[1][k]), k
    assert np.count_nonzero(runs[0]["uvwp"][:, 2]) > 1000


@pytest.mark.parametrize("n", [20011, 300000])
def test_module_sort_by_order_repair_equals_the_sort_from_scratch(n):
    """module_sort in every step: the sort that runs ahead repairs the order of the previous one (stayers / movers /
    merge) -- same keys and permutation as the stable sort from scratch, hence the same bits everywhere (random numbers
    follow the slots); checked against the run without the repair, against the oracle, and the last sort's keys and
    permutation against the oracle's (ties by index)."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=n)
    ctl.update(sort_dt=180.0, mixing_dt=180.0)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    runs = []
    times = cases.step_times(o.ctl)[:9]
    for repair in (1, 0):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("sort_repair", repair)
        s.timesteps_init(0.0, 0.0)
        for t in times:
            s.run_timestep(t)
        runs.append(s.state())
        runs[-1]["ctr"] = s.get_cache()["rng_ctr"]
        if repair:
            keys, perm = s.get_sort()
        s.close()
    for k in ("time", "lon", "lat", "p", "q", "uvwp"):
        assert np.array_equal(runs[0][k], runs[1][k]), k
    for t in times[:-1]:
        o.run_timestep(t)
    # the sort of the last step on the oracle: keys / permutation of module_sort at that time, then the rest of the step
    ko, po = o.sort()
    assert np.array_equal(keys, ko[po]) and np.array_equal(perm, po)


@pytest.mark.parametrize("advect", [4, 2, 1], ids=["rk4", "midpoint", "euler"])
@pytest.mark.parametrize("tile", [1024, 96])
def test_lds_tile_trajectories_equal_the_launches_without_a_tile(advect, tile):
    """Option lds_tile (the wind grid staged through an LDS tile per workgroup, traj_tile_kernel): runs of
    pure trajectory steps read their corner records from the tile where the stencil lies inside it and from global memory
    otherwise -- the bits of the launches without a tile, and the oracle's positions; with particles across the date
    line and at the poles, some of them released later, a tile far too small for a workgroup's box (96 cells), and the
    internal re-sort in between."""
    ctl, clim, m0, m1, atm = cases.make_case("advect", n=30011, fields=("u", "v", "w", "ps"), quantities=("m",))
    ctl.update(advect=advect)
    atm["lon"][:2000] = np.linspace(-180.0, 179.99, 2000)
    atm["lat"][:1000] = 89.95
    atm["lat"][1000:2000] = -89.95
    atm["time"][::11] = 540.0
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    times = cases.step_times(o.ctl)[:14]
    runs = []
    for cells in (0, tile):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("lds_tile", cells)
        s.set_option("locality_sort_interval", 5)
        s.timesteps_init(atm["time"].min(), atm["time"].max())
        s.run_timestep(times[0])
        s.synchronize()
        s.profile_begin()
        s.run_timesteps(times[1], 4)
        launches, _ = s.profile_end()
        assert launches == 1
        s.run_timesteps(times[5], 9)
        runs.append(s.state())
        s.close()
    for k in ("time", "lon", "lat", "p", "q"):
        assert np.array_equal(runs[0][k], runs[1][k]), k
    for t in times:
        o.run_timestep(t)
    r = o.state()
    assert np.array_equal(runs[1]["time"], r["time"])
    for k in ("lon", "lat", "p"):
        assert cases.rel_err(runs[1][k], r[k]) <= TOL, k


def test_met_swap_over_two_intervals():
    """mptrac_get_met's pointer swap (mptrac.c:6486-6499): 2 h with 3 snapshots."""
    ctl, clim, m0, m1, atm = cases.make_case("diff", n=3000)
    ctl["t_stop"] = 7200.0
    m2 = synthetic_met("C1", 7200.0, 0.8, fields=cases.PRESSURE_LEVEL_FIELDS)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    swapped = False
    for t in cases.step_times(o.ctl):
        if t > 3600.0 and not swapped:
            o.swap_met(m2)
            s.swap_met(m2)
            swapped = True
        o.run_timestep(t)
        s.run_timestep(t)
    assert swapped
    _compare(o, s)
    s.close()


@pytest.mark.parametrize("prefetch", [False, True])
def test_axes_follow_met0_across_a_handover(prefetch):
    """The reference tolerates axes that differ by up to 1e-3 between meteo files (mptrac.c:6543-6556) and always
    interpolates and sorts on the axes of the current met0: after a hand-over those are the old met1's.  Three
    snapshots whose latitude / pressure axes differ by a few 1e-4."""
    from mptrac_amd.synth import Met
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=3000)
    ctl.update(t_stop=7200.0, sort_dt=720.0)

    def shifted(m, dlat, fp):
        return Met(m.time, m.lon, m.lat + dlat, m.p * fp, m.f3, m.f2)
    m2 = synthetic_met("C1", 7200.0, 0.8, fields=cases.PRESSURE_LEVEL_FIELDS)
    m1, m2 = shifted(m1, 4e-4, 1.0 + 3e-7), shifted(m2, -3e-4, 1.0 - 2e-7)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    if prefetch:
        s.prefetch_met(m2)
    swapped = False
    for t in cases.step_times(o.ctl):
        if t > 3600.0 and not swapped:
            o.swap_met(m2)
            if prefetch:
                s.commit_met()
            else:
                s.swap_met(m2)
            swapped = True
        o.run_timestep(t)
        s.run_timestep(t)
    assert swapped
    _compare(o, s)
    s.close()


@pytest.mark.parametrize("case", ["diff", "full", "zeta_full", "bound_pbl_zeta"])
def test_met_prefetch_and_commit_equal_the_synchronous_swap(case):
    """mphip_prefetch_met / mphip_commit_met (upload of the next snapshot on a copy stream beside the
    time steps) against the oracle's pointer swap over 3 h with 4 snapshots: the prefetch is issued right
    after each hand-over, the staging arrays rotate twice."""
    ctl, clim, m0, m1, atm = cases.make_case(case, n=3000)
    ctl["t_stop"] = 10800.0
    fields = tuple(m0.f3) + tuple(m0.f2)
    m2 = synthetic_met("C1", 7200.0, 0.8, fields=fields)
    m3 = synthetic_met("C1", 10800.0, 1.1, fields=fields)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    cases.prepare(o)
    cases.prepare(s)
    with pytest.raises(RuntimeError):
        s.commit_met()                     # nothing prefetched yet
    s.prefetch_met(m2)
    with pytest.raises(RuntimeError):
        s.prefetch_met(m3)                 # one snapshot at a time
    upcoming = [(3600.0, m2, m3), (7200.0, m3, None)]
    for t in cases.step_times(o.ctl):
        if upcoming and t > upcoming[0][0]:
            _, new1, after = upcoming.pop(0)
            o.swap_met(new1)
            s.commit_met()
            if after is not None:
                s.prefetch_met(after)
        o.run_timestep(t)
        s.run_timestep(t)
    assert not upcoming
    _compare(o, s)
    s.close()


def test_backward_trajectories():
    ctl, clim, m0, m1, atm = cases.make_case("turb", n=2000)
    ctl.update(direction=-1, t_stop=0.0)
    atm["time"][:] = 3600.0
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(3600.0, 3600.0)
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
        s.run_timestep(t)
    _compare(o, s)
    assert np.all(o.time == 0.0)
    s.close()


# ---------------------------------------------------------------------------
# sort, mixing, grid sums
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("lon0,n", [(-180.0, 20000), (0.0, 20000), (-180.0, 1), (-180.0, 4097)])
def test_sort_keys_and_order_bit_exact(lon0, n):
    """module_sort: keys, permutation (ties by original index) and permuted
    arrays identical.  lon0 = 0 pins the reference's raw-longitude rule on a
    0...360 grid (all lon < 0 land in ix = 0)."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=n, lon0=lon0)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    keys_o, perm_o = o.sort()
    keys_s, perm_s = s.sort()
    assert np.array_equal(np.sort(keys_o), keys_s)      # device returns the sorted keys
    assert np.array_equal(perm_o, perm_s)
    assert np.all(np.diff(keys_s) >= 0)
    g, r = s.state(), o.state()
    for k in ("time", "lon", "lat", "p", "q"):
        assert np.array_equal(g[k], r[k]), k
    if lon0 == 0.0 and n > 1:
        ix = (keys_s // (m0.ny * m0.np)).astype(int)
        assert (ix == 0).sum() >= (r["lon"] < 0).sum()
    s.close()


@pytest.mark.parametrize("bits", [8, 9, 10])
def test_sort_digit_widths(bits):
    """The radix sort picks the digit width with the fewest passes for the key range of the grid (8, 9 or
    10 bits); every width must give the oracle's permutation (ties by original index)."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=30011)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.set_option("sort_bits", bits)
    keys_o, perm_o = o.sort()
    keys_s, perm_s = s.sort()
    assert np.array_equal(np.sort(keys_o), keys_s) and np.array_equal(perm_o, perm_s)
    g, r = s.state(), o.state()
    for k in ("time", "lon", "lat", "p", "q"):
        assert np.array_equal(g[k], r[k]), k
    s.close()


def _oracle_takes_device_state(o, s):
    """Both sides continue from the same bits (the device's)."""
    g = s.state()
    for k in ("time", "p", "lon", "lat"):
        getattr(o, k)[:] = g[k]
    o.q[:] = g["q"]
    return g


@pytest.mark.parametrize("mode", [1, 2, 0], ids=["ordered", "ordered_chains", "atomics"])
def test_mixing_and_grid_sums(mode):
    """module_mixing and the gridded-output sums from identical inputs.  deterministic_sums 1 (default) adds every
    cell's summands in the reference's order of the particle index: every bit equals the serial code's;
    0 = floating-point atomics, order of arrival."""
    o, s = _pair("full", n=20000)
    s.set_option("deterministic_sums", min(mode, 1))
    s.set_option("sum_path", 2 if mode == 2 else 0)
    ts = cases.step_times(o.ctl)
    for t in ts[:2]:
        o.run_timestep(t)
        s.run_timestep(t)
    t = ts[1]                      # particle times equal the last step's t
    _oracle_takes_device_state(o, s)
    o.module("mixing", t)
    s.module("mixing", t)
    g, r = s.state(), o.state()
    co, mo, so = o.grid_sums(o.time[0])
    cs, ms, ss = s.grid_sums(o.time[0])
    assert co.sum() > 0 and np.array_equal(co, cs)       # counts: exact
    if mode:
        assert np.array_equal(g["q"], r["q"])
        assert np.array_equal(ms, mo) and np.array_equal(ss, so)
    else:
        err, row = cases.q_rows_err(o.ctl, g["q"], r["q"])
        assert err <= 1e-12, (row, err)
        assert cases.rel_err(ms, mo) <= 1e-12 and cases.rel_err(ss, so) <= 1e-12
    s.close()


def test_ordered_cell_sums_with_long_lists():
    """~100 particles per cell of the default mixing grid plus 30000 in one cell and 3000 in another (cells beyond
    64 particles are walked by the whole wave), stored in the locality order (the sequence in external order comes
    from the permutation, many short runs), with and without an ensemble: bit-identical to the serial sums and
    from run to run."""
    from mptrac_amd.ctl import ctl_from_quantities
    names = cases.QUANTITIES + ("aoa", "ens")
    n = 120000
    for nens in (0, 3):
        ctl = {k: v for k, v in cases.CASES["full"].items() if k not in ("mixing_nx", "mixing_ny", "mixing_nz")}
        ctl.update(mixing_dt=180.0, nens=nens, grid_nx=36, grid_ny=18, grid_nz=10, **ctl_from_quantities(names))
        m0 = synthetic_met("C1", 0.0, 1.0, fields=cases.PRESSURE_LEVEL_FIELDS)
        m1 = synthetic_met("C1", 3600.0, 1.25, fields=cases.PRESSURE_LEVEL_FIELDS)
        atm = _crowded(n, names)
        rng = np.random.default_rng(5)
        big = rng.permutation(n)[:33000]
        atm["lon"][big[:30000]] = 3.5 + 0.2 * rng.random(30000)
        atm["lat"][big[:30000]] = 4.5 + 0.2 * rng.random(30000)
        atm["p"][big[:30000]] = atm["p"][big[0]] * (1.0 + 1e-3 * rng.random(30000))
        atm["lon"][big[30000:]] = 7.5 + 0.2 * rng.random(3000)
        atm["lat"][big[30000:]] = 2.5 + 0.2 * rng.random(3000)
        atm["p"][big[30000:]] = atm["p"][big[30000]] * (1.0 + 1e-3 * rng.random(3000))
        atm["q"][names.index("ens")] = (np.arange(n) * 5 % max(nens, 1)).astype(np.float64)
        clim = cases.load_clim_tropo()
        o = B.Oracle(ctl, clim, m0, m1, atm)
        o.timesteps_init()
        results = []
        for rep in range(2):
            s = hip.Simulation(ctl, clim, m0, m1, atm)
            s.set_option("locality_sort_interval", 1)
            s.timesteps_init(0.0, 0.0)
            ts = cases.step_times(s.ctl)
            for t in ts[:2]:
                s.run_timestep(t)          # (the second step runs module_mixing inside)
            if rep == 0:
                o.dt[:] = s.get_cache()["dt"]
                _oracle_takes_device_state(o, s)
                o.module("mixing", ts[1])
            s.module("mixing", ts[1])
            g = s.state()
            results.append((g["q"], *s.grid_sums(ts[1])))
            if rep == 0:
                r = o.state()
                im = names.index("m")
                assert np.abs(r["q"][im] - atm["q"][im]).max() > 1e-4
                assert np.array_equal(g["q"], r["q"])
                co, mo, so = o.grid_sums(ts[1])
                assert co.max() > 30000
                assert np.array_equal(co, results[0][1]) and np.array_equal(mo, results[0][2])
                assert np.array_equal(so, results[0][3])
            s.close()
        for x, y in zip(*results):
            assert np.array_equal(x, y)


GRID_SHAPES = [
    # nx, ny, nz, lon0, lon1, lat0, lat1, z0, z1
    (1, 1, 1, -180.0, 180.0, -90.0, 90.0, 0.0, 40.0),          # one cell holds everything
    (360, 180, 1, -180.0, 180.0, -90.0, 90.0, 0.0, 40.0),      # the default 2-D output grid
    (7, 5, 200, -30.0, 40.0, -20.0, 35.0, 2.0, 22.0),          # columns longer than a cell group, most particles outside
    (37, 19, 129, -180.0, 180.0, -90.0, 90.0, -5.0, 35.0),     # a column is one cell more than a group
    (3, 2, 64, 10.0, 10.5, 20.0, 20.2, 5.0, 6.0),              # (almost) nobody inside
]


@pytest.mark.parametrize("shape", GRID_SHAPES, ids=["1x1x1", "360x180x1", "7x5x200", "37x19x129", "tiny_window"])
@pytest.mark.parametrize("order", ["external", "locality"])
@pytest.mark.parametrize("path", [1, 2], ids=["groups", "chains"])
def test_ordered_grid_sums_on_odd_grids(shape, order, path):
    """Gridded-output sums (counts, sums of q and of q^2) against the serial sums, bit for bit, on grids that
    stress the grouping of the ordered sums: a single cell, one level, columns longer than a group of cells, a
    window that holds few or no particles -- with the particles stored in the caller's order and in the internal
    locality order (external index = a permutation), through both algorithms of the ordered sums (a wave per
    group of cells; a lane per (cell, value) chain of the list sorted by cell -- normally chosen by crowding)."""
    nx, ny, nz, lon0, lon1, lat0, lat1, z0, z1 = shape
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=30011)
    ctl = dict(ctl, grid_nx=nx, grid_ny=ny, grid_nz=nz, grid_lon0=lon0, grid_lon1=lon1, grid_lat0=lat0,
               grid_lat1=lat1, grid_z0=z0, grid_z1=z1)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.set_option("locality_sort_interval", 1 if order == "locality" else 0)
    s.set_option("sum_path", path)
    s.timesteps_init(0.0, 0.0)
    ts = cases.step_times(s.ctl)
    for t in ts[:3]:
        s.run_timestep(t)
    _oracle_takes_device_state(o, s)
    co, mo, so = o.grid_sums(ts[2])
    cs, ms, ss = s.grid_sums(ts[2])
    assert np.array_equal(co, cs)
    assert np.array_equal(mo, ms) and np.array_equal(so, ss)
    if nx * ny * nz == 1:
        assert co[0] == 30011
    s.close()


@pytest.mark.parametrize("path", [1, 2], ids=["groups", "chains"])
def test_gridded_sums_from_value_records_equal_the_sums_from_the_arrays(path):
    """The ordered gridded sums gather a particle's values from one record per particle (option grid_records,
    default on from two quantities) instead of one array per quantity: the same doubles, and the serial code's."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=30011)
    ctl = dict(ctl, grid_nx=36, grid_ny=18, grid_nz=2, grid_z0=0.0, grid_z1=30.0)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.set_option("locality_sort_interval", 1)
    s.set_option("sum_path", path)
    s.timesteps_init(0.0, 0.0)
    ts = cases.step_times(s.ctl)
    for t in ts[:3]:
        s.run_timestep(t)
    _oracle_takes_device_state(o, s)
    want = o.grid_sums(ts[2])
    for records in (1, 0, 1):
        s.set_option("grid_records", records)
        got = s.grid_sums(ts[2])
        for a, b in zip(got, want):
            assert np.array_equal(a, b), records
    s.close()


@pytest.mark.parametrize("mode", ["groups", "chains", "atomics"])
def test_gridded_sums_with_a_vertical_weighting_function(mode):
    """GRID_KERNEL: every summand of the gridded output is kernel(z) * q (and its square), the kernel linear
    between its nodes and constant beyond them (kernel_weight, mptrac.c:3298-3320, 13866).  The weight goes through
    the device's logarithm (Z(p)), so the sums agree with the serial code to rounding, not to the bit; without
    nodes the weight is exactly one again."""
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=30011)
    ctl = dict(ctl, grid_nx=36, grid_ny=18, grid_nz=4, grid_z0=0.0, grid_z1=32.0)
    kz = np.array([2.0, 5.0, 9.0, 14.0, 20.0])
    kw = np.array([0.1, 0.7, 1.0, 0.4, 0.05])
    o = B.Oracle(ctl, clim, m0, m1, atm)
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.set_option("deterministic_sums", 0 if mode == "atomics" else 1)
    s.set_option("sum_path", {"groups": 1, "chains": 2, "atomics": 0}[mode])
    s.timesteps_init(0.0, 0.0)
    ts = cases.step_times(s.ctl)
    for t in ts[:3]:
        s.run_timestep(t)
    _oracle_takes_device_state(o, s)
    plain = o.grid_sums(ts[2])
    s.set_grid_kernel(kz, kw)
    co, mo, so = o.grid_sums(ts[2], kernel=(kz, kw))
    cs, ms, ss = s.grid_sums(ts[2])
    assert np.array_equal(co, cs) and np.abs(mo - plain[1]).max() > 0.1
    scale = np.abs(mo).max(axis=1, keepdims=True)
    assert np.abs(ms - mo).max() <= 1e-12 * scale.max() and np.abs(ss - so).max() <= 1e-12 * np.abs(so).max()
    with pytest.raises(hip.MphipError):
        s.set_grid_kernel([3.0, 1.0], [1.0, 1.0])        # heights must ascend
    s.set_grid_kernel()                                   # off again: the serial code's bits (ordered sums)
    cs, ms, ss = s.grid_sums(ts[2])
    if mode != "atomics":
        assert np.array_equal(ms, plain[1]) and np.array_equal(ss, plain[2])
    s.close()


@pytest.mark.parametrize("path", [0, 1, 2], ids=["auto", "groups", "chains"])
@pytest.mark.parametrize("shape", [(4, 2, 1), (36, 18, 5)], ids=["crowded", "sparse"])
def test_gridded_counts_without_quantities(shape, path):
    """NQ = 0: the gridded output is the particle count per cell alone (the reference's write_grid supports
    it, mptrac.c:13815-13872) -- through every algorithm of the ordered sums, which size their work by the
    number of values per cell (none here)."""
    ctl, clim, m0, m1, atm = cases.make_case("advect", n=30011, quantities=())
    nx, ny, nz = shape
    ctl = dict(ctl, grid_nx=nx, grid_ny=ny, grid_nz=nz, grid_z0=0.0, grid_z1=30.0)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    assert s.nq == 0
    s.set_option("sum_path", path)
    s.timesteps_init(0.0, 0.0)
    ts = cases.step_times(s.ctl)
    for t in ts[:3]:
        s.run_timestep(t)
    _oracle_takes_device_state(o, s)
    co = o.grid_sums(ts[2])[0]
    cs = s.grid_sums(ts[2])[0]
    assert co.sum() > 20000 and np.array_equal(co, cs)
    s.close()


def test_one_quantity_handed_back_in_the_callers_order():
    """mphip_update_quantity: one quantity array, in the caller's order, replaces the device's -- with the
    particles stored in the caller's order and in the internal locality order (the array then goes through the
    permutation); everything else is untouched, and the next steps use the new values (the mass enters decay,
    mixing and deposition): the oracle, given the same array, stays in step."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=20011)
    for interval in (0, 1):
        o = B.Oracle(ctl, clim, m0, m1, atm)
        o.timesteps_init()
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("locality_sort_interval", interval)
        s.timesteps_init(0.0, 0.0)
        ts = cases.step_times(s.ctl)
        for t in ts[:3]:
            o.run_timestep(t)
            s.run_timestep(t)
        before = s.state()
        im = list(cases.QUANTITIES).index("m")
        new_m = 2.0 + np.arange(s.n) * 1e-3
        s.update_quantity(im, new_m)
        o.q[im][:] = new_m
        after = s.state()
        assert np.array_equal(after["q"][im], new_m)
        for key in ("time", "lon", "lat", "p", "uvwp"):
            assert np.array_equal(after[key], before[key]), key
        rest = [iq for iq in range(s.nq) if iq != im]
        assert np.array_equal(after["q"][rest], before["q"][rest])
        for t in ts[3:7]:
            o.run_timestep(t)
            s.run_timestep(t)
        _compare(o, s)
        with pytest.raises(hip.MphipError):
            s.update_quantity(99, new_m)
        s.close()


def test_context_reused_with_other_particle_counts():
    """One context, particle sets of 3000 -> 1500 -> 2250 -> 3000 -> 4500 particles with module_sort (and its sort that
    runs ahead, whose buffers trade places with the context's when a prepared sort is adopted, as two buffers of the
    order repair do) every step: every phase gives the bits of a fresh context started from the same inputs and
    random-number counter.  (1500 -> 2250 is the step at which a capacity remembered from 3000 would let the repair
    write 2250 pairs into buffers of 1500.)"""
    ctl, clim, m0, m1, _ = cases.make_case("full", n=10)
    ctl = dict(ctl, sort_dt=180.0, mixing_dt=180.0)
    sets = [cases.make_case("full", n=n, seed=100 + n)[4] for n in (3000, 1500, 2250, 3000, 4500)]
    s = hip.Simulation(ctl, clim, m0, m1, sets[0])
    s.timesteps_init(0.0, 0.0)
    ts = cases.step_times(s.ctl)
    for k, atm in enumerate(sets):
        if k:
            s.replace_particles(atm)
        ctr = s.get_cache()["rng_ctr"]
        f = hip.Simulation(ctl, clim, m0, m1, atm, rng_ctr=ctr)
        f.timesteps_init(0.0, 0.0)
        for t in ts[:6]:
            s.run_timestep(t)
            f.run_timestep(t)
        a, b = s.state(), f.state()
        for key in ("time", "lon", "lat", "p", "q", "uvwp"):
            assert np.array_equal(a[key], b[key]), (k, key)
        f.close()
    s.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 8193])
def test_ordered_sums_at_tile_and_wave_boundaries(n):
    """Particle counts around the wave (64) and tile (4096) sizes of the run compaction and of the sort, in both
    storage orders: module_mixing and the gridded sums against the serial code, bit for bit."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=n)
    ctl = dict(ctl, mixing_dt=180.0)
    for interval in (0, 1):
        o = B.Oracle(ctl, clim, m0, m1, atm)
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("locality_sort_interval", interval)
        s.timesteps_init(0.0, 0.0)
        ts = cases.step_times(s.ctl)
        for t in ts[:2]:
            s.run_timestep(t)
        _oracle_takes_device_state(o, s)
        o.module("mixing", ts[1])
        s.module("mixing", ts[1])
        g, r = s.state(), o.state()
        assert np.array_equal(g["q"], r["q"])
        co, mo, so = o.grid_sums(ts[1])
        cs, ms, ss = s.grid_sums(ts[1])
        assert np.array_equal(co, cs) and np.array_equal(mo, ms) and np.array_equal(so, ss)
        s.close()


# ---------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------

def test_empty_and_single_particle():
    for n in (0, 1, 63, 257):
        ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=max(n, 1))
        if n == 0:
            atm = {k: (v[:0] if k != "q" else v[:, :0]) for k, v in atm.items()}
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.timesteps_init(0.0, 0.0)
        for t in cases.step_times(s.ctl)[:3]:
            s.run_timestep(t)
        g = s.get_atm()
        assert len(g["lon"]) == n
        if n:
            o = B.Oracle(ctl, clim, m0, m1, atm)
            o.timesteps_init()
            for t in cases.step_times(o.ctl)[:3]:
                o.run_timestep(t)
            _compare(o, s)
        s.close()


def test_special_positions_and_times():
    """Poles, date line, above the model top, below the surface, particles not
    yet released (time > t) and already finished (time > t_stop)."""
    ctl, clim, m0, m1, atm = cases.make_case("diff", n=64)
    atm["lon"][:8] = [-180.0, 179.999999, 0.0, 359.5, -359.5, 720.25, 180.0, -180.000001]
    atm["lat"][8:16] = [90.0, -90.0, 89.9995, -89.9995, 91.0, -93.5, 270.5, -271.0]
    atm["p"][16:20] = [0.05, 1200.0, 1013.25, 300.0]
    atm["time"][20:24] = [400.0, 7200.0, 180.0, -50.0]
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(atm["time"].min(), atm["time"].max())
    for t in cases.step_times(o.ctl)[:6]:
        o.run_timestep(t)
        s.run_timestep(t)
    _compare(o, s)
    s.close()


def test_regional_domain_stops_particles_outside():
    """module_timesteps zeroes dt outside a regional meteo domain
    (mptrac.c:6012-6013, 6033-6036)."""
    from mptrac_amd.synth import Met
    ctl, clim, m0, m1, atm = cases.make_case("advect", n=2000)

    def crop(m):
        i0, i1, j0, j1 = 100, 200, 60, 140
        return Met(m.time, m.lon[i0:i1], m.lat[j0:j1], m.p,
                   {k: v[i0:i1, j0:j1] for k, v in m.f3.items()}, {k: v[i0:i1, j0:j1] for k, v in m.f2.items()})
    r0, r1 = crop(m0), crop(m1)
    o = B.Oracle(ctl, clim, r0, r1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, r0, r1, atm)
    s.timesteps_init(0.0, 0.0)
    for t in cases.step_times(o.ctl)[:5]:
        o.run_timestep(t)
        s.run_timestep(t)
    moved = o.time > 0
    assert 0 < moved.sum() < len(moved)
    _compare(o, s)
    s.close()


def test_strided_meteo_upload_matches_compact():
    """met_t holds fixed-extent arrays (float u[EX][EY][EP]); uploading through
    strides must equal uploading a compact copy."""
    ctl, clim, m0, m1, atm = cases.make_case("zeta_full", n=1500, grid="tiny")
    a = hip.Simulation(ctl, clim, m0, m1, atm)
    b = hip.Simulation(ctl, clim, m0, m1, atm)
    EX, EY, EP = m0.nx + 3, m0.ny + 5, m0.npl + 4
    from mptrac_amd.synth import FIELDS_2D, FIELDS_3D, FIELDS_ML
    keep = []
    for slot, m in ((0, m0), (1, m1)):
        mm = hip.MphipMet()
        mm.time, mm.coord_type, mm.nx, mm.ny, mm.np = m.time, 0, m.nx, m.ny, m.np
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        mm.lon, mm.lat, mm.p = m.lon.ctypes.data_as(dp), m.lat.ctypes.data_as(dp), m.p.ctypes.data_as(dp)
        mm.sx, mm.sy, mm.sx2 = EY * EP, EP, EY
        mm.npl, mm.sx_ml, mm.sy_ml = m.npl, EY * EP, EP
        for i, k in enumerate(FIELDS_3D):
            if k not in m.f3:
                continue
            big = np.full((EX, EY, EP), np.nan, dtype=np.float32)
            big[:m.nx, :m.ny, :(m.npl if k in FIELDS_ML else m.np)] = m.f3[k]
            keep.append(big)
            mm.f3[i] = big.ctypes.data_as(fp)
        for i, k in enumerate(FIELDS_2D):
            if k not in m.f2:
                continue
            big = np.full((EX, EY), np.nan, dtype=np.float32)
            big[:m.nx, :m.ny] = m.f2[k]
            keep.append(big)
            mm.f2[i] = big.ctypes.data_as(fp)
        b._chk(b.L.mphip_update_met(b.h, slot, C.byref(mm)))
    for sim in (a, b):
        sim.timesteps_init(0.0, 0.0)
        for t in cases.step_times(sim.ctl)[:4]:
            sim.run_timestep(t)
    ga, gb = a.state(), b.state()
    for k in ("lon", "lat", "p", "q"):
        assert np.array_equal(ga[k], gb[k]), k
    a.close()
    b.close()


def test_missing_field_is_an_error_not_a_fallback():
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=100, fields=("u", "v", "w", "ps", "pbl"))
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    with pytest.raises(hip.MphipError):
        s.run_timestep(180.0)
    s.close()



# ---------------------------------------------------------------------------
# module_meteo
# ---------------------------------------------------------------------------

def _row_err(a, b):
    """Largest error of a quantity row relative to max(|b|, 1e-6 max|row|); NaN patterns must agree."""
    nan = np.isnan(b)
    if not np.array_equal(np.isnan(a), nan):
        return float("inf")
    a, b = np.where(nan, 0.0, a), np.where(nan, 0.0, b)
    scale = np.maximum(np.abs(b), 1e-6 * max(np.max(np.abs(b)), 1e-300))
    return float(np.max(np.abs(a - b) / scale))


def test_module_meteo_every_quantity():
    """All 60 quantities of module_meteo's SET_ATM list (mptrac.c:5091-5163), NQ_MAX at a time, at
    generic and special positions (poles, date line, outside the pressure range, NaN neighbourhoods); the
    climatology quantities from the reference's HNO3 table and synthetic tables for the other four."""
    from mptrac_amd.ctl import METEO_QUANTITIES, ctl_from_quantities
    from mptrac_amd.synth import FIELDS_METEO_ONLY
    fields = cases.PRESSURE_LEVEL_FIELDS + FIELDS_METEO_ONLY
    m0 = synthetic_met("C1", 0.0, 1.0, fields=fields)
    m1 = synthetic_met("C1", 3600.0, 1.25, fields=fields)
    import refclim
    zm = {"hno3": refclim.load_zonal_mean()}
    zm.update({name: refclim.synthetic_zonal_mean(5 + k, scale=10.0 ** -(9 + k))
               for k, name in enumerate(("oh", "h2o2", "ho2", "o1d"))})
    clim = cases.load_clim_tropo() + (zm,)
    n = 20000
    for first in range(0, len(METEO_QUANTITIES), 15):
        names = ("m",) + METEO_QUANTITIES[first:first + 15]
        atm = synthetic_particles(n, seed=99 + first, quantities=names, time=1234.5)
        atm["lon"][:8] = [-180.0, 179.999999, 0.0, 359.5, -359.5, 720.25, 180.0, -180.000001]
        atm["lat"][8:16] = [90.0, -90.0, 89.9995, -89.9995, 91.0, -93.5, 45.0, -45.0]
        atm["p"][16:20] = [0.05, 1200.0, 1013.25, 300.0]
        atm["time"][20:24] = [0.0, 3600.0, 1800.0, 5000.0]
        ctl = dict(cases.BASE, **ctl_from_quantities(names))
        o = B.Oracle(ctl, clim, m0, m1, atm)
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        o.module("meteo")
        s.module("meteo")
        g, r = s.state(), o.state()
        for k in ("time", "lon", "lat", "p"):
            assert np.array_equal(g[k], r[k])
        assert np.array_equal(g["q"][0], r["q"][0])
        for i, name in enumerate(names[1:], start=1):
            assert np.any(r["q"][i] != 0.0) or name in ("swc",), name
            assert _row_err(g["q"][i], r["q"][i]) <= 1e-11, (name, _row_err(g["q"][i], r["q"][i]))
        s.close()


@pytest.mark.parametrize("coord_type", [0, 1], ids=["latlon", "cartesian"])
def test_module_meteo_oh_with_diurnal_scaling(coord_type):
    """clim_oh (mptrac.c:89-120) with OH_CHEM_BETA > 0: the zonal mean times exp(-beta / cos(sza)) at the
    particle's longitude and latitude -- on a Cartesian grid at the reference point of the projection, where tnat
    still takes the particle's own latitude coordinate (mptrac.c:5160)."""
    import refclim
    from mptrac_amd.ctl import ctl_from_quantities
    names = ("oh", "hno3", "tnat", "tice", "tsts", "h2o")
    ctl, _, m0, m1, _ = cases.make_case("meteo", n=10)
    atm = synthetic_particles(5000, seed=4, quantities=names, time=1800.0)
    atm["time"][::3] = 360547200.0 + 3600.0 * np.arange(len(atm["time"][::3]))     # days and nights, months
    ctl = dict(cases.BASE, **ctl_from_quantities(names), oh_chem_beta=0.6, met_coord_type=coord_type,
               met_utm_ref_lat=48.15, met_utm_ref_lon=371.57)
    if coord_type == 1:
        m0.coord_type = m1.coord_type = 1
    clim = cases.load_clim_tropo() + ({"hno3": refclim.load_zonal_mean(), "oh": refclim.synthetic_zonal_mean(8, scale=1e-13)},)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    o.module("meteo")
    s.module("meteo")
    g, r = s.state(), o.state()
    oh = r["q"][0]
    assert oh.min() >= 0 and len(np.unique(np.round(oh / oh.max(), 6))) > (100 if coord_type == 0 else 5)
    for i, name in enumerate(names):
        assert _row_err(g["q"][i], r["q"][i]) <= 1e-11, (name, _row_err(g["q"][i], r["q"][i]))
    assert np.array_equal(r["q"][4], 0.5 * (r["q"][3] + r["q"][2]), equal_nan=True)
    s.close()


def test_module_meteo_climatology_rules():
    """A climatology quantity without its table and tsts without tice / tnat are refused (the second with the
    reference's message, mptrac.c:5074-5076); removing a table takes effect."""
    import refclim
    from mptrac_amd.ctl import ctl_from_quantities
    _, _, m0, m1, _ = cases.make_case("meteo", n=10)
    hno3 = refclim.load_zonal_mean()
    for names, zm, message in ((("hno3",), {}, "HNO3 climatology was not uploaded"),
                               (("tnat", "h2o"), {}, "HNO3 climatology was not uploaded"),
                               (("ho2",), {"hno3": hno3}, "HO2 climatology was not uploaded"),
                               (("tsts", "tnat", "h2o"), {"hno3": hno3}, "Need T_ice and T_NAT to calculate T_STS!")):
        atm = synthetic_particles(100, seed=1, quantities=names, time=100.0)
        s = hip.Simulation(dict(cases.BASE, **ctl_from_quantities(names)), cases.load_clim_tropo() + (zm,), m0, m1, atm)
        with pytest.raises(hip.MphipError, match=message):
            s.module("meteo")
        s.close()
    atm = synthetic_particles(100, seed=1, quantities=("hno3",), time=100.0)
    s = hip.Simulation(dict(cases.BASE, **ctl_from_quantities(("hno3",))), cases.load_clim_tropo() + ({"hno3": hno3},), m0, m1, atm)
    s.module("meteo")
    assert s.state()["q"][0].max() > 1e-10
    s.update_clim_zm("hno3")
    with pytest.raises(hip.MphipError, match="HNO3 climatology was not uploaded"):
        s.module("meteo")
    s.close()


@pytest.mark.parametrize("case", ["meteo", "meteo_gated"])
def test_deferred_module_meteo_is_not_observable(case):
    """module_meteo of a time step is launched only when its result can be seen (option lazy_meteo, on by
    default): with downloads, a gridded output, a meteo hand-over and module_sort in between, every download
    carries the bits of the run that launches it inside each step.  `meteo_gated` (MET_DT_OUT 1800 > DT_MOD)
    has steps without module_meteo, before which a pending launch must be evaluated."""
    ctl, clim, m0, m1, atm = cases.make_case(case, n=3000)
    ctl["t_stop"] = 7200.0
    m2 = synthetic_met("C1", 7200.0, 0.8, fields=tuple(m0.f3) + tuple(m0.f2))
    runs = []
    for lazy in (0, 1):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("lazy_meteo", lazy)
        s.timesteps_init(0.0, 0.0)
        seen, swapped = [], False
        for k, t in enumerate(cases.step_times(s.ctl)):
            if t > 3600.0 and not swapped:
                s.swap_met(m2)
                swapped = True
            s.run_timestep(t)
            if k in (0, 3, 4, 11, 25):
                seen.append(s.state())
            if k == 17:
                seen.append(s.grid_sums(t))
        seen.append(s.state())
        runs.append(seen)
        s.close()
    for a, b in zip(*runs):
        if isinstance(a, dict):
            for key in ("time", "lon", "lat", "p", "q"):
                assert np.array_equal(a[key], b[key], equal_nan=True), key
        else:       # gridded sums: added in the order of the particle index, the same bits every run
            cnt_a, mean_a, sig_a = a
            cnt_b, mean_b, sig_b = b
            assert np.array_equal(cnt_a, cnt_b) and cnt_a.sum() > 0
            assert np.array_equal(mean_a, mean_b, equal_nan=True)
            assert np.array_equal(sig_a, sig_b, equal_nan=True)


def test_module_meteo_missing_field_is_an_error():
    from mptrac_amd.ctl import ctl_from_quantities
    names = ("m", "t", "pv")
    m0 = synthetic_met("tiny", 0.0, 1.0, fields=cases.PRESSURE_LEVEL_FIELDS)
    m1 = synthetic_met("tiny", 3600.0, 1.25, fields=cases.PRESSURE_LEVEL_FIELDS)
    atm = synthetic_particles(100, quantities=names)
    s = hip.Simulation(dict(cases.BASE, **ctl_from_quantities(names)), cases.load_clim_tropo(), m0, m1, atm)
    with pytest.raises(hip.MphipError, match="pv"):
        s.module("meteo")
    s.close()


def test_module_meteo_in_internal_order_and_shards():
    """Quantities land in the right external slot with the locality order on, and sharded contexts
    reproduce the single-context bits."""
    ctl, clim, m0, m1, atm = cases.make_case("meteo", n=6001)
    a = hip.Simulation(ctl, clim, m0, m1, atm)
    a.set_option("locality_sort_interval", 0)
    b = hip.Simulation(ctl, clim, m0, m1, atm)
    b.set_option("locality_sort_interval", 2)
    parts = [hip.Simulation(ctl, clim, m0, m1, atm, shard=hip.shard_range(6001, r, 2)) for r in range(2)]
    sims = [a, b] + parts
    for sim in sims:
        sim.timesteps_init(0.0, 0.0)
    for t in cases.step_times(a.ctl)[:5]:
        for sim in sims:
            sim.run_timestep(t)
    ga, gb = a.state(), b.state()
    gp = [p.state() for p in parts]
    for k in ("lon", "lat", "p"):
        assert np.array_equal(ga[k], gb[k], equal_nan=True), k
        assert np.array_equal(ga[k], np.concatenate([g[k] for g in gp]), equal_nan=True), k
    assert np.array_equal(ga["q"], gb["q"], equal_nan=True)
    assert np.array_equal(ga["q"], np.concatenate([g["q"] for g in gp], axis=1), equal_nan=True)
    for sim in sims:
        sim.close()

# ---------------------------------------------------------------------------
# sharding (one process per GPU in production; two contexts on one GPU here)
# ---------------------------------------------------------------------------

def test_index_range_shards_reproduce_the_single_context_run():
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=10001)
    full = hip.Simulation(ctl, clim, m0, m1, atm)
    parts = [hip.Simulation(ctl, clim, m0, m1, atm, shard=hip.shard_range(10001, r, 3)) for r in range(3)]
    for sim in [full] + parts:
        sim.timesteps_init(0.0, 0.0)
        for t in cases.step_times(sim.ctl)[:5]:
            sim.run_timestep(t)
    g = full.state()
    for k in ("lon", "lat", "p", "uvwp"):
        joined = np.concatenate([p.state()[k] for p in parts])
        assert np.array_equal(joined, g[k]), k
    # gridded-output sums: shard sums add up to the single-context sums
    t = g["time"][0]
    cf, mf, sf = full.grid_sums(t)
    cs = sum(p.grid_sums(t)[0] for p in parts)
    ms = sum(p.grid_sums(t)[1] for p in parts)
    assert np.array_equal(cf, cs) and cases.rel_err(ms, mf) <= 1e-12
    for sim in [full] + parts:
        sim.close()


def test_rccl_communicator_single_rank():
    """The native reduction path (mphip_comm_init: ncclAllReduce of the mixing sums + 32-bit counts and of the
    gridded-output sums, on the context's stream, librccl loaded with dlopen) with a one-rank communicator --
    what one GPU can exercise of it -- against the oracle, and bit-identical to the run without a communicator."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=12000)
    ctl["mixing_dt"] = 180.0
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    runs = []
    for with_comm in (True, False):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        if with_comm:
            s.comm_init(1, 0, hip.Simulation.comm_unique_id())
        s.timesteps_init(0.0, 0.0)
        for t in cases.step_times(s.ctl)[:8]:
            s.run_timestep(t)
            if with_comm:
                o.run_timestep(t)
        if with_comm:
            _compare(o, s, tol=1e-12)
            co, mo, so = o.grid_sums(o.time[0])
            cs, ms, ss = s.grid_sums(o.time[0])
            assert np.array_equal(co, cs) and cases.rel_err(ms, mo) <= 1e-12
            s.comm_destroy()
        runs.append(s.state())
        s.close()
    for k in ("time", "lon", "lat", "p", "uvwp"):
        assert np.array_equal(runs[0][k], runs[1][k]), k
    assert np.array_equal(runs[0]["q"], runs[1]["q"])      # (ordered cell sums: the same bits every run)


# ---------------------------------------------------------------------------
# full-size properties (BASELINE sizes; no oracle run at this size)
# ---------------------------------------------------------------------------

def test_full_size_sort_properties_1e7():
    n = 10 ** 7
    ctl, clim, m0, m1, atm = cases.make_case("advect", n=n, grid="C2", fields=("u", "v", "w", "ps"),
                                             quantities=("m",))
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    keys, perm = s.sort()
    assert np.all(np.diff(keys) >= 0)                                  # sortedness
    assert np.array_equal(np.bincount(perm, minlength=n), np.ones(n, dtype=np.int64))   # a permutation
    g = s.get_atm()
    assert np.array_equal(g["lon"], atm["lon"][perm]) and np.array_equal(g["q"][0], atm["q"][0][perm])
    ties = keys[1:] == keys[:-1]
    assert np.all(perm[1:][ties] > perm[:-1][ties])                    # stable
    keys2, perm2 = s.sort()                                            # idempotent
    assert np.array_equal(keys2, keys) and np.array_equal(perm2, np.arange(n))
    s.close()


def test_full_size_subsample_against_oracle_1e6():
    """10^6 particles on the 137-level grid; advection is per-particle, so a
    random subsample run through the oracle must match the same particles of
    the full device run."""
    n = 10 ** 6
    ctl, clim, m0, m1, atm = cases.make_case("advect", n=n, grid="C2", fields=("u", "v", "w", "ps"),
                                             quantities=("m",))
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    pick = np.random.default_rng(1).choice(n, 5000, replace=False)
    sub = {k: (v[pick].copy() if k != "q" else v[:, pick].copy()) for k, v in atm.items()}
    o = B.Oracle(ctl, clim, m0, m1, sub)
    o.timesteps_init()
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
        s.run_timestep(t)
    g = s.get_atm()
    for k, ref in (("lon", o.lon), ("lat", o.lat), ("p", o.p)):
        assert cases.rel_err(g[k][pick], ref) <= TOL
    assert np.all(g["time"] == 3600.0)
    s.close()


# ---------------------------------------------------------------------------
# internal locality order (mphip_set_option "locality_sort_interval")
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("case", ["conv_sedi", "full"])
def test_locality_order_is_not_observable(case):
    """The device stores particles in grid-cell order and re-sorts every few
    steps; random numbers follow the external slot and downloads restore the
    caller's order, so results are bit-identical with the feature off, on, and
    on with downloads in between."""
    ctl, clim, m0, m1, atm = cases.make_case(case, n=6001)
    runs = {}
    for name, interval, peek in (("off", 0, False), ("every3", 3, False), ("every1_peek", 1, True)):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("locality_sort_interval", interval)
        s.timesteps_init(0.0, 0.0)
        for k, t in enumerate(cases.step_times(s.ctl)[:9]):
            s.run_timestep(t)
            if peek and k % 4 == 1:
                s.state()               # download in the middle of the run
        runs[name] = s.state()
        runs[name]["ctr"] = s.get_cache()["rng_ctr"]
        s.close()
    tol_q = 0.0        # (the mixing sums add in the order of the external index, whatever the stored order)
    for name in ("every3", "every1_peek"):
        for k in ("time", "lon", "lat", "p", "uvwp"):
            assert np.array_equal(runs[name][k], runs["off"][k]), (name, k)
        assert cases.rel_err(runs[name]["q"], runs["off"]["q"]) <= tol_q
        assert runs[name]["ctr"] == runs["off"]["ctr"]


def test_cartesian_coordinates():
    """MET_COORD_TYPE 1 (x/y in metres, e.g. UTM): DX2COORD/DY2COORD pass
    distances through, positions are clamped to the domain, the tropopause
    weight uses MET_UTM_REF_LAT (mptrac.h:966-989, mptrac.c:2782-2803, 12755)."""
    from mptrac_amd.synth import Met
    ctl, clim, g0, g1, atm = cases.make_case("conv_sedi", n=4000, grid="tiny")

    def cart(m):
        x = 600000.0 + 2000.0 * np.arange(m.nx)          # 2 km mesh
        y = 5200000.0 + 2500.0 * np.arange(m.ny)
        return Met(m.time, x, y, m.p, m.f3, m.f2, coord_type=1)
    m0, m1 = cart(g0), cart(g1)
    rng = np.random.default_rng(7)
    atm["lon"] = 600000.0 + rng.uniform(-3000.0, 2000.0 * (m0.nx - 1) + 3000.0, 4000)   # some outside: clamped
    atm["lat"] = 5200000.0 + rng.uniform(-3000.0, 2500.0 * (m0.ny - 1) + 3000.0, 4000)
    ctl.update(met_coord_type=1, met_utm_ref_lat=48.15, dt_mod=60.0, t_stop=1200.0)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
        s.run_timestep(t)
    moved = o.time > 0                       # particles outside the regional domain keep dt = 0
    assert 0 < moved.sum() < len(moved) and np.ptp(o.lon[moved]) > 1e4
    _compare(o, s)
    s.close()


def test_reference_dd_test_golden_on_the_device():
    """The reference's tests/dd_test goldens (144 parcels, six hours of midpoint advection through the
    wind tool's solid-body rotation + decay bookkeeping, seven hourly files) against the HIP back end: every
    printed digit of longitude, latitude and altitude -- the device twin of
    test_oracle_pins.py::test_dd_test_golden_trajectories."""
    import refcases
    from mptrac_amd.clim import load_clim_tropo
    ctl, atm, t0, gold = refcases.dd_test_case()
    mets = [refcases.wind_tool_met(t0 + 3600.0 * i) for i in range(8)]
    s = hip.Simulation(ctl, load_clim_tropo(), mets[0], mets[1], atm)
    s.timesteps_init(t0, t0)
    assert s.ctl.t_start == t0
    imet, t, checked = 0, t0, 0
    while True:
        if t > mets[imet + 1].time:
            imet += 1
            s.swap_met(mets[imet + 1])
        s.run_timestep(t)
        hours = (t - t0) / 3600.0
        if hours == int(hours):
            g, st = gold[int(hours)], s.state()
            idx = st["q"][0].astype(int)
            order = np.argsort(idx)
            assert np.array_equal(idx[order], g[:, 4].astype(int))
            z = 7.0 * np.log(1013.25 / st["p"][order])
            for col, arr in ((1, z), (2, st["lon"][order]), (3, st["lat"][order])):
                assert all(refcases.fmt_g(a) == b for a, b in zip(arr, g[:, col])), (hours, col)
            assert np.all(st["time"] == t)
            checked += 1
        if t >= ctl["t_stop"]:
            break
        t += ctl["dt_mod"]
    assert checked == 7
    s.close()


def test_reference_coord_test_golden_on_the_device():
    """The reference's own tests/coord_test golden files (see ref_coord.py and the oracle pin of the same
    name) against the HIP back end: every printed digit of the thirteen golden particle files, and the oracle
    within the usual bar."""
    import ref_coord as R
    from mptrac_amd.ctl import ctl_from_quantities
    mets = [R.load_met(h) for h in range(3)]
    ctl = dict(R.CTL, **ctl_from_quantities(R.QUANTITIES))
    from mptrac_amd.clim import load_clim_tropo
    clim = load_clim_tropo()
    s = hip.Simulation(ctl, clim, mets[0], mets[1], R.initial_particles())
    s.timesteps_init(R.T0, R.T0)
    worst = R.run_against_golden(s, mets)
    assert worst["x"] <= R.TOL_XY and worst["y"] <= R.TOL_XY, worst
    assert worst["z"] <= R.TOL_REL and worst["q"] <= R.TOL_REL, worst
    o = B.Oracle(ctl, clim, mets[0], mets[1], R.initial_particles())
    o.timesteps_init()
    R.run_against_golden(o, mets)
    _compare(o, s)
    s.close()


def test_full_size_stochastic_parity_1e6():
    """10^6 particles on the 137-level grid with every stochastic module on
    (random numbers depend on the global particle index, so the oracle runs the
    whole set): 6 steps against the 16-thread oracle."""
    n = 10 ** 6
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=n, grid="C2",
                                             fields=("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel"),
                                             quantities=("m", "rp", "rhop"))
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    for t in cases.step_times(o.ctl)[:7]:
        o.run_timestep(t)
        s.run_timestep(t)
    _compare(o, s)
    s.close()


def test_full_size_mixing_properties_1e7():
    """module_mixing at BASELINE configs[4]'s size -- 10^7 particles, the default 360 x 180 x 90 boxes -- with the
    particles stored in the caller's (random) order and in the internal locality order (no oracle run at this
    size; properties instead): identical bits from run to run and in both storage orders (the sums add in the
    order of the particle index, whatever the storage order); with one mixing parameter for all particles mixing
    conserves the total of every box -- hence the global total -- to rounding; the gridded output counts every
    particle once; the atomic variant agrees to 1e-12."""
    from mptrac_amd.ctl import ctl_from_quantities
    n = 10 ** 7
    names = ("m", "vmr")
    ctl = dict(cases.BASE, mixing_trop=1e-2, mixing_strat=1e-2, mixing_dt=180.0, **ctl_from_quantities(names))
    m0 = synthetic_met("C1", 0.0, 1.0, fields=cases.PRESSURE_LEVEL_FIELDS)
    m1 = synthetic_met("C1", 3600.0, 1.25, fields=cases.PRESSURE_LEVEL_FIELDS)
    atm = synthetic_particles(n, seed=3, quantities=names)
    atm["q"][1] = 1e-9 * (1.0 + np.abs(atm["lat"]) / 90.0)
    clim = cases.load_clim_tropo()
    results = {}
    for name, interval, mode in (("caller", 0, 1), ("locality", 1, 1), ("locality_again", 1, 1), ("atomics", 1, 0)):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("locality_sort_interval", interval)
        s.set_option("deterministic_sums", mode)
        s.timesteps_init(0.0, 0.0)
        s.run_timestep(0.0)          # (dt = 0: nothing moves; the particles are put into the storage order and mixed)
        s.module("mixing", 0.0)      # mixed once more, in that order
        results[name] = s.get_atm()["q"]
        if name == "caller":
            cnt, mean, _ = s.grid_sums(0.0)
            assert int(cnt.sum()) == n
            assert abs(mean[0].sum() - atm["q"][0].sum()) <= 1e-9 * atm["q"][0].sum()
        s.close()
    assert np.abs(results["caller"][0] - atm["q"][0]).max() > 1e-6      # (masses follow the longitude: boxes are nearly uniform)
    for k in range(2):
        total0, total1 = float(np.sum(atm["q"][k], dtype=np.longdouble)), float(np.sum(results["caller"][k], dtype=np.longdouble))
        assert abs(total1 - total0) <= 1e-12 * abs(total0)
    assert np.array_equal(results["caller"], results["locality"])
    assert np.array_equal(results["locality"], results["locality_again"])
    err = float(np.max(np.abs(results["atomics"] - results["caller"]) / np.abs(results["caller"])))
    assert err <= 1e-12, err


def test_sort_scales_to_1e8_keys():
    """The radix sort's two-level scan covers 10^8 particles in one context
    (288 GB of HBM hold them easily): sortedness, permutation, stability."""
    n = 10 ** 8
    m0 = synthetic_met("C1", 0.0, 1.0, fields=("u", "v", "w", "ps"))
    m1 = synthetic_met("C1", 3600.0, 1.2, fields=("u", "v", "w", "ps"))
    atm = synthetic_particles(n, quantities=())
    ctl = dict(cases.BASE, nq=0)
    s = hip.Simulation(ctl, load_clim(), m0, m1, atm)
    keys, perm = s.sort()
    assert np.all(np.diff(keys) >= 0)
    assert np.array_equal(np.bincount(perm, minlength=n), np.ones(n, dtype=np.int64))
    ties = keys[1:] == keys[:-1]
    assert np.all(perm[1:][ties] > perm[:-1][ties])
    s.close()


def load_clim():
    from mptrac_amd.clim import load_clim_tropo
    return load_clim_tropo()


@pytest.mark.parametrize("lon0,lat_reverse", [(-180.0, False), (0.0, True)])
def test_wind_cache_with_cell_changes_in_every_stage(lon0, lat_reverse):
    """Coarse grid and a long time step: most particles change their grid cell between Runge-Kutta stages
    (and before module_diff_meso), so the predicated corner reloads run in nearly every lane; also on a
    0 ... 360 longitude axis with a north-to-south latitude axis (ERA5 order)."""
    from mptrac_amd.ctl import ctl_from_quantities
    names = ("m", "rp", "rhop")
    ctl = dict(cases.BASE, dt_mod=1800.0, t_stop=5 * 1800.0, dt_met=10800.0, diffusion=1, turb_dz_trop=0.1,
               conv_cape=0.0, **ctl_from_quantities(names))
    m0 = synthetic_met("tiny", 0.0, 1.0, fields=cases.PRESSURE_LEVEL_FIELDS, lon0=lon0, lat_reverse=lat_reverse)
    m1 = synthetic_met("tiny", 10800.0, 1.3, fields=cases.PRESSURE_LEVEL_FIELDS, lon0=lon0, lat_reverse=lat_reverse)
    atm = synthetic_particles(20000, quantities=names, lon=(lon0, lon0 + 360.0))
    clim = cases.load_clim_tropo()
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
        s.run_timestep(t)
    _compare(o, s)
    s.close()


@pytest.mark.parametrize("case,field", [("zeta_full", "zetal"), ("mlp_full", "pl")])
def test_model_levels_with_a_non_monotonic_height_column(case, field):
    """The packed model-level path needs strictly monotonic zetal / pl columns; a single column that is not
    (zeta can fold near the surface in real data) sends the launch to the instantiation that repeats the
    reference's bisection read by read -- same bits as the oracle either way."""
    ctl, clim, m0, m1, atm = cases.make_case(case, n=4000)
    for m in (m0, m1):
        z = m.f3[field]
        z[100:140, 60:120, 3] = z[100:140, 60:120, 1] - 0.5      # a fold in the lowest levels of a patch
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    for t in cases.step_times(o.ctl)[:8]:
        o.run_timestep(t)
        s.run_timestep(t)
    _compare(o, s)
    s.close()


@pytest.mark.parametrize("model_levels", [23, 137])
@pytest.mark.parametrize("case", ["zeta_full", "mlp_full"])
def test_model_level_count_differs_from_the_pressure_levels(case, model_levels):
    """met_t::npl (model levels) is independent of met_t::np (pressure levels, mptrac.h:3856-3862): fewer and many
    more model levels than the 60 pressure levels of the grid, single steps and steps that share a launch."""
    ctl = dict(cases.CASES[case])
    names = cases.QUANTITIES_ML
    ctl.update(ctl_from_quantities(names))
    fields = cases.PRESSURE_LEVEL_FIELDS + ("pl", "ul", "vl", "wl", "zetal", "zeta_dotl")
    m0 = synthetic_met("C1", 0.0, 1.0, fields=fields, model_levels=model_levels)
    m1 = synthetic_met("C1", 3600.0, 1.25, fields=fields, model_levels=model_levels)
    assert m0.npl == model_levels != m0.np
    atm = synthetic_particles(6000, seed=5, quantities=names)
    for name_q in ("zeta", "eta"):
        atm["q"][list(names).index(name_q)] = 320.0 + 1680.0 * ((atm["lat"] + 85.0) / 170.0)
    clim = cases.load_clim_tropo()
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    ts = cases.step_times(o.ctl)
    for t in ts[:10]:
        o.run_timestep(t)
    for t in ts[:3]:
        s.run_timestep(t)
    s.run_timesteps(ts[3], 7)
    _compare(o, s)
    s.close()


# ---------------------------------------------------------------------------
# the BASELINE configurations on their own grids / default mixing grid / ensembles / long run
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("advect", [4, 2], ids=["rk4", "midpoint"])
def test_c3_grid_stochastic_parity(advect):
    """BASELINE configs[2] on ITS OWN grid (0.5 deg, 721 x 361 x 137 -- the grid bench.py runs): 2 x 10^5
    particles, RK4 (and the reference's default integrator, the midpoint scheme) + turbulent + mesoscale diffusion
    + convection + sedimentation, six steps against the multi-threaded oracle.  Exercises the 137-level pressure
    table and the 24-bit index arithmetic at the extents the headline number is quoted on."""
    n = 200000
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=n, grid="C3",
                                             fields=("u", "v", "w", "t", "ps", "pbl", "cape", "cin", "pel"),
                                             quantities=("m", "rp", "rhop"))
    ctl["advect"] = advect
    assert (m0.nx, m0.ny, m0.np) == (721, 361, 137)
    B.lib().orc_set_num_threads(B.usable_cores())
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    for t in cases.step_times(o.ctl)[:7]:
        o.run_timestep(t)
        s.run_timestep(t)
    _compare(o, s)
    s.close()


def _crowded(n, names, seed=5):
    """Particles inside a 20 deg x 10 deg x 5 km box: ~100 per cell of the default 1 deg x 1 deg x 1 km mixing grid."""
    atm = synthetic_particles(n, seed=seed, quantities=names, lon=(0.0, 20.0), lat=(0.0, 10.0), z=(5.0, 10.0))
    if "vmr" in names:
        atm["q"][list(names).index("vmr")] = 1e-9 * (1.0 + atm["lat"] / 10.0)
    return atm


def test_default_mixing_grid_360x180x90():
    """module_mixing on the reference's DEFAULT grid (MIXING_NX/NY/NZ 360 x 180 x 90 = 5.8 x 10^6 cells,
    mptrac.c:7631-7648) -- BASELINE configs[4] leaves it at that -- with module_sort, decay and deposition, ten steps."""
    from mptrac_amd.ctl import ctl_from_quantities
    names = cases.QUANTITIES + ("aoa",)
    ctl = {k: v for k, v in cases.CASES["full"].items() if k not in ("mixing_nx", "mixing_ny", "mixing_nz")}
    ctl.update(mixing_dt=180.0, sort_dt=540.0, **ctl_from_quantities(names))
    m0 = synthetic_met("C1", 0.0, 1.0, fields=cases.PRESSURE_LEVEL_FIELDS)
    m1 = synthetic_met("C1", 3600.0, 1.25, fields=cases.PRESSURE_LEVEL_FIELDS)
    atm = _crowded(100000, names)
    clim = cases.load_clim_tropo()
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    assert (s.ctl.mixing_nx, s.ctl.mixing_ny, s.ctl.mixing_nz) == (360, 180, 90)
    for t in cases.step_times(o.ctl)[:11]:
        o.run_timestep(t)
        s.run_timestep(t)
    im = names.index("m")
    assert np.abs(o.state()["q"][im] - atm["q"][im]).max() > 1e-4       # mixing moved the masses
    _compare(o, s)
    s.close()


@pytest.mark.parametrize("nens", [4, 1])
def test_ensemble_mixing(nens):
    """NENS > 0: every ensemble member mixes inside its own copy of the grid (index ens * ngrid + cell,
    mptrac.c:5291-5294, 5305-5316), member id from the quantity `ens`."""
    from mptrac_amd.ctl import ctl_from_quantities
    names = ("m", "vmr", "ens", "aoa")
    ctl = dict(cases.BASE, diffusion=1, turb_dz_trop=0.1, mixing_trop=1e-2, mixing_strat=1e-4, mixing_dt=180.0,
               mixing_nx=72, mixing_ny=36, mixing_nz=30, nens=nens, **ctl_from_quantities(names))
    m0 = synthetic_met("C1", 0.0, 1.0, fields=cases.PRESSURE_LEVEL_FIELDS)
    m1 = synthetic_met("C1", 3600.0, 1.25, fields=cases.PRESSURE_LEVEL_FIELDS)
    atm = _crowded(40000, names, seed=11)
    ie = names.index("ens")
    atm["q"][ie] = (np.arange(40000) * 7 % nens).astype(np.float64)
    atm["q"][names.index("m")] += atm["q"][ie]          # members differ: mixing across them would show
    clim = cases.load_clim_tropo()
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, m0, m1, atm)
    s.timesteps_init(0.0, 0.0)
    for t in cases.step_times(o.ctl)[:8]:
        o.run_timestep(t)
        s.run_timestep(t)
    r = o.state()
    assert np.array_equal(r["q"][ie], atm["q"][ie])
    assert np.abs(r["q"][0] - atm["q"][0]).max() > 1e-3
    _compare(o, s)
    s.close()


@pytest.mark.parametrize("sort_dt", [180.0, 360.0])
def test_sort_ahead_of_time_is_not_observable(sort_dt):
    """module_sort of the next time step starts on a second stream as soon as this step's particles have moved
    (option sort_ahead, default on) and is taken over if the next call comes with the expected time: the same
    bits as sorting when the call arrives -- with a sort every step and every other step, a download in between,
    a single-module call in between (drops the prepared sort) and a meteo hand-over."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=30000)
    ctl = dict(ctl, sort_dt=sort_dt, mixing_dt=180.0, t_stop=7200.0)
    m2 = synthetic_met("C1", 7200.0, 1.5, fields=cases.PRESSURE_LEVEL_FIELDS)
    runs = []
    for ahead in (1, 0):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("sort_ahead", ahead)
        s.timesteps_init(0.0, 0.0)
        seen = []
        for k, t in enumerate(cases.step_times(s.ctl)[:30]):
            if k == 21:
                s.swap_met(m2)
            s.run_timestep(t)
            if k == 5:
                seen.append(s.state())              # download between two steps
            if k == 9:
                s.module("position", t)             # touches the particles: a prepared sort must be dropped
            if k == 12:
                seen.append(s.sort())               # module_sort on its own
        seen.append(s.state())
        seen.append(s.get_cache()["dt"])
        runs.append(seen)
        s.close()
    for a, b in zip(*runs):
        if isinstance(a, dict):
            for key in ("time", "lon", "lat", "p", "q", "uvwp"):
                assert np.array_equal(a[key], b[key]), key
        elif isinstance(a, tuple):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        else:
            assert np.array_equal(a, b)


def test_keys_and_deposition_flags_from_the_step_kernel_are_not_observable():
    """BASELINE configs[4]'s schedule (module_sort and module_mixing in every step, decay, both deposition modules): the
    launch that moves the particles also writes the keys of the next module_sort, its module_timesteps, module_mixing's
    box and -- round 6 -- which particles the deposition launch behind module_mixing has to look at (EmitKeys, option
    emit_keys, default on).  With the option off a key kernel and the deposition launch derive all of that again from
    the stored state: the same bits, with particles released later (dt = 0 in the first steps), and the oracle's."""
    ctl, clim, m0, m1, atm = cases.make_case("full", n=30011)
    ctl = dict(ctl, sort_dt=180.0, mixing_dt=180.0)
    atm["time"][::7] = 540.0
    atm["p"][::5] = 1013.25 * np.exp(-np.linspace(0.01, 1.5, len(atm["p"][::5])) / 7.0)      # a fifth near the ground: both modules busy
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    times = cases.step_times(o.ctl)[:9]
    runs = []
    for emit in (1, 0):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("emit_keys", emit)
        s.timesteps_init(atm["time"].min(), atm["time"].max())
        for t in times:
            s.run_timestep(t)
        runs.append(s.state())
        if emit:
            for t in times:
                o.run_timestep(t)
            _compare(o, s)
        s.close()
    for k in ("time", "lon", "lat", "p", "q", "uvwp"):
        assert np.array_equal(runs[0][k], runs[1][k]), k
    wet, dry = (list(cases.QUANTITIES).index(k) for k in ("mloss_wet", "mloss_dry"))
    assert runs[0]["q"][wet].max() > 0 and runs[0]["q"][dry].max() > 0


def test_deposition_launch_with_packed_waves_equals_the_fused_tail():
    """The deposition modules behind module_mixing run in a kernel of their own that first packs the particles
    with anything to do into full waves (option compact_depo, default on); with the option off they run as the
    tail of the fused kernel -- the same bits, for the exponential-law and the Henry-law wet deposition, as single
    modules and inside time steps."""
    for case in ("full", "wet_henry"):
        ctl, clim, m0, m1, atm = cases.make_case(case, n=20011)
        ctl = dict(ctl, mixing_dt=180.0, mixing_trop=1e-3, mixing_strat=1e-6)
        runs = []
        for compact in (1, 0):
            s = hip.Simulation(ctl, clim, m0, m1, atm)
            s.set_option("compact_depo", compact)
            s.timesteps_init(0.0, 0.0)
            ts = cases.step_times(s.ctl)
            for t in ts[:6]:
                s.run_timestep(t)
            s.module("wet_depo", ts[5])
            s.module("dry_depo", ts[5])
            runs.append(s.state())
            s.close()
        for k in ("time", "lon", "lat", "p", "q", "uvwp"):
            assert np.array_equal(runs[0][k], runs[1][k]), (case, k)
        assert np.abs(runs[0]["q"] - atm["q"]).max() > 0


def test_long_run_400_steps_with_prefetched_handovers():
    """Drift check (was tools/gpu_soak.py): 2 x 10^4 particles, every module of the `full` case with
    module_meteo, module_sort every 10 and mixing every 5 steps, 400 time steps over 20 h, three meteo hand-overs
    through mphip_prefetch_met / mphip_commit_met, downloads on the way."""
    from mptrac_amd.ctl import ctl_from_quantities
    from mptrac_amd.synth import FIELDS_METEO_ONLY
    names = ("m", "vmr", "rp", "rhop", "loss_rate", "mloss_decay", "mloss_wet", "mloss_dry", "aoa", "t", "u", "ps", "theta")
    ctl = dict(cases.CASES["full"])
    ctl.update(ctl_from_quantities(names))
    ctl.update(t_stop=4 * 18000.0, dt_met=18000.0, met_dt_out=0.1, sort_dt=1800.0, mixing_dt=900.0)
    fields = tuple(cases.PRESSURE_LEVEL_FIELDS) + tuple(FIELDS_METEO_ONLY)
    mets = [synthetic_met("C1", 18000.0 * k, 1.0 + 0.1 * k, fields=fields) for k in range(6)]
    atm = synthetic_particles(20000, seed=7, quantities=names)
    clim = cases.load_clim_tropo()
    B.lib().orc_set_num_threads(B.usable_cores())
    o = B.Oracle(ctl, clim, mets[0], mets[1], atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, mets[0], mets[1], atm)
    s.timesteps_init(0.0, 0.0)
    s.prefetch_met(mets[2])
    imet = 0
    times = cases.step_times(o.ctl)
    assert len(times) == 401
    for k, t in enumerate(times):
        if t > mets[imet + 1].time:
            imet += 1
            o.swap_met(mets[imet + 1])
            s.commit_met()
            if imet + 2 < len(mets):
                s.prefetch_met(mets[imet + 2])
        o.run_timestep(t)
        s.run_timestep(t)
        if k % 97 == 0:
            _compare(o, s)
    assert imet == 3
    _compare(o, s)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case,over", [("conv_sedi", {}), ("conv_sedi", dict(advect=2)), ("advect", dict(advect=1)),
                                       ("turb", {}), ("full", {}), ("zeta_full", {}), ("mlp_full", dict(advect=2)),
                                       ("conv_sedi", dict(bound_lat0=-60.0, bound_lat1=60.0, bound_p0=1100.0, bound_p1=200.0,
                                                          bound_mass=2.0, bound_dps=300.0))],
                         ids=["conv_sedi", "conv_sedi-midpoint", "advect-euler", "turb", "full", "zeta_full", "mlp_full-midpoint",
                              "conv_sedi-bound"])
def test_big_grid_instantiations_equal_the_lean_ones(case, over):
    """Grids whose packed wind records exceed 4 GB take instantiations with 64-bit byte offsets (kBigGrid: the gated
    kernels, pressure and model levels).  Option big_grid forces them on a grid that fits 32 bits: only the addressing
    differs, so the results are the lean kernels' bit for bit -- single steps and steps that share a launch -- and the
    oracle's within the bar."""
    ctl, clim, m0, m1, atm = cases.make_case(case, n=5003)
    ctl = dict(ctl, **over)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    times = cases.step_times(o.ctl)
    runs = {}
    for name, big in (("lean", 0), ("big", 1)):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        s.set_option("big_grid", big)
        s.timesteps_init(0.0, 0.0)
        for t in times[:3]:
            s.run_timestep(t)
        s.run_timesteps(times[3], 7)
        runs[name] = s.state()
        runs[name]["ctr"] = s.get_cache()["rng_ctr"]
        s.close()
    for k in ("time", "lon", "lat", "p", "uvwp", "q"):
        assert np.array_equal(runs["big"][k], runs["lean"][k], equal_nan=True), k
    assert runs["big"]["ctr"] == runs["lean"]["ctr"]
    for t in times[:10]:
        o.run_timestep(t)
    r = o.state()
    assert np.array_equal(runs["big"]["time"], r["time"])
    for k in ("lon", "lat", "p"):
        assert cases.rel_err(runs["big"][k], r[k]) <= TOL, k


_BATCH_CASES = [(c, None) for c in ("advect", "turb", "diff", "conv_sedi", "full", "zeta_full", "mlp_full")] + [
    # the reference's default integrator (midpoint) and Euler: two-stage instantiations; trajectories with winds
    # from the model levels: the gated lean model-level instantiation
    ("advect", 2), ("conv_sedi", 2), ("turb", 2), ("diff", 1), ("zeta_full", 2), ("zeta_full", 1), ("mlp_full", 2),
    ("advect_zeta", None), ("advect_zeta", 2),
    # subsets without a kernel of their own take the gated instantiation, also several steps per launch: a gas tracer
    # (diffusion and convection, no sedimentation), turbulent diffusion + convection + sedimentation (own kernel for
    # single steps only), convection alone
    ("conv_sedi", "gas"), ("conv_sedi", "gas2"), ("conv_sedi", "turb_only"), ("conv_thresh", None), ("conv_thresh", 2),
    # module_meteo quantities: in every step (MET_DT_OUT below DT_MOD: only the last evaluation of a batch can be seen),
    # in every third step (a batch ends behind a step that schedules it when the next one does not), and evaluated
    # inside the step that schedules it (option lazy_meteo 0: a batch ends behind every such step)
    ("meteo", None), ("meteo", "every_third"), ("meteo", "eager_third"),
    # module_sort and module_mixing due in every fourth step only: the steps between share launches
    ("full", "sparse"),
    # convection due in every fourth step only (CONV_DT): the steps between share launches, without it
    ("conv_sedi", "conv_sparse"),
    # boundary conditions (module_bound_cond before and after the other modules of a step): the gated instantiation
    ("conv_sedi", "bound"), ("advect", "bound2"),
    # ... with winds from the model levels: the gated lean model-level instantiation
    ("zeta_full", "bound"), ("mlp_full", "bound2"),
    # the closure inside the boundary layer (TURB_PBL_SCHEME 1): gated instantiations of their own, in both integrators
    ("pbl", None), ("pbl_meso", None), ("pbl_meso", 2),
    # module_isosurf (isobaric, isopycnic, balloon; also on particles that are not released yet): the same instantiations
    ("isosurf_p", None), ("isosurf_rho", None), ("isosurf_rho", 2), ("isosurf_balloon", None),
    # ADVECT 0 (the C ABI accepts it): no module_advect, no instantiation for several steps -- single steps
    ("diff", 0)]
_BATCH_OVERRIDES = {"gas": dict(qnt_rp=-1, qnt_rhop=-1), "gas2": dict(qnt_rp=-1, qnt_rhop=-1, advect=2),
                    "turb_only": dict(turb_mesox=0.0, turb_mesoz=0.0), "every_third": dict(met_dt_out=540.0),
                    "eager_third": dict(met_dt_out=540.0), "sparse": dict(sort_dt=720.0, mixing_dt=720.0),
                    "conv_sparse": dict(conv_dt=720.0),
                    "bound": dict(bound_lat0=-60.0, bound_lat1=60.0, bound_p0=1100.0, bound_p1=200.0, bound_mass=2.0,
                                  bound_mass_trend=1e-6, bound_dps=300.0),
                    "bound2": dict(bound_lat0=-60.0, bound_lat1=60.0, bound_p0=1100.0, bound_p1=200.0, bound_mass=2.0,
                                   bound_pbl=1, advect=2)}


@pytest.mark.gpu
# variant: None, an integrator (ADVECT) or a key of _BATCH_OVERRIDES
@pytest.mark.parametrize("case,variant", _BATCH_CASES,
                         ids=[c if v is None else f"{c}-{v if isinstance(v, str) else 'advect%d' % v}" for c, v in _BATCH_CASES])
def test_run_timesteps_equals_the_step_by_step_loop(case, variant):
    """mphip_run_timesteps (the reference's time loop, trac.c:204-226, as one call): runs of steps with nothing
    scheduled between them share a kernel launch in which every particle takes its steps one after the other;
    same bits as one mphip_run_timestep per step -- state, uvwp and the counter of the random numbers --,
    whether the batches are long, short, cut by the internal re-sort, or (module sets with module_sort / mixing:
    "full") not possible at all; winds from the model levels (zeta / pressure advection) share launches too, and so
    does every integrator (ADVECT 4, 2, 1).  Where sharing is possible it must happen: seven quiet steps, one launch."""
    ctl, clim, m0, m1, atm = cases.make_case(case, n=5003)
    if isinstance(variant, str):
        ctl = dict(ctl, **_BATCH_OVERRIDES[variant])
    elif variant is not None:
        ctl = dict(ctl, advect=variant)
    o = B.Oracle(ctl, clim, m0, m1, atm)
    o.timesteps_init()
    cases.prepare(o)
    times = cases.step_times(o.ctl)
    runs, counts = {}, {}
    for name, multi, interval in (("loop", None, 4), ("batched", 64, 4), ("pairs", 2, 4), ("no_resort", 64, 0), ("off", 0, 4)):
        s = hip.Simulation(ctl, clim, m0, m1, atm)
        cases.prepare(s)
        s.set_option("locality_sort_interval", interval)
        if variant == "eager_third":
            s.set_option("lazy_meteo", 0)
        s.timesteps_init(0.0, 0.0)
        if multi is None:
            s.run_timestep(times[0])
            s.synchronize()
            s.profile_begin()
            for t in times[1:8]:
                s.run_timestep(t)
            counts["loop7"], _ = s.profile_end()      # step-kernel launches of the same seven steps, one call each
            for t in times[8:12]:
                s.run_timestep(t)
        else:
            s.set_option("multi_step", multi)
            s.run_timestep(times[0])
            s.synchronize()
            s.profile_begin()
            s.run_timesteps(times[1], 7)
            launches, _ = s.profile_end()
            counts[name] = launches
            if name == "no_resort" and variant in ("every_third", "eager_third", "conv_sparse"):
                assert 1 < launches < 7, (case, variant, launches)
            elif name == "no_resort" and variant == "sparse":
                assert launches < counts["loop7"], (case, variant, launches, counts)
            elif name == "no_resort" and variant == 0:
                assert launches == 7, (case, variant, launches)
            elif name == "no_resort" and case != "full":
                assert launches == 1, (case, variant, launches)
            if name == "off" or (case == "full" and variant != "sparse"):      # (module_mixing splits the launch of a step)
                assert launches >= 7 if case == "full" else launches == 7, (case, variant, name, launches)
            s.run_timesteps(times[8], 4)
        runs[name] = s.state()
        runs[name]["ctr"] = s.get_cache()["rng_ctr"]
        s.close()
    for name in ("batched", "pairs", "no_resort", "off"):
        for k in ("time", "lon", "lat", "p", "uvwp", "q"):
            assert np.array_equal(runs[name][k], runs["loop"][k], equal_nan=True), (name, k)
        assert runs[name]["ctr"] == runs["loop"]["ctr"], name
    for t in times[:12]:
        o.run_timestep(t)
    r = o.state()
    assert np.array_equal(runs["batched"]["time"], r["time"])
    for k in ("lon", "lat", "p"):
        assert cases.rel_err(runs["batched"][k], r[k]) <= TOL, k
