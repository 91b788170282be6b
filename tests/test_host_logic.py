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


@pytest.mark.parametrize("n", [3001, 3000])
def test_oracle_subsample_draws_the_full_runs_random_numbers(n):
    """Subsample mode of the oracle (orc_cache_t::ip_global): a few particles of a large run get the random
    numbers the full run binds to their slots, rs[3 * ip + k] (mptrac.c:4645-4647; flat Box-Muller pairs,
    mptrac.c:5820-5826 -- with an odd particle count the last pair reaches the extra uniform of mptrac.c:5797),
    and the counter advances as in the full run.  The picked particles of a full oracle run and the subsample
    run agree in every bit, in any order of the list."""
    ctl, clim, m0, m1, atm = cases.make_case("conv_sedi", n=n, grid="tiny")
    full = B.Oracle(ctl, clim, m0, m1, atm)
    pick = np.random.default_rng(7).choice(n, 257, replace=False)
    pick[:3] = (n - 1, 0, n - 2)
    sub_atm = {k: (v[pick].copy() if k != "q" else v[:, pick].copy()) for k, v in atm.items()}
    sub = B.Oracle(ctl, clim, m0, m1, sub_atm, ip_global=pick, np_global=n)
    full.timesteps_init()
    sub.timesteps_init()
    for t in cases.step_times(full.ctl)[:6]:
        full.run_timestep(t)
        sub.run_timestep(t)
    assert sub.cache.rng_ctr == full.cache.rng_ctr
    f, g = full.state(), sub.state()
    for k in ("time", "lon", "lat", "p", "uvwp"):
        assert np.array_equal(f[k][pick], g[k]), k
    assert np.array_equal(f["q"][:, pick], g["q"])
    assert np.abs(g["uvwp"]).max() > 0


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


# ---------------------------------------------------------------------------
# host layer: control files, classic netCDF, rendezvous (no GPU: the library is only loaded)
# ---------------------------------------------------------------------------

def _host_lib():
    import ctypes as C
    from mptrac_amd import build
    lib, _ = build.build_host()
    return C.CDLL(lib)


def test_job_from_environment_and_index_range_shards():
    """One process per GPU in the C driver: rank / world size / rendezvous address come from the launcher's
    environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT), and every rank keeps the index range
    [np rank / world, np (rank + 1) / world) of the particles it read -- the ranges of all ranks tile the set."""
    import ctypes as C
    import subprocess
    import sys
    from mptrac_amd import build
    lib, _ = build.build_host()
    NP, NQ = build.HOST_DIMS["NP"], build.HOST_DIMS["NQ"]
    code = r"""
import ctypes as C, sys, json
NP, NQ = %d, %d
class Job(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("local_rank", C.c_int), ("port", C.c_int), ("addr", C.c_char * 64)]
class Atm(C.Structure):
    _fields_ = [("np", C.c_int), ("time", C.c_double * NP), ("p", C.c_double * NP), ("lon", C.c_double * NP),
                ("lat", C.c_double * NP), ("q", (C.c_double * NP) * NQ)]
L = C.CDLL(%r)
job = Job()
L.mptrac_amd_job_from_env(C.byref(job))
atm = Atm()
atm.np = 1003
for i in range(1003):
    atm.time[i] = i; atm.lon[i] = 2 * i; atm.q[NQ - 1][i] = 3 * i
L.mptrac_amd_shard(C.byref(atm), C.byref(job))
print(json.dumps({"rank": job.rank, "world": job.world, "local": job.local_rank, "port": job.port,
                  "addr": job.addr.decode(), "np": atm.np, "first": atm.time[0], "last": atm.time[atm.np - 1],
                  "lon0": atm.lon[0], "q0": atm.q[NQ - 1][0]}))
""" % (NP, NQ, lib)
    import json
    import os
    seen = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="10.1.2.3",
                   MASTER_PORT="29000")
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert (d["rank"], d["world"], d["local"], d["addr"]) == (rank, 3, rank, "10.1.2.3") and d["port"] == 29001
        lo, hi = 1003 * rank // 3, 1003 * (rank + 1) // 3
        assert d["np"] == hi - lo and d["first"] == lo and d["last"] == hi - 1
        assert d["lon0"] == 2 * lo and d["q0"] == 3 * lo
        seen.append((lo, hi))
    assert seen[0][0] == 0 and seen[-1][1] == 1003 and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    # without a launcher: one rank
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert (d["rank"], d["world"], d["np"], d["addr"]) == (0, 1, 1003, "127.0.0.1")


def test_sharded_run_steps_through_the_times_of_the_whole_particle_file():
    """Ranks of one run must agree on t_start / t_stop (their all-reduces pair up step by step): with release
    times that grow along the file -- the usual emission file -- every rank's module_timesteps_init gives the
    range of the WHOLE file, not of its own index range; a rank whose range is empty (more ranks than particles)
    gets it too.  (tests/c/shard_times.c)"""
    from hostfiles import compile_c_test
    exe = compile_c_test("shard_times")
    from mptrac_amd import hip
    for n, world in ((10, 4), (2, 4), (7, 1), (1003, 8), (5, 8)):   # (8 ranks: one node of the baseline's configs[3] / [4])
        seen = []
        for rank in range(world):
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
            if world > 1:
                env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
            out = subprocess.run([exe, str(n), "DT_MOD", "180"], env=env, capture_output=True, text=True, timeout=120)
            assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
            seen.append((int(line[1]), float(line[2]), float(line[3])))
        assert sum(d[0] for d in seen) == n
        # ... the index ranges of the C driver are the ones of the Python harness (bench.py, tests)
        assert [d[0] for d in seen] == [hip.shard_range(n, r, world)[1] - hip.shard_range(n, r, world)[0] for r in range(world)]
        assert all(d[1:] == (900.0, 1000.0 + 900.0 * (n - 1)) for d in seen), seen
        if n < world:
            assert any(d[0] == 0 for d in seen)


def test_control_file_is_read_again_by_every_read_ctl(tmp_path):
    """The parsed control file is cached per path between look-ups, but mptrac_read_ctl always sees the file as
    it is now (the reference re-reads it for every key): an ensemble driver may rewrite the file in place --
    same name, same length, possibly the same time stamp.  (tests/c/reread_ctl.c)"""
    from hostfiles import compile_c_test
    exe = compile_c_test("reread_ctl")
    out = subprocess.run([exe, str(tmp_path / "trac.ctl")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "RESULT ok" in out.stdout, out.stdout[-2000:]


def test_control_file_lookup_rules(tmp_path):
    """scan_ctl: "NAME = VALUE" lines, first match wins, case-insensitive, NAME[i] / NAME[*], command-line pairs
    override the file, '-' means arguments only, defaults for missing keys."""
    import ctypes as C
    L = _host_lib()
    L.scan_ctl.restype = C.c_double
    L.scan_ctl.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_int, C.c_char_p, C.c_char_p]
    path = str(tmp_path / "trac.ctl")
    open(path, "w").write("# comment line\nDT_MOD = 240\ndt_mod = 999\nQNT_NAME[*] = any\nQNT_NAME[1] = rp\n"
                          "TWO_TOKENS 5\nT_STOP =   3600   trailing words\n")
    args = [b"trac", b"dirlist", b"trac.ctl", b"atm.tab", b"T_STOP", b"7200", b"ATM_BASENAME", b"atm"]
    argv = (C.c_char_p * len(args))(*args)

    def scan(name, idx=-1, default=b"", filename=path.encode()):
        buf = C.create_string_buffer(5000)
        v = L.scan_ctl(filename, len(args), argv, name, idx, default, buf)
        return v, buf.value.decode()
    assert scan(b"DT_MOD") == (240.0, "240")                     # first match in the file
    assert scan(b"T_STOP") == (7200.0, "7200")                   # the command line overrides the file
    assert scan(b"QNT_NAME", 1)[1] == "any"                      # NAME[*] comes first in the file
    assert scan(b"QNT_NAME", 0)[1] == "any"
    assert scan(b"TWO_TOKENS", default=b"7") == (7.0, "7")       # a setting needs name, separator and value
    assert scan(b"ADVECT", default=b"2") == (2.0, "2")
    assert scan(b"atm_basename")[1] == "atm"                     # names are case-insensitive
    assert scan(b"DT_MOD", default=b"180", filename=b"-") == (180.0, "180")   # '-': arguments only
    C.CDLL(None).fflush(None)    # (scan_ctl echoes the settings through C stdio: inside this test's capture, not at exit)


def test_species_presets_and_rejected_keys(tmp_path):
    """SPECIES sets the defaults of MOLMASS and the Henry constants (mptrac.c:7291-7383); a species that implies
    the OH chemistry, and switches this build does not implement, stop the run instead of being ignored."""
    import subprocess
    from mptrac_amd import build
    import hostfiles as hf
    _, trac = build.build_host()
    tmp = str(tmp_path)
    open(os.path.join(tmp, "dirlist"), "w").write(tmp + "\n")
    open(os.path.join(tmp, "atm.tab"), "w").write("0 10 0 0 1\n")

    def run(keys):
        hf.write_ctl(os.path.join(tmp, "trac.ctl"), dict({"NQ": 1, "QNT_NAME[0]": "m", "MET_TYPE": 1}, **keys))
        r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm.tab"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT)
        return r.returncode, r.stdout.decode()
    rc, out = run({"SPECIES": "SO2"})
    assert rc != 0 and "OH chemistry" in out and "MOLMASS = 64.066" in out
    rc, out = run({"SPECIES": "SO2", "OH_CHEM_REACTION": 0, "METBASE": os.path.join(tmp, "nothing")})
    assert "WET_DEPO_IC_H[0] = 0.013" in out and "WET_DEPO_IC_H[1] = 2900" in out and "WET_DEPO_BC_H[0] = 0.013" in out
    assert "OH chemistry" not in out                       # (stops later: no device / no meteo files here)
    rc, out = run({"SPECIES": "CO2", "MOLMASS": 44.5, "METBASE": os.path.join(tmp, "nothing")})
    assert "MOLMASS = 44.5" in out and "WET_DEPO_IC_H[1] = 2400" in out
    for key in ("RADIO_DEPO", "RADIO_DECAY", "KPP_CHEM", "H2O2_CHEM_REACTION"):
        rc, out = run({key: 1})
        assert rc != 0 and key in out and "not implemented" in out


def test_classic_netcdf_reader_against_scipy():
    """The host layer's own CDF-1 / CDF-2 reader (no netCDF library in the image) on the three meteo files of
    the reference's tests/coord_test: axes, dimensions, attributes and every value of t, u, v, w, sp."""
    import ctypes as C
    from scipy.io import netcdf_file
    L = _host_lib()
    L.ncc_open.restype = C.c_void_p
    L.ncc_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.ncc_close.argtypes = [C.c_void_p]
    L.ncc_find_var.argtypes = [C.c_void_p, C.c_char_p]
    L.ncc_var_ndims.argtypes = [C.c_void_p, C.c_int]
    L.ncc_var_dim.restype = C.c_longlong
    L.ncc_var_dim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char_p)]
    L.ncc_get_att.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_double)]
    L.ncc_read_double.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.POINTER(C.c_double)]
    L.ncc_read_float.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.POINTER(C.c_float)]
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_coord_test")
    for hour in range(3):
        path = os.path.join(here, "era5_utm32_2025_05_01_%02d.nc" % hour)
        err = C.create_string_buffer(256)
        nc = L.ncc_open(path.encode(), err, 256)
        assert nc, err.value
        f = netcdf_file(path, "r", mmap=False)
        for name in ("x", "y", "plev", "t", "u", "v", "w", "sp"):
            var, v = f.variables[name], L.ncc_find_var(nc, name.encode())
            assert v >= 0 and L.ncc_var_ndims(nc, v) == len(var.shape)
            for d, (dname, dlen) in enumerate(zip(var.dimensions, var.shape)):
                got = C.c_char_p()
                assert L.ncc_var_dim(nc, v, d, C.byref(got)) == dlen and got.value.decode() == dname
            ref = np.array(var[:]).ravel()
            if var.typecode() == "d":
                out = np.empty(ref.size)
                assert L.ncc_read_double(nc, v, 0, 0, ref.size, out.ctypes.data_as(C.POINTER(C.c_double)))
            else:
                out = np.empty(ref.size, dtype=np.float32)
                assert L.ncc_read_float(nc, v, 0, 0, ref.size, out.ctypes.data_as(C.POINTER(C.c_float)))
            assert np.array_equal(out, ref, equal_nan=True), name
            fill = getattr(var, "_FillValue", None)
            if fill is not None:
                got = C.c_double()
                assert L.ncc_get_att(nc, v, b"_FillValue", C.byref(got)) and np.float32(got.value) == np.float32(fill)
        assert L.ncc_find_var(nc, b"no_such_variable") < 0
        L.ncc_close(nc)
    bad = os.path.join(here, "atm_2025_05_01_00_00_00.tab")
    err = C.create_string_buffer(256)
    assert not L.ncc_open(bad.encode(), err, 256) and b"classic" in err.value


def test_rank_rendezvous_hands_the_identifier_to_every_rank():
    """mptrac_amd_bcast (host/rendezvous.c): rank 0's 128 bytes reach three other processes over TCP on
    127.0.0.1, whichever side is up first."""
    import socket
    import subprocess
    from mptrac_amd import build
    lib, _ = build.build_host()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = ("import ctypes as C, sys, time\n"
              "L = C.CDLL(%r)\nrank, world, port = int(sys.argv[1]), 4, %d\n"
              "time.sleep(0.4 * (rank == 0))\n"
              "buf = C.create_string_buffer(bytes((7 * i + 1) %% 256 for i in range(128)) if rank == 0 else bytes(128), 128)\n"
              "assert L.mptrac_amd_bcast(buf, 128, rank, world, b'127.0.0.1', port) == 1\n"
              "assert buf.raw == bytes((7 * i + 1) %% 256 for i in range(128)), rank\nprint('ok', rank)\n") % (lib, port)
    for addr in ("127.0.0.1", "localhost"):       # a launcher may export a host name
        w = worker.replace("b'127.0.0.1'", "b%r" % addr)
        procs = [subprocess.Popen([sys.executable, "-c", w, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                 for r in range(4)]
        for r, p in enumerate(procs):
            out = p.communicate(timeout=120)[0].decode()
            assert p.returncode == 0 and "ok %d" % r in out, out


def test_rank_rendezvous_gives_up_when_a_peer_never_comes():
    """Rank 0 waits for its peers for MPTRAC_RENDEZVOUS_TIMEOUT seconds, not for ever; so does a peer whose
    rank 0 never listens; a name that does not resolve fails at once."""
    import ctypes as C
    import socket
    import time
    L = _host_lib()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MPTRAC_RENDEZVOUS_TIMEOUT"] = "1"
    try:
        buf = C.create_string_buffer(128)
        for rank in (0, 1):
            t0 = time.time()
            assert L.mptrac_amd_bcast(buf, 128, rank, 2, b"127.0.0.1", port) == 0
            assert 0.9 < time.time() - t0 < 10.0
        t0 = time.time()
        assert L.mptrac_amd_bcast(buf, 128, 1, 2, b"no-such-host.invalid", port) == 0
        assert time.time() - t0 < 10.0
    finally:
        del os.environ["MPTRAC_RENDEZVOUS_TIMEOUT"]


def test_trac_command_line_conventions_of_the_reference_cli_test():
    """tests/cli_test/run.sh of the reference, for the driver: no arguments fail with the standard diagnostic;
    -h and --help succeed and print a usage section, with extra arguments too."""
    import subprocess
    from mptrac_amd import build
    _, trac = build.build_host()
    r = subprocess.run([trac], capture_output=True, text=True)
    assert r.returncode != 0 and "Missing or invalid command-line arguments." in r.stdout + r.stderr
    for flag in ("-h", "--help"):
        for extra in ([], ["extra-arg"]):
            r = subprocess.run([trac, flag] + extra, capture_output=True, text=True)
            assert r.returncode == 0 and "Usage:" in r.stdout


@pytest.mark.parametrize("style", ["old", "new"])
def test_hdf5_reader_on_chunked_compressed_files(tmp_path, style):
    """nc_hdf5.c on chunked, filtered data, which none of the reference's files has -- in the two on-disk styles
    netCDF-4 files come in: `old` = version-0 superblock, version-1 object headers, symbol-table group, version-1
    filter pipeline; `new` = version-2 superblock and object headers, link messages, version-2 dataspace / pipeline,
    version-3 attributes; both with version-1 chunk B-trees.  Shuffle + deflate, edge chunks that overhang the
    array, a chunk that was never written (fill value), packed shorts with scale / offset attributes, big-endian
    storage.  The files are made by tests/h5write.py from the format specification (no HDF5 library in the image)."""
    import subprocess
    import h5write
    from hostfiles import compile_c_test
    rng = np.random.default_rng(12)
    t3 = rng.normal(250.0, 20.0, (5, 13, 17)).astype("<f4")
    packed = rng.integers(-30000, 30000, (7, 10)).astype("<i2")
    big = rng.normal(0.0, 1.0, (4, 6)).astype(">f8")
    lev = np.array([1000.0, 850.0, 500.0, 250.0, 100.0])
    holes = np.arange(6 * 8, dtype="<f4").reshape(6, 8)
    w = h5write.Writer(style)
    w.dataset("t", t3, chunks=(2, 5, 8), shuffle=True, deflate=4)
    w.dataset("q", packed, chunks=(4, 4), deflate=1, attrs=(("scale_factor", np.float64(0.25)), ("add_offset", np.float64(-3.0))))
    w.dataset("b", big)
    w.dataset("lev", lev)
    w.dataset("holes", holes, chunks=(3, 4), shuffle=True, fill=np.float32(-7.5), skip_chunks=((1, 0),))
    path = str(tmp_path / "old_style.nc")
    w.close(path)
    exe = compile_c_test("nc_dump")
    res = subprocess.run([exe, path, "t", "q", "b", "lev", "holes"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "RESULT done" in res.stdout, res.stdout[-2000:]
    vals = {ln.split()[1]: np.array(ln.split()[2:], dtype=np.float64) for ln in res.stdout.splitlines() if ln.startswith("values ")}
    assert np.array_equal(vals["t"], t3.astype(np.float64).ravel())
    assert np.array_equal(vals["q"], packed.astype(np.float64).ravel())
    assert np.array_equal(vals["b"], big.astype(np.float64).ravel())
    assert np.array_equal(vals["lev"], lev)
    want = holes.astype(np.float64).copy()
    want[3:6, 0:4] = -7.5
    assert np.array_equal(vals["holes"], want.ravel())
    assert "att q scale_factor 0.25" in res.stdout and "att q add_offset -3" in res.stdout
    assert "var t 3" in res.stdout


def test_netcdf4_axis_names_come_from_the_dimension_lists(tmp_path):
    """The HDF5 reader names a variable's axes through its DIMENSION_LIST attribute (variable-length lists of
    object references in a global heap), as netCDF-4 does; naming by length is only the fall-back for files
    without the lists.  Built without the fall-back, the reader still names every axis of the reference's own
    netCDF-4 files (both superblock versions)."""
    import subprocess
    from mptrac_amd import build
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "nc_dump_lists")
    src = [os.path.join(here, "c", "nc_dump.c")] + [os.path.join(build.HOST_DIR, f) for f in ("nc_classic.c", "nc_hdf5.c")]
    subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-DNC_HDF5_NO_LENGTH_FALLBACK", "-I", build.HOST_DIR, "-o", exe, *src,
                           "-lz", "-lm"])
    want = {os.path.join(here, "golden", "ref_data", "cams_H2O2.nc"): "header H2O2 time=12 press=25 lat=241",
            os.path.join(here, "golden", "ref_dd_test", "init", "data.3.nc"): "header m time=1 NPARTS=16"}
    for path, line in want.items():
        out = subprocess.run([exe, path], capture_output=True, text=True, timeout=60).stdout
        assert line in out.splitlines() and "phony_dim" not in out, out


def test_atm2grid_command_line_conventions():
    """atm2grid follows the reference's tool conventions (tests/cli_test) like the driver."""
    import subprocess
    from mptrac_amd import build
    build.build_host()
    r = subprocess.run([build.ATM2GRID_BIN], capture_output=True, text=True)
    assert r.returncode != 0 and "Missing or invalid command-line arguments." in r.stdout + r.stderr
    for flag in ("-h", "--help"):
        r = subprocess.run([build.ATM2GRID_BIN, flag, "extra-arg"], capture_output=True, text=True)
        assert r.returncode == 0 and "Usage:" in r.stdout


def test_conversion_tools_command_line_conventions():
    """atm_conv and met_conv follow the reference's tool conventions (tests/cli_test)."""
    import subprocess
    from mptrac_amd import build
    build.build_host()
    for exe in (build.ATM_CONV_BIN, build.MET_CONV_BIN):
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode != 0 and "Missing or invalid command-line arguments." in r.stdout + r.stderr
        for flag in ("-h", "--help"):
            r = subprocess.run([exe, flag], capture_output=True, text=True)
            assert r.returncode == 0 and "Usage:" in r.stdout
