"""BASELINE configs at their full sizes on the GPU box, checked against the oracle on a subsample.

The stochastic modules bind the random numbers to the particle's slot of the whole run (rs[3 * ip + k],
src/mptrac.c:4645-4647, 5797-5826), so a few thousand particles of a 10^7 or 10^8 run can only be checked by an
oracle that draws the numbers of those slots: orc_cache_t::ip_global (oracle/mptrac_oracle.h; its agreement with
a full oracle run is a CPU test, tests/test_host_logic.py::test_oracle_subsample_draws_the_full_runs_random_numbers).

  * configs[2] (C3): 10^7 particles, 721 x 361 x 137, RK4 + turbulent + mesoscale diffusion + convection +
    sedimentation, 20 steps through mphip_run_timesteps -- the very call bench.py times; the same with winds from the
    model levels (bench workload C3z) and with the boundary-layer closure (C3p);
  * configs[3] (C4): 10^8 particles as eight index-range shards of 1.25 x 10^7 (eight contexts on the one GPU of
    the box, one host thread each, their gridded output summed through the all-reduce hook) against ONE context
    holding all 10^8, and against the oracle on a subsample;
  * configs[4] (C5), the part one GPU runs: 10^7 particles with module_sort, inter-parcel mixing on the default boxes,
    decay and deposition in every step -- against the oracle running ALL particles (no subsample can follow the sort and
    the mixing);
  * mixing in every step without module_sort (bench workload C3x), all particles in the oracle;
  * module_mixing's exchange between ranks restricted to the occupied levels (three contexts through the hook).
(Every visible GPU with the library's own RCCL communicator: tests/test_zz_every_visible_gpu.py.)
"""
import ctypes
import os
import sys
import threading

import numpy as np
import pytest

import cases
from mptrac_amd import hip
from oracle import binding as B

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from mptrac_amd import build as _build
# BASELINE north_star: positions within 1e-10 relative of the CPU reference -- and with MPTRAC_AMD_EXACT=1, the
# reference-rounding build, no difference at all (rel_err <= 0: the oracle's bits at the full sizes, too)
TOL = 0.0 if _build.exact_requested() else 1e-10


def _c3_inputs(n, first=0, n_steps=20, workload="C3"):
    import bench
    ctl, clim, met0, met1, atm, _, _ = bench.build_inputs(workload, 0, 1, n_steps + 1, particles=n)
    if first:
        from mptrac_amd.synth import synthetic_particles
        atm = synthetic_particles(n, seed=12345, quantities=("m", "rp", "rhop"), first=first)
    return ctl, clim, met0, met1, atm


def _oracle_on_subsample(ctl, clim, met0, met1, atm_sub, pick, n_total, n_steps):
    o = B.Oracle(ctl, clim, met0, met1, atm_sub, ip_global=pick, np_global=n_total)
    o.timesteps_init()
    for k in range(n_steps + 1):
        o.run_timestep(k * o.ctl.dt_mod)
    return o


def _take(atm, idx):
    return {k: (v[idx].copy() if k != "q" else v[:, idx].copy()) for k, v in atm.items()}


@pytest.mark.parametrize("n_steps", [20, 65])
def test_c3_at_1e7_on_its_own_grid_against_the_oracle_subsample(n_steps):
    """BASELINE configs[2] as bench.py runs it: 10^7 particles, 20 steps in one mphip_run_timesteps call (the
    driver's arguments) -- and 65, bench.py's default 60 and a few more: the call then contains the re-sort of the
    internal locality order (every 60 steps), which splits the launch and moves every particle to another slot."""
    n = 10 ** 7
    ctl, clim, met0, met1, atm = _c3_inputs(n, n_steps=n_steps)
    s = hip.Simulation(ctl, clim, met0, met1, atm)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    s.run_timestep(0.0)
    s.run_timesteps(dt, n_steps)
    g = s.state()
    cnt, mean, _ = s.grid_sums(n_steps * dt)
    ctr = s.get_cache()["rng_ctr"]
    s.close()

    pick = np.random.default_rng(20250930).choice(n, 5000, replace=False)
    pick[:2] = (0, n - 1)
    o = _oracle_on_subsample(ctl, clim, met0, met1, _take(atm, pick), pick, n, n_steps)
    assert o.cache.rng_ctr == ctr                                         # the counter of the full run
    for k, ref in (("lon", o.lon), ("lat", o.lat), ("p", o.p)):
        err = cases.rel_err(g[k][pick], ref)
        assert err <= TOL, (k, err)
    assert np.array_equal(g["time"][pick], o.time)
    assert np.array_equal(g["uvwp"][pick], o.uvwp)                         # float statistics: identical bits
    assert np.array_equal(g["q"][:, pick], o.q)                            # m, rp, rhop untouched by these modules
    # every particle took every step, stayed on the globe and is counted once
    assert np.all(g["time"] == n_steps * dt)
    for k in ("lon", "lat", "p"):
        assert np.all(np.isfinite(g[k])), k
    assert g["lon"].min() >= -180.0 and g["lon"].max() < 180.0 and np.abs(g["lat"]).max() <= 90.0 and g["p"].min() > 0.0
    assert np.abs(g["lon"] - atm["lon"]).max() > 1.0                       # (they did move)
    inside = _inside_output_grid(o.ctl, g)
    assert int(cnt.sum()) == int(np.count_nonzero(inside))
    assert abs(mean[0].sum() - g["q"][0][inside].sum()) <= 1e-9 * n


@pytest.mark.parametrize("workload", ["C3z", "C3p", "C3d", "C3m", "C2"])
def test_c3_variants_at_1e7_against_the_oracle_subsample(workload):
    """The two other step kernels bench.py times on the 721 x 361 x 137 grid at their own size: winds from the model
    levels (ADVECT_VERT_COORD 1, intpol_met_4d_zeta: mptrac.c:2808-2981, 3681-3757 -- workload C3z) and the
    boundary-layer closure (TURB_PBL_SCHEME 1, module_diff_pbl: mptrac.c:4343-4584 -- workload C3p), the reference's
    default integrator (ADVECT 2, C3d) and module_meteo in every step (C3m); 10^7 particles, 20 steps in one
    mphip_run_timesteps call, 5 000 of them against the oracle.  C2: BASELINE configs[1] as it stands -- 10^6 particles,
    advection + turbulent diffusion only (the lean "advect + turb" instantiation), 361 x 181 x 137."""
    import bench
    n, n_steps = bench.WORKLOADS[workload][1], 20
    ctl, clim, met0, met1, atm = _c3_inputs(n, workload=workload)
    s = hip.Simulation(ctl, clim, met0, met1, atm)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    s.run_timestep(0.0)
    s.synchronize()
    s.profile_begin()
    s.run_timesteps(dt, n_steps)
    launches, _ = s.profile_end()
    g = s.state()
    ctr = s.get_cache()["rng_ctr"]
    s.close()
    assert launches == 1 or workload == "C3m"                            # (the lean instantiations: the steps share a launch)

    pick = np.random.default_rng(20250930).choice(n, 5000, replace=False)
    pick[:2] = (0, n - 1)
    o = _oracle_on_subsample(ctl, clim, met0, met1, _take(atm, pick), pick, n, n_steps)
    assert o.cache.rng_ctr == ctr
    for k, ref in (("lon", o.lon), ("lat", o.lat), ("p", o.p)):
        err = cases.rel_err(g[k][pick], ref)
        assert err <= TOL, (workload, k, err)
    assert np.array_equal(g["time"][pick], o.time)
    assert np.array_equal(g["uvwp"][pick], o.uvwp)
    err, row = cases.q_rows_err(o.ctl, g["q"][:, pick], o.q)               # (C3z: the zeta coordinate is advected)
    assert err <= TOL, (workload, row, err)
    assert np.all(g["time"] == n_steps * dt)
    for k in ("lon", "lat", "p"):
        assert np.all(np.isfinite(g[k])), k
    assert g["lon"].min() >= -180.0 and g["lon"].max() < 180.0 and np.abs(g["lat"]).max() <= 90.0 and g["p"].min() > 0.0
    moved = np.abs(g["p"][pick] - atm["p"][pick]) > 1e-3
    assert np.count_nonzero(moved) > 4000


def test_c3_at_1e7_across_two_meteo_hand_overs():
    """configs[2] over three meteo intervals of one hour: 10^7 particles, 45 steps, the next snapshot uploaded beside the
    steps and handed over between two calls (mphip_prefetch_met / mphip_commit_met: get_met's swap of the two time
    levels, mptrac.c:6403-6491), the steps of an interval in one mphip_run_timesteps call; 5 000 particles against
    the oracle that swaps its snapshots the same way."""
    from mptrac_amd.synth import synthetic_met
    import bench
    n = 10 ** 7
    ctl, clim, _, _, atm, _, _ = bench.build_inputs("C3", 0, 1, 1, particles=n)
    fields = bench.WORKLOADS["C3"][4]
    ctl.update(dt_met=3600.0, t_stop=3 * 3600.0)
    mets = [synthetic_met("C3", 3600.0 * k, 1.0 + 0.1 * k, fields=fields) for k in range(4)]
    pick = np.random.default_rng(20250930).choice(n, 5000, replace=False)
    o = B.Oracle(ctl, clim, mets[0], mets[1], _take(atm, pick), ip_global=pick, np_global=n)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, mets[0], mets[1], atm)
    s.timesteps_init(0.0, 0.0)
    s.prefetch_met(mets[2])
    dt = s.ctl.dt_mod
    times = [k * dt for k in range(46)]
    imet, pending = 0, []
    for t in times:
        if t > mets[imet + 1].time:
            s.run_timesteps(pending[0], len(pending))
            pending = []
            imet += 1
            o.swap_met(mets[imet + 1])
            s.commit_met()
            if imet + 2 < len(mets):
                s.prefetch_met(mets[imet + 2])
        o.run_timestep(t)
        pending.append(t)
    s.run_timesteps(pending[0], len(pending))
    g = s.state()
    ctr = s.get_cache()["rng_ctr"]
    s.close()
    assert imet == 2 and o.cache.rng_ctr == ctr
    for k, ref in (("lon", o.lon), ("lat", o.lat), ("p", o.p)):
        err = cases.rel_err(g[k][pick], ref)
        assert err <= TOL, (k, err)
    assert np.array_equal(g["time"][pick], o.time) and np.array_equal(g["uvwp"][pick], o.uvwp)
    assert np.all(g["time"] == times[-1])


def _inside_output_grid(ctl, g):
    """The particles write_grid bins (mptrac.c:13847-13860) on the default output grid."""
    z = 7.0 * np.log(1013.25 / g["p"])
    return ((g["lon"] >= ctl.grid_lon0) & (g["lon"] < ctl.grid_lon1) & (g["lat"] >= ctl.grid_lat0)
            & (g["lat"] < ctl.grid_lat1) & (z >= ctl.grid_z0) & (z < ctl.grid_z1))


def test_c5_at_1e7_with_the_default_mixing_grid_against_the_full_oracle():
    """BASELINE configs[4], the part one GPU runs, at its own size: 10^7 particles on 721 x 361 x 137 with module_sort and
    inter-parcel mixing (default 360 x 180 x 90 boxes) in every step, decay, wet and dry deposition.  module_sort rebinds
    the random-number slots and module_mixing couples the particles, so no subsample can follow: the oracle runs ALL
    particles for three calls of the time step (two of them move the particles; the second module_sort is the order
    repair at full size).  Positions and quantities <= 1e-10, the last sort's keys and permutation identical."""
    import bench
    n = 10 ** 7
    ctl, clim, met0, met1, atm, _, _ = bench.build_inputs("C5", 0, 1, 4, particles=n)
    B.lib().orc_set_num_threads(B.usable_cores())
    o = B.Oracle(ctl, clim, met0, met1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, met0, met1, atm)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    for k in range(3):
        s.run_timestep(k * dt)
    g = s.state()
    keys, perm = s.get_sort()
    ctr = s.get_cache()["rng_ctr"]
    s.close()
    for k in range(2):
        o.run_timestep(k * dt)
    before = o.state()
    ko, po = o.sort()                       # module_sort of the third call: keys and permutation (ties by index)
    assert np.array_equal(perm, po) and np.array_equal(keys, ko[po])
    # (o.sort() has re-ordered the oracle's particles; put them back and let the third call run as a whole)
    for name in ("time", "lon", "lat", "p"):
        getattr(o, name)[:] = before[name]
    o.q[:] = before["q"]
    o.run_timestep(2 * dt)
    r = o.state()
    assert o.cache.rng_ctr == ctr
    assert np.array_equal(g["time"], r["time"])
    for k in ("lon", "lat", "p"):
        err = cases.rel_err(g[k], r[k])
        assert err <= TOL, (k, err)
    err, row = cases.q_rows_err(o.ctl, g["q"], r["q"])
    assert err <= TOL, (row, err)
    assert np.abs(r["q"][0] - atm["q"][0][po]).max() > 1e-6          # (mixing, decay and deposition did something)


def test_mixing_without_module_sort_at_1e7_against_the_full_oracle():
    """Inter-parcel mixing in every step WITHOUT module_sort (bench workload C3x): the particles stay in the caller's
    order as far as the caller can see, the library stores them in its internal locality order, and the ordered cell
    sums of module_mixing (mptrac.c:5169-5347) have to follow the caller's index through that order -- the other
    branch of the ordered sums than configs[4] takes.  10^7 particles, the default boxes, all particles in the oracle;
    three calls one at a time, then five steps in one mphip_run_timesteps call."""
    import bench
    n = 10 ** 7
    ctl, clim, met0, met1, atm, _, _ = bench.build_inputs("C3x", 0, 1, 9, particles=n)
    B.lib().orc_set_num_threads(B.usable_cores())
    o = B.Oracle(ctl, clim, met0, met1, atm)
    o.timesteps_init()
    s = hip.Simulation(ctl, clim, met0, met1, atm)
    s.timesteps_init(0.0, 0.0)
    dt = s.ctl.dt_mod
    for k in range(3):
        s.run_timestep(k * dt)
    s.run_timesteps(3 * dt, 5)
    g = s.state()
    ctr = s.get_cache()["rng_ctr"]
    s.close()
    for k in range(8):
        o.run_timestep(k * dt)
    r = o.state()
    assert o.cache.rng_ctr == ctr
    assert np.array_equal(g["time"], r["time"]) and np.array_equal(g["uvwp"], r["uvwp"])
    for k in ("lon", "lat", "p"):
        err = cases.rel_err(g[k], r[k])
        assert err <= TOL, (k, err)
    err, row = cases.q_rows_err(o.ctl, g["q"], r["q"])
    assert err <= TOL, (row, err)
    assert np.abs(r["q"][0] - atm["q"][0]).max() > 1e-6               # (mixing did something)


class _ThreadAllreduce:
    """All-reduce over N contexts of one process, one host thread each: the partial sums meet on the host and
    are added in rank order (what a collective over N processes does, through the hook of mphip_set_allreduce)."""

    def __init__(self, nranks):
        self.n = nranks
        self.barrier = threading.Barrier(nranks)
        self.parts = [None] * nranks
        self.calls = [0] * nranks
        self.rt = ctypes.CDLL("libamdhip64.so")
        self.rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def hook(self, rank):
        def fn(ptr, count):
            host = np.empty(count)
            assert self.rt.hipMemcpy(host.ctypes.data, ctypes.c_void_p(int(ptr)), count * 8, 2) == 0
            self.parts[rank] = host
            self.barrier.wait()
            total = self.parts[0].copy()
            for r in range(1, self.n):
                total += self.parts[r]
            self.barrier.wait()
            assert self.rt.hipMemcpy(ctypes.c_void_p(int(ptr)), total.ctypes.data, count * 8, 1) == 0
            self.calls[rank] += 1
        return fn


def test_c4_1e8_particles_as_eight_shards_equal_one_context_and_the_oracle_subsample():
    """BASELINE configs[3] on one GPU: 10^8 particles sharded by index range over eight contexts (replicated
    meteo grids, no exchange in the step, gridded output summed over the shards), against one context with all
    10^8 particles.  Positions: identical bits (random numbers follow the global index).  Gridded output: counts
    exact, sums to 1e-13 (eight partial sums instead of one serial sum)."""
    world, n_shard, n_steps = 8, 12_500_000, 20
    n = world * n_shard
    ctl, clim, met0, met1, atm = _c3_inputs(n)
    one = hip.Simulation(ctl, clim, met0, met1, atm)
    one.timesteps_init(0.0, 0.0)
    dt = one.ctl.dt_mod
    one.run_timestep(0.0)
    one.run_timesteps(dt, n_steps)
    ref = one.state()
    ref_cnt, ref_mean, ref_sig = one.grid_sums(n_steps * dt)
    ctr = one.get_cache()["rng_ctr"]
    one.close()
    assert np.all(ref["time"] == n_steps * dt) and np.all(np.isfinite(ref["p"]))

    pick = np.sort(np.random.default_rng(4).choice(n, 4000, replace=False))
    pick[0], pick[-1] = 0, n - 1
    o = _oracle_on_subsample(ctl, clim, met0, met1, _take(atm, pick), pick, n, n_steps)
    assert o.cache.rng_ctr == ctr
    for k, r in (("lon", o.lon), ("lat", o.lat), ("p", o.p)):
        assert cases.rel_err(ref[k][pick], r) <= TOL, k
    assert np.array_equal(ref["uvwp"][pick], o.uvwp)

    ar = _ThreadAllreduce(world)
    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            lo, hi = hip.shard_range(n, rank, world)
            s = hip.Simulation(ctl, clim, met0, met1, atm, shard=(lo, hi))
            s.set_allreduce(ar.hook(rank))
            s.timesteps_init(0.0, 0.0)
            s.run_timestep(0.0)
            s.run_timesteps(dt, n_steps)
            g = s.state()
            out = s.grid_sums(n_steps * dt)
            results[rank] = (lo, hi, g, out)
            s.close()
        except BaseException as exc:      # noqa: BLE001  (a dead rank must not leave the others at the barrier)
            errors.append((rank, repr(exc)))
            ar.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert ar.calls == [1] * world          # one exchange: counts and sums travel in one buffer of doubles
    for rank in range(world):
        lo, hi, g, (cnt, mean, sig) = results[rank]
        for k in ("time", "lon", "lat", "p"):
            assert np.array_equal(g[k], ref[k][lo:hi]), (rank, k)
        assert np.array_equal(g["uvwp"], ref["uvwp"][lo:hi]), rank
        assert np.array_equal(g["q"], ref["q"][:, lo:hi]), rank
        assert np.array_equal(cnt, ref_cnt), rank                      # every rank holds the reduced output
        assert int(cnt.sum()) == int(ref_cnt.sum()) > 0.99 * n
        assert cases.rel_err_strict(mean, ref_mean, floor=1e-30) <= 1e-13
        assert cases.rel_err_strict(sig, ref_sig, floor=1e-30) <= 1e-13


def test_mixing_exchange_of_the_occupied_levels_equals_the_whole_grid():
    """N > 1 ranks exchange the cell sums of module_mixing only for the band of levels that holds particles on any
    rank (exchange_occupied_levels: a small all-reduce of the per-level occupancy, then pack / all-reduce / unpack of the
    band).  Three index-range shards on the one GPU (three contexts, one host thread each, the all-reduce hook) on the
    reference's default 360 x 180 x 90 mixing grid: identical bits with the exchange of the whole grid, the hook sees a
    third of the bytes, and the quantities agree with ONE context holding all particles to 1e-13 (three partial sums
    instead of one serial sum)."""
    world, n = 3, 60003
    ctl, clim, m0, m1, atm = cases.make_case("full", n=n)
    ctl.update(sort_dt=-999.0, mixing_dt=180.0, mixing_nx=360, mixing_ny=180, mixing_nz=90)
    times = None
    one = hip.Simulation(ctl, clim, m0, m1, atm)
    one.timesteps_init(0.0, 0.0)
    times = cases.step_times(one.ctl)[:6]
    for t in times:
        one.run_timestep(t)
    ref = one.state()
    one.close()
    results = {}
    for levels in (1, 0):     # (option mix_exchange_levels; 0 is the default)
        ar = _ThreadAllreduce(world)
        counts_seen = [[] for _ in range(world)]
        out, errors = [None] * world, []

        def rank_main(rank, levels=levels, ar=ar, counts_seen=counts_seen, out=out, errors=errors):
            try:
                lo, hi = hip.shard_range(n, rank, world)
                s = hip.Simulation(ctl, clim, m0, m1, atm, shard=(lo, hi))
                s.set_option("mix_exchange_levels", levels)
                inner = ar.hook(rank)

                def hook(ptr, count):
                    counts_seen[rank].append(count)
                    inner(ptr, count)
                s.set_allreduce(hook)
                s.timesteps_init(0.0, 0.0)
                for t in times:
                    s.run_timestep(t)
                out[rank] = (lo, hi, s.state())
                s.close()
            except BaseException as exc:      # noqa: BLE001
                errors.append((rank, repr(exc)))
                ar.barrier.abort()
        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        results[levels] = (out, counts_seen)
    full = 360 * 180 * 90
    whole, band = results[0][1][0], results[1][1][0]
    assert whole and all(c % full == 0 for c in whole)                 # the whole grid: sums of the mixed quantities, then counts
    assert 90 in band and all(c == 90 or c % (360 * 180) == 0 for c in band)   # the occupancy, then whole levels of every column
    assert sum(band) < 0.5 * sum(whole)
    for rank in range(world):
        lo, hi, g1 = results[1][0][rank]
        _, _, g0 = results[0][0][rank]
        for k in ("time", "lon", "lat", "p", "q", "uvwp"):
            assert np.array_equal(g1[k], g0[k]), (rank, k)
        for k in ("time", "lon", "lat", "p", "uvwp"):
            assert np.array_equal(g1[k], ref[k][lo:hi]), (rank, k)
        err = float(np.max(np.abs(g1["q"] - ref["q"][:, lo:hi]) / np.maximum(np.abs(ref["q"][:, lo:hi]).max(axis=1, keepdims=True), 1e-300)))
        assert err <= 1e-13, (rank, err)
    assert np.abs(ref["q"][0] - atm["q"][0]).max() > 1e-6          # (mixing and the loss modules did something)
