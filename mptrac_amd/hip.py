"""ctypes binding of the C ABI in include/mptrac_hip.h and a thin host-side
mirror of the reference's high-level interface (``mptrac_alloc`` ...
``mptrac_update_host``, src/mptrac.h:7246-7736) on top of it.

There is no CPU fallback: if the HIP library or a device is missing, creating
a ``Simulation`` raises.
"""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

from . import build as _build
from .ctl import TRACER_SERIES, ZONAL_MEANS, fill_ctl, make_ctl_struct
from .synth import FIELDS_2D, FIELDS_3D, FIELDS_ML

NQ_MAX = 16
MOD = {
    "timesteps": 1 << 0, "position": 1 << 1, "advect": 1 << 2, "diff_turb": 1 << 3, "diff_meso": 1 << 4,
    "convection": 1 << 5, "sedi": 1 << 6, "position2": 1 << 7, "loss_zero": 1 << 8, "decay": 1 << 9,
    "wet_depo": 1 << 10, "dry_depo": 1 << 11, "advect_init": 1 << 12, "diff_pbl": 1 << 13, "meteo": 1 << 14,
    "isosurf": 1 << 15, "sort": 1 << 16, "mixing": 1 << 17, "bound_cond": 1 << 18, "bound_cond2": 1 << 19,
    "isosurf_init": 1 << 20,
}

MphipCtl = make_ctl_struct("MphipCtl")
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


class MphipMet(C.Structure):
    _fields_ = [("time", C.c_double), ("coord_type", C.c_int), ("nx", C.c_int), ("ny", C.c_int),
                ("np", C.c_int), ("npl", C.c_int), ("lon", _dp), ("lat", _dp), ("p", _dp),
                ("sx", C.c_longlong), ("sy", C.c_longlong), ("sx2", C.c_longlong),
                ("sx_ml", C.c_longlong), ("sy_ml", C.c_longlong),
                ("f3", _fp * len(FIELDS_3D)), ("f2", _fp * len(FIELDS_2D))]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)

_lib = None


class MphipError(RuntimeError):
    pass


_LIVE = weakref.WeakSet()     # contexts not closed yet: closed at exit while the HIP runtime is still up


@atexit.register
def _close_live_contexts():
    for sim in list(_LIVE):
        try:
            sim.close()
        except Exception:
            pass


def lib_path():
    """The library load() takes: MPHIP_LIB, or with MPTRAC_AMD_EXACT=1 the reference-rounding build, else the default."""
    if os.environ.get("MPHIP_LIB"):
        return os.environ["MPHIP_LIB"]
    return _build.EXACT_LIB if _build.exact_requested() else _build.HIP_LIB


def load(build=True):
    """Load libmptrac_hip.so (building it first if sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build_hip() if build else lib_path()
    if not os.path.exists(path):
        raise MphipError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(path)
    L.mphip_sizeof_ctl.restype = C.c_size_t
    L.mphip_sizeof_met.restype = C.c_size_t
    L.mphip_version.restype = C.c_char_p
    L.mphip_last_error.restype = C.c_char_p
    L.mphip_last_error.argtypes = [C.c_void_p]
    L.mphip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.mphip_destroy.argtypes = [C.c_void_p]
    L.mphip_destroy.restype = None
    L.mphip_update_ctl.argtypes = [C.c_void_p, C.POINTER(MphipCtl)]
    L.mphip_update_clim.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int]
    L.mphip_update_clim_zm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
    L.mphip_update_clim_ts.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp]
    L.mphip_update_met.argtypes = [C.c_void_p, C.c_int, C.POINTER(MphipMet)]
    L.mphip_swap_met.argtypes = [C.c_void_p]
    L.mphip_prefetch_met.argtypes = [C.c_void_p, C.POINTER(MphipMet)]
    L.mphip_commit_met.argtypes = [C.c_void_p]
    L.mphip_prefetch_done.argtypes = [C.c_void_p]
    L.mphip_discard_prefetch.argtypes = [C.c_void_p]
    L.mphip_update_atm.argtypes = [C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_int,
                                   _dp, _dp, _dp, _dp, C.POINTER(_dp)]
    L.mphip_get_atm.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, C.POINTER(_dp)]
    L.mphip_update_quantity.argtypes = [C.c_void_p, C.c_int, _dp]
    L.mphip_update_cache.argtypes = [C.c_void_p, _fp, C.POINTER(C.c_uint64)]
    L.mphip_get_cache.argtypes = [C.c_void_p, _fp, _dp, C.POINTER(C.c_uint64)]
    L.mphip_update_iso.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int]
    L.mphip_get_iso.argtypes = [C.c_void_p, _dp]
    L.mphip_run_timestep.argtypes = [C.c_void_p, C.c_double]
    L.mphip_run_timesteps.argtypes = [C.c_void_p, C.c_double, C.c_int]
    L.mphip_module.argtypes = [C.c_void_p, C.c_uint, C.c_double]
    L.mphip_get_sort.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_int)]
    L.mphip_grid_sums.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_int), _dp, _dp]
    L.mphip_set_grid_kernel.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.mphip_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
    L.mphip_comm_unique_id.argtypes = [C.c_void_p]
    L.mphip_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.mphip_comm_destroy.argtypes = [C.c_void_p]
    L.mphip_comm_query.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mphip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.mphip_synchronize.argtypes = [C.c_void_p]
    L.mphip_profile_begin.argtypes = [C.c_void_p]
    L.mphip_profile_end.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), _dp]
    L.mphip_test_sincosf.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, _fp, _fp]
    L.mphip_test_rng.argtypes = [C.c_void_p, C.c_uint64, C.c_longlong, C.c_int, _dp]
    L.mphip_test_piece.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
    if hasattr(L, "mphip_test_libm"):      # (absent from libraries built before round 6: A/B runs through MPHIP_LIB)
        L.mphip_test_libm.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_longlong, _dp]
    if L.mphip_sizeof_ctl() != C.sizeof(MphipCtl):
        raise MphipError("mphip_ctl_t layout mismatch between header and Python mirror")
    if L.mphip_sizeof_met() != C.sizeof(MphipMet):
        raise MphipError("mphip_met_t layout mismatch between header and Python mirror")
    _lib = L
    return L


def _ptr(a, t):
    return a.ctypes.data_as(t)


def shard_range(n, rank, world):
    """Contiguous index range [lo, hi) of rank `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


class Simulation:
    """Host-side handle: one device context + the reference's call sequence.

    ``atm`` is a dict with float64 arrays time, p, lon, lat and q[nq][np]
    holding *all* particles; this process uploads its shard
    ``[lo, hi)`` (default: everything).
    """

    def __init__(self, ctl_kw, clim, met0, met1, atm, device=0, shard=None, rng_ctr=0, n_total=None):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.mphip_create(C.byref(self.h), device)
        if rc:
            raise MphipError(f"mphip_create failed with code {rc} (no usable HIP device?)")
        self.ctl = fill_ctl(MphipCtl(), **ctl_kw)
        self._cb = None
        _LIVE.add(self)
        self._mets = [None, None]
        self._next_met = None
        time, lat, tropo = clim[:3]
        tropo = np.ascontiguousarray(tropo, dtype=np.float64)
        self._chk(self.L.mphip_update_clim(self.h, len(time), len(lat),
                                           _ptr(np.ascontiguousarray(time, dtype=np.float64), _dp),
                                           _ptr(np.ascontiguousarray(lat, dtype=np.float64), _dp),
                                           _ptr(tropo, _dp), tropo.shape[1]))
        # zonal-mean climatologies: optional fourth element {name: (time, p, lat, vmr[ntime][np][nlat])}
        # ... and trace-gas time series {name: (time, vmr)}
        for name, tab in (clim[3] if len(clim) > 3 else {}).items():
            if name in TRACER_SERIES:
                self.update_clim_ts(name, *tab)
            else:
                self.update_clim_zm(name, *tab)
        self.update_ctl()
        self.set_met(0, met0)
        self.set_met(1, met1)
        # n_total given: `atm` holds only this process's range [lo, hi) of a
        # simulation with n_total particles; otherwise `atm` holds all of them
        self.atm_is_local = n_total is not None
        self.n_total = n_total if self.atm_is_local else len(atm["time"])
        self.lo, self.hi = shard if shard is not None else (0, self.n_total)
        self.n = self.hi - self.lo
        self.nq = self.ctl.nq
        self.update_atm(atm)
        ctr = C.c_uint64(rng_ctr)
        self._chk(self.L.mphip_update_cache(self.h, None, C.byref(ctr)))

    def update_clim_zm(self, name, time=(), p=(), lat=(), vmr=()):
        """A zonal-mean climatology of clim_t for module_meteo (hno3, oh, h2o2, ho2, o1d); no nodes: removed."""
        time, p, lat, vmr = (np.ascontiguousarray(a, dtype=np.float64) for a in (time, p, lat, vmr))
        if len(time):
            assert vmr.shape == (len(time), len(p), len(lat))
        self._chk(self.L.mphip_update_clim_zm(self.h, ZONAL_MEANS.index(name), len(time), len(p), len(lat),
                                              _ptr(time, _dp), _ptr(p, _dp), _ptr(lat, _dp), _ptr(vmr, _dp)))

    def update_clim_ts(self, name, time=(), vmr=()):
        """Surface time series of a trace gas for module_bound_cond (ccl4, ccl3f, ccl2f2, n2o, sf6); no nodes: removed."""
        time, vmr = (np.ascontiguousarray(a, dtype=np.float64) for a in (time, vmr))
        assert time.shape == vmr.shape
        self._chk(self.L.mphip_update_clim_ts(self.h, TRACER_SERIES.index(name), len(time), _ptr(time, _dp), _ptr(vmr, _dp)))

    # -- plumbing -----------------------------------------------------------
    def _chk(self, rc):
        if rc:
            raise MphipError(self.L.mphip_last_error(self.h).decode())

    def close(self):
        _LIVE.discard(self)
        if self.h:
            self.L.mphip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        # not at interpreter shutdown: the HIP runtime may already be gone then (the atexit hook below
        # has closed every live context before that)
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    # -- mptrac_update_device ------------------------------------------------
    def update_ctl(self):
        self._chk(self.L.mphip_update_ctl(self.h, C.byref(self.ctl)))

    def _met_struct(self, met):
        m = MphipMet()
        m.time, m.coord_type, m.nx, m.ny, m.np = met.time, met.coord_type, met.nx, met.ny, met.np
        m.lon, m.lat, m.p = _ptr(met.lon, _dp), _ptr(met.lat, _dp), _ptr(met.p, _dp)
        m.sx, m.sy, m.sx2 = met.ny * met.np, met.np, met.ny
        m.npl = met.npl if any(k in met.f3 for k in FIELDS_ML) else 0
        m.sx_ml, m.sy_ml = met.ny * met.npl, met.npl
        for i, k in enumerate(FIELDS_3D):
            m.f3[i] = _ptr(met.f3[k], _fp) if k in met.f3 else None
        for i, k in enumerate(FIELDS_2D):
            m.f2[i] = _ptr(met.f2[k], _fp) if k in met.f2 else None
        return m

    def set_met(self, slot, met):
        m = self._met_struct(met)
        self._chk(self.L.mphip_update_met(self.h, slot, C.byref(m)))
        self._mets[slot] = met

    def swap_met(self, new_met1):
        """mptrac_get_met when t passes met1: pointer swap, then read the next
        snapshot into met1 (src/mptrac.c:6486-6499)."""
        self._chk(self.L.mphip_swap_met(self.h))
        self._mets[0] = self._mets[1]
        self.set_met(1, new_met1)

    def prefetch_met(self, next_met):
        """Start the upload of the snapshot after met1 beside the time steps
        (copy stream); its arrays must stay untouched until commit_met()."""
        m = self._met_struct(next_met)
        self._chk(self.L.mphip_prefetch_met(self.h, C.byref(m)))
        self._next_met = next_met

    def commit_met(self):
        """met1 -> met0, prefetched snapshot -> met1 (no host wait)."""
        self._chk(self.L.mphip_commit_met(self.h))
        self._mets[0] = self._mets[1]
        self._mets[1] = self._next_met
        self._next_met = None

    def discard_prefetch(self):
        self._chk(self.L.mphip_discard_prefetch(self.h))
        self._next_met = None

    def prefetch_done(self):
        return bool(self.L.mphip_prefetch_done(self.h))

    def update_atm(self, atm):
        sl = slice(0, self.n) if self.atm_is_local else slice(self.lo, self.hi)
        arrs = [np.ascontiguousarray(atm[k][sl], dtype=np.float64) for k in ("time", "p", "lon", "lat")]
        assert all(len(a) == self.n for a in arrs)
        q = np.asarray(atm["q"], dtype=np.float64)
        q = q.reshape(-1, q.shape[-1]) if q.size else np.zeros((self.nq, 0))
        q = np.ascontiguousarray(q[:self.nq, sl])
        qp = (_dp * NQ_MAX)()
        for iq in range(self.nq):
            qp[iq] = _ptr(q[iq], _dp)
        self._chk(self.L.mphip_update_atm(self.h, self.n, self.lo, self.n_total, self.nq,
                                          *[_ptr(a, _dp) for a in arrs], qp))

    def update_quantity(self, iq, values):
        """One quantity array in the caller's order (mphip_update_quantity)."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.shape == (self.n,)
        self._chk(self.L.mphip_update_quantity(self.h, int(iq), _ptr(v, _dp)))

    def replace_particles(self, atm):
        """A new particle set (any count) in the same context: what a C caller does when it hands a different
        atm_t to mptrac_update_device."""
        self.atm_is_local = False
        self.n_total = len(atm["time"])
        self.lo, self.hi = 0, self.n_total
        self.n = self.n_total
        self.update_atm(atm)

    # -- mptrac_update_host ---------------------------------------------------
    def get_atm(self, out=None):
        """Particle arrays in the caller's order; `out` = a dict returned earlier, to download into the
        same host arrays again (what a C caller's persistent atm_t does)."""
        if out is None:
            out = {k: np.empty(self.n) for k in ("time", "p", "lon", "lat")}
            out["q"] = np.empty((self.nq, self.n))
        q = out["q"]
        qp = (_dp * NQ_MAX)()
        for iq in range(self.nq):
            qp[iq] = _ptr(q[iq], _dp)
        self._chk(self.L.mphip_get_atm(self.h, _ptr(out["time"], _dp), _ptr(out["p"], _dp),
                                       _ptr(out["lon"], _dp), _ptr(out["lat"], _dp), qp))
        out["q"] = q
        return out

    def get_cache(self):
        uvwp = np.empty((self.n, 3), dtype=np.float32)
        dt = np.empty(self.n)
        ctr = C.c_uint64()
        self._chk(self.L.mphip_get_cache(self.h, _ptr(uvwp, _fp), _ptr(dt, _dp), C.byref(ctr)))
        return {"uvwp": uvwp, "dt": dt, "rng_ctr": ctr.value}

    def state(self):
        s = self.get_atm()
        s["uvwp"] = self.get_cache()["uvwp"]
        return s

    def set_balloon(self, ts, ps):
        """The pressure time series module_isosurf_init reads for ISOSURF 4
        (src/mptrac.c:4925-4951)."""
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        ps = np.ascontiguousarray(ps, dtype=np.float64)
        self._chk(self.L.mphip_update_iso(self.h, None, _ptr(ts, _dp), _ptr(ps, _dp), len(ts)))

    def get_iso(self):
        iso = np.empty(self.n)
        self._chk(self.L.mphip_get_iso(self.h, _ptr(iso, _dp)))
        return iso

    # -- stepping -------------------------------------------------------------
    def timesteps_init(self, tmin, tmax):
        """module_timesteps_init (src/mptrac.c:6046-6073) -- host logic on the
        global minimum / maximum particle time."""
        c = self.ctl
        if c.direction == 1:
            c.t_start = tmin
            if c.t_stop > 1e99:
                c.t_stop = tmax
        else:
            c.t_start = tmax
            if c.t_stop > 1e99:
                c.t_stop = tmin
        if c.direction * (c.t_stop - c.t_start) <= 0:
            raise MphipError("Nothing to do! Check T_STOP and DIRECTION!")
        c.t_start = float((np.floor if c.direction == 1 else np.ceil)(c.t_start / c.dt_mod) * c.dt_mod)
        self.update_ctl()

    def run_timestep(self, t):
        self._chk(self.L.mphip_run_timestep(self.h, t))

    def module(self, name, t=0.0):
        self._chk(self.L.mphip_module(self.h, MOD[name], t))

    def sort(self):
        self._chk(self.L.mphip_module(self.h, MOD["sort"], 0.0))
        keys = np.empty(self.n)
        perm = np.empty(self.n, dtype=np.int32)
        self._chk(self.L.mphip_get_sort(self.h, _ptr(keys, _dp), _ptr(perm, C.POINTER(C.c_int))))
        return keys, perm

    def get_sort(self):
        """Sorted keys and permutation of the last module_sort (mphip_get_sort), without sorting again."""
        keys = np.empty(self.n)
        perm = np.empty(self.n, dtype=np.int32)
        self._chk(self.L.mphip_get_sort(self.h, _ptr(keys, _dp), _ptr(perm, C.POINTER(C.c_int))))
        return keys, perm

    def grid_sums(self, t, out=None):
        """Counts, sums of q and of q^2 per output cell (mphip_grid_sums).  `out` = (cnt, mean, sigma) of an
        earlier call: the arrays are filled again instead of allocated (a caller that writes one output after the
        other, like write_grid with its own buffers)."""
        ncell = self.ctl.grid_nx * self.ctl.grid_ny * self.ctl.grid_nz
        if out is not None:
            cnt, mean, sigma = out
            assert cnt.shape == (ncell,) and mean.shape == (self.nq, ncell) and sigma.shape == (self.nq, ncell)
        else:
            cnt = np.zeros(ncell, dtype=np.int32)
            mean = np.zeros((self.nq, ncell))
            sigma = np.zeros((self.nq, ncell))
        self._chk(self.L.mphip_grid_sums(self.h, t, _ptr(cnt, C.POINTER(C.c_int)), _ptr(mean, _dp),
                                         _ptr(sigma, _dp)))
        return cnt, mean, sigma

    def set_grid_kernel(self, kz=(), kw=()):
        """Vertical weighting function of the gridded output (GRID_KERNEL); no nodes: off."""
        kz = np.ascontiguousarray(kz, dtype=np.float64)
        kw = np.ascontiguousarray(kw, dtype=np.float64)
        assert kz.shape == kw.shape
        self._chk(self.L.mphip_set_grid_kernel(self.h, len(kz), _ptr(kz, _dp), _ptr(kw, _dp)))

    def synchronize(self):
        self._chk(self.L.mphip_synchronize(self.h))

    def run_timesteps(self, t_first, nsteps):
        """nsteps consecutive time steps starting at t_first (mphip_run_timesteps)."""
        self._chk(self.L.mphip_run_timesteps(self.h, float(t_first), int(nsteps)))

    def set_option(self, name, value):
        self._chk(self.L.mphip_set_option(self.h, name.encode(), float(value)))

    def set_allreduce(self, fn):
        """fn(device_pointer:int, count:int) -> None; sums `count` doubles at
        that device address over all ranks."""
        def _cb(ptr, count, _user):
            try:
                fn(ptr, count)
                return 0
            except Exception as e:   # surfaced as an error code to the C side
                print("allreduce hook failed:", e)
                return 1
        self._cb = ALLREDUCE_FN(_cb)
        self._chk(self.L.mphip_set_allreduce(self.h, self._cb, None))

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL identifier (created on rank 0, handed to the other ranks by the host)."""
        buf = C.create_string_buffer(128)
        if load().mphip_comm_unique_id(buf):
            raise MphipError("mphip_comm_unique_id failed (librccl not loadable?)")
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        """RCCL communicator of this context: the gridded reductions become all-reduces on its stream."""
        self._chk(self.L.mphip_comm_init(self.h, int(nranks), int(rank), C.c_char_p(bytes(unique_id))))

    def comm_destroy(self):
        self._chk(self.L.mphip_comm_destroy(self.h))

    def comm_query(self):
        """(ranks, rank) as the RCCL communicator reports them; (0, 0) without one."""
        n, r = C.c_int(), C.c_int()
        self._chk(self.L.mphip_comm_query(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def profile_begin(self):
        self._chk(self.L.mphip_profile_begin(self.h))

    def profile_end(self):
        n = C.c_longlong()
        ms = C.c_double()
        self._chk(self.L.mphip_profile_end(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # -- device self tests ------------------------------------------------------
    def test_sincosf(self, first_bits, count):
        c = np.empty(count, dtype=np.float32)
        s = np.empty(count, dtype=np.float32)
        self._chk(self.L.mphip_test_sincosf(self.h, first_bits, count, _ptr(c, _fp), _ptr(s, _fp)))
        return c, s

    def test_piece(self, piece, reps=1):
        """Profiling aid (tools/piece_cost.py): run one building block of the step kernel per particle."""
        chk = C.c_double(0.0)
        self._chk(self.L.mphip_test_piece(self.h, int(piece), int(reps), C.byref(chk)))
        return chk.value

    def test_rng(self, ctr, n, method):
        out = np.empty(n)
        self._chk(self.L.mphip_test_rng(self.h, ctr, n, method, _ptr(out, _dp)))
        return out

    def test_libm(self, fn, x, y=None, lds=False):
        """exp / log / pow / sqrt of the arrays as the kernels evaluate them (mphip_libm.h; tables from LDS if asked)."""
        op = {"exp": 0, "log": 1, "pow": 2, "sqrt": 3, "cos": 4, "sin": 5}[fn] + (16 if lds else 0)
        x = np.ascontiguousarray(x, dtype=np.float64)
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float64)
        out = np.empty_like(x)
        self._chk(self.L.mphip_test_libm(self.h, op, _ptr(x, _dp), _ptr(y, _dp) if y is not None else None, len(x), _ptr(out, _dp)))
        return out
