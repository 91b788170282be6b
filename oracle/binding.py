"""ctypes binding of the CPU oracle (oracle/mptrac_oracle.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (mptrac_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from mptrac_amd.ctl import TRACER_SERIES, ZONAL_MEANS, make_ctl_struct, fill_ctl
from mptrac_amd.synth import FIELDS_2D, FIELDS_3D

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "build", "libmptrac_oracle.so")
NQ_MAX = 16

OrcCtl = make_ctl_struct("OrcCtl")
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


class OrcMet(C.Structure):
    _fields_ = [("time", C.c_double), ("coord_type", C.c_int), ("nx", C.c_int), ("ny", C.c_int),
                ("np", C.c_int), ("npl", C.c_int), ("lon", _dp), ("lat", _dp), ("p", _dp),
                ("f3", _fp * len(FIELDS_3D)), ("f2", _fp * len(FIELDS_2D))]


class OrcAtm(C.Structure):
    _fields_ = [("np", C.c_int), ("nq", C.c_int), ("time", _dp), ("p", _dp), ("lon", _dp),
                ("lat", _dp), ("q", _dp * NQ_MAX)]


class OrcCache(C.Structure):
    _fields_ = [("dt", _dp), ("rs", _dp), ("uvwp", _fp), ("rng_ctr", C.c_uint64),
                ("iso_var", _dp), ("iso_ts", _dp), ("iso_ps", _dp), ("iso_n", C.c_int),
                ("ip_global", C.POINTER(C.c_int64)), ("np_global", C.c_int64)]


class OrcZm(C.Structure):
    _fields_ = [("ntime", C.c_int), ("np", C.c_int), ("nlat", C.c_int), ("pad", C.c_int),
                ("time", _dp), ("p", _dp), ("lat", _dp), ("vmr", _dp)]


class OrcTs(C.Structure):
    _fields_ = [("ntime", C.c_int), ("pad", C.c_int), ("time", _dp), ("vmr", _dp)]


class OrcClim(C.Structure):
    _fields_ = [("tropo_ntime", C.c_int), ("tropo_nlat", C.c_int), ("tropo_time", C.c_double * 12),
                ("tropo_lat", C.c_double * 73), ("tropo", (C.c_double * 73) * 12), ("zm", OrcZm * len(ZONAL_MEANS)),
                ("ts", OrcTs * len(TRACER_SERIES))]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("mptrac_oracle.c", "mptrac_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU
    quota.  Inside a container os.cpu_count() is the whole host; an OpenMP team
    of that size on a 16-core quota spends its time being throttled."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        if "OMP_NUM_THREADS" not in os.environ:
            _lib.orc_set_num_threads(usable_cores())
        _lib.orc_sizeof_ctl.restype = C.c_size_t
        assert _lib.orc_sizeof_ctl() == C.sizeof(OrcCtl), "orc_ctl_t layout mismatch"
        _lib.orc_locate_irr.argtypes = [_dp, C.c_int, C.c_double]
        _lib.orc_locate_reg.argtypes = [_dp, C.c_int, C.c_double]
        _lib.orc_clim_tropo.restype = C.c_double
        _lib.orc_clim_tropo.argtypes = [C.POINTER(OrcClim), C.c_double, C.c_double]
        _lib.orc_sedi.restype = C.c_double
        _lib.orc_sedi.argtypes = [C.c_double] * 4
        _lib.orc_squares.restype = C.c_uint64
        _lib.orc_squares.argtypes = [C.c_uint64]
        _lib.orc_tropo_weight.restype = C.c_double
        _lib.orc_tropo_weight.argtypes = [C.POINTER(OrcCtl), C.POINTER(OrcClim)] + [C.c_double] * 3
        _lib.orc_module_rng.argtypes = [C.POINTER(OrcCtl), C.POINTER(OrcCache), C.c_size_t, C.c_int]
        _lib.orc_run_timestep.argtypes = [C.POINTER(OrcCtl), C.POINTER(OrcCache), C.POINTER(OrcClim),
                                          C.POINTER(OrcMet), C.POINTER(OrcMet), C.POINTER(OrcAtm),
                                          C.c_double]
        _lib.orc_module_mixing.argtypes = [C.POINTER(OrcCtl), C.POINTER(OrcClim), C.POINTER(OrcAtm),
                                           C.c_double]
        _lib.orc_module_timesteps.argtypes = [C.POINTER(OrcCtl), C.POINTER(OrcCache), C.POINTER(OrcMet),
                                              C.POINTER(OrcAtm), C.c_double]
        _lib.orc_grid_sums.argtypes = [C.POINTER(OrcCtl), C.POINTER(OrcAtm), C.c_double,
                                       C.POINTER(C.c_int), _dp, _dp]
        _lib.orc_grid_sums_kernel.argtypes = [C.POINTER(OrcCtl), C.POINTER(OrcAtm), C.c_double, C.c_int, _dp, _dp,
                                              C.POINTER(C.c_int), _dp, _dp]
        _lib.orc_intpol_met_time_3d.argtypes = [C.POINTER(OrcMet), C.POINTER(OrcMet), C.c_int] + \
            [C.c_double] * 4 + [_dp]
        _lib.orc_clim_ts.restype = C.c_double
        _lib.orc_clim_ts.argtypes = [C.POINTER(OrcTs), C.c_double]
        _lib.orc_clim_zm.restype = C.c_double
        _lib.orc_clim_zm.argtypes = [C.POINTER(OrcZm)] + [C.c_double] * 3
        _lib.orc_clim_oh.restype = C.c_double
        _lib.orc_clim_oh.argtypes = [C.POINTER(OrcCtl), C.POINTER(OrcClim)] + [C.c_double] * 4
        for fn, nargs in (("orc_rh", 3), ("orc_rhice", 3), ("orc_tdew", 2), ("orc_tice", 2), ("orc_theta", 2),
                          ("orc_zeta", 3), ("orc_lapse_rate", 2), ("orc_cos_sza", 3), ("orc_nat_temperature", 3)):
            getattr(_lib, fn).restype = C.c_double
            getattr(_lib, fn).argtypes = [C.c_double] * nargs
        _lib.orc_intpol_met_time_2d.argtypes = [C.POINTER(OrcMet), C.POINTER(OrcMet), C.c_int] + \
            [C.c_double] * 3 + [_dp]
    return _lib


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    """CPU mirror of the simulation state; same call names as the product's
    ``Simulation`` so parity tests read symmetrically."""

    def __init__(self, ctl_kw, clim, met0, met1, atm, rng_ctr=0, ip_global=None, np_global=None):
        """ip_global / np_global: `atm` is a subsample of a run with np_global particles, particle i being that
        run's particle ip_global[i] (random numbers follow the full run's slots; per-particle modules only)."""
        self.lib = lib()
        self.ctl = fill_ctl(OrcCtl(), **ctl_kw)
        time, lat, tropo = clim[:3]
        self.clim = OrcClim()
        # zonal-mean climatologies: optional fourth element {name: (time, p, lat, vmr[ntime][np][nlat])}
        self._zm = {}
        for name, tab in (clim[3] if len(clim) > 3 else {}).items():
            if name in TRACER_SERIES:      # a trace-gas time series (time, vmr)
                arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in tab]
                assert len(arrs) == 2 and arrs[0].shape == arrs[1].shape and len(arrs[0]) >= 2, name
                self._zm[name] = arrs
                z = self.clim.ts[TRACER_SERIES.index(name)]
                z.ntime = len(arrs[0])
                z.time, z.vmr = (_ptr(a, _dp) for a in arrs)
                continue
            arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in tab]
            assert arrs[3].shape == (len(arrs[0]), len(arrs[1]), len(arrs[2])), name
            self._zm[name] = arrs
            z = self.clim.zm[ZONAL_MEANS.index(name)]
            z.ntime, z.np, z.nlat = (len(a) for a in arrs[:3])
            z.time, z.p, z.lat, z.vmr = (_ptr(a, _dp) for a in arrs)
        self.clim.tropo_ntime, self.clim.tropo_nlat = len(time), len(lat)
        for i, v in enumerate(time):
            self.clim.tropo_time[i] = v
        for i, v in enumerate(lat):
            self.clim.tropo_lat[i] = v
        for i in range(len(time)):
            for j in range(len(lat)):
                self.clim.tropo[i][j] = tropo[i, j]
        self._mets = [None, None]
        self.met = [OrcMet(), OrcMet()]
        self.set_met(0, met0)
        self.set_met(1, met1)
        n = len(atm["time"])
        self.n = n
        self.time = np.array(atm["time"], dtype=np.float64)
        self.p = np.array(atm["p"], dtype=np.float64)
        self.lon = np.array(atm["lon"], dtype=np.float64)
        self.lat = np.array(atm["lat"], dtype=np.float64)
        self.q = np.array(atm["q"], dtype=np.float64).reshape(-1, n) if self.ctl.nq else np.zeros((0, n))
        assert self.q.shape[0] == self.ctl.nq
        self.atm = OrcAtm()
        self.atm.np, self.atm.nq = n, self.ctl.nq
        self.atm.time, self.atm.p = _ptr(self.time, _dp), _ptr(self.p, _dp)
        self.atm.lon, self.atm.lat = _ptr(self.lon, _dp), _ptr(self.lat, _dp)
        for iq in range(self.ctl.nq):
            self.atm.q[iq] = _ptr(self.q[iq], _dp)
        self.dt = np.zeros(n)
        self.rs = np.zeros(3 * n + 1)
        self.uvwp = np.zeros((n, 3), dtype=np.float32)
        self.iso_var = np.zeros(n)
        self.cache = OrcCache(_ptr(self.dt, _dp), _ptr(self.rs, _dp), _ptr(self.uvwp, _fp), rng_ctr,
                              _ptr(self.iso_var, _dp), None, None, 0, None, 0)
        if ip_global is not None:
            self._ip_global = np.ascontiguousarray(ip_global, dtype=np.int64)
            assert self._ip_global.shape == (n,) and np_global is not None and int(self._ip_global.max(initial=-1)) < np_global
            self.cache.ip_global = _ptr(self._ip_global, C.POINTER(C.c_int64))
            self.cache.np_global = int(np_global)

    def set_met(self, slot, met):
        self._mets[slot] = met      # keep arrays alive
        m = self.met[slot]
        m.time, m.coord_type, m.nx, m.ny, m.np = met.time, met.coord_type, met.nx, met.ny, met.np
        m.npl = met.npl
        m.lon, m.lat, m.p = _ptr(met.lon, _dp), _ptr(met.lat, _dp), _ptr(met.p, _dp)
        for i, k in enumerate(FIELDS_3D):
            m.f3[i] = _ptr(met.f3[k], _fp) if k in met.f3 else None
        for i, k in enumerate(FIELDS_2D):
            m.f2[i] = _ptr(met.f2[k], _fp) if k in met.f2 else None

    def set_balloon(self, ts, ps):
        """Balloon pressure time series of ISOSURF 4 (module_isosurf_init reads it from a file)."""
        self._iso_ts = np.ascontiguousarray(ts, dtype=np.float64)
        self._iso_ps = np.ascontiguousarray(ps, dtype=np.float64)
        self.cache.iso_ts, self.cache.iso_ps = _ptr(self._iso_ts, _dp), _ptr(self._iso_ps, _dp)
        self.cache.iso_n = len(self._iso_ts)

    def swap_met(self, new_met1):
        """mptrac_get_met's pointer swap (mptrac.c:6488-6491) + new met1."""
        old1 = self._mets[1]
        self.set_met(0, old1)
        self.set_met(1, new_met1)

    def timesteps_init(self):
        self.lib.orc_module_timesteps_init(C.byref(self.ctl), C.byref(self.atm))

    def run_timestep(self, t):
        self.lib.orc_run_timestep(C.byref(self.ctl), C.byref(self.cache), C.byref(self.clim),
                                  C.byref(self.met[0]), C.byref(self.met[1]), C.byref(self.atm),
                                  C.c_double(t))

    def module(self, name, t=None):
        """Call one orc_module_* by name with the reference's argument list."""
        L, c = self.lib, C.byref
        ctl, cache, clim, m0, m1, atm = (c(self.ctl), c(self.cache), c(self.clim), c(self.met[0]),
                                         c(self.met[1]), c(self.atm))
        if name == "timesteps":
            L.orc_module_timesteps(ctl, cache, m0, atm, C.c_double(t))
        elif name == "position":
            L.orc_module_position(cache, m0, m1, atm)
        elif name == "advect":
            L.orc_module_advect(ctl, cache, m0, m1, atm)
        elif name == "diff_turb":
            L.orc_module_diff_turb(ctl, cache, clim, m0, m1, atm)
        elif name == "diff_pbl":
            L.orc_module_diff_pbl(ctl, cache, m0, m1, atm)
        elif name == "diff_meso":
            L.orc_module_diff_meso(ctl, cache, m0, m1, atm)
        elif name == "convection":
            L.orc_module_convection(ctl, cache, m0, m1, atm)
        elif name == "sedi":
            L.orc_module_sedi(ctl, cache, m0, m1, atm)
        elif name == "decay":
            L.orc_module_decay(ctl, cache, clim, atm)
        elif name == "mixing":
            L.orc_module_mixing(ctl, clim, atm, C.c_double(t))
        elif name == "wet_depo":
            L.orc_module_wet_depo(ctl, cache, m0, m1, atm)
        elif name == "dry_depo":
            L.orc_module_dry_depo(ctl, cache, m0, m1, atm)
        elif name == "meteo":
            L.orc_module_meteo(ctl, clim, m0, m1, atm)
        elif name == "isosurf_init":
            L.orc_module_isosurf_init(ctl, cache, m0, m1, atm)
        elif name == "isosurf":
            L.orc_module_isosurf(ctl, cache, m0, m1, atm)
        elif name in ("bound_cond", "bound_cond2"):
            L.orc_module_bound_cond(ctl, cache, clim, m0, m1, atm)
        else:
            raise KeyError(name)

    def sort(self):
        keys = np.zeros(self.n)
        perm = np.zeros(self.n, dtype=np.int32)
        self.lib.orc_module_sort(C.byref(self.ctl), C.byref(self.met[0]), C.byref(self.atm),
                                 _ptr(keys, _dp), _ptr(perm, C.POINTER(C.c_int)))
        return keys, perm

    def grid_sums(self, t, kernel=None):
        """kernel = (kz, kw): the vertical weighting function of GRID_KERNEL (already normalised)"""
        ncell = self.ctl.grid_nx * self.ctl.grid_ny * self.ctl.grid_nz
        cnt = np.zeros(ncell, dtype=np.int32)
        mean = np.zeros((self.ctl.nq, ncell))
        sigma = np.zeros((self.ctl.nq, ncell))
        kz = np.ascontiguousarray(kernel[0] if kernel else [], dtype=np.float64)
        kw = np.ascontiguousarray(kernel[1] if kernel else [], dtype=np.float64)
        self.lib.orc_grid_sums_kernel(C.byref(self.ctl), C.byref(self.atm), C.c_double(t), len(kz), _ptr(kz, _dp),
                                      _ptr(kw, _dp), _ptr(cnt, C.POINTER(C.c_int)), _ptr(mean, _dp), _ptr(sigma, _dp))
        return cnt, mean, sigma

    def state(self):
        return {"time": self.time.copy(), "p": self.p.copy(), "lon": self.lon.copy(),
                "lat": self.lat.copy(), "q": self.q.copy(), "uvwp": self.uvwp.copy()}
