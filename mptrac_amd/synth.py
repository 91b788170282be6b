"""Synthetic ERA5-shaped meteorology and seeded particle sets.

Analytic fields on a lat/lon/pressure grid laid out like the reference's
``met_t`` arrays (``[ix][iy][ip]``, level index fastest, float32;
src/mptrac.h:3886-4012).  The longitude axis runs -180 ... +180 with the
periodic column appended, as ``read_met_periodic`` leaves it
(src/mptrac.c:11714-11771); the pressure axis is the log-pressure ladder of
the reference's ``wind`` tool (src/wind.c:129-130).  Nothing here is read from
a file and nothing depends on a seed except the particle positions.
"""
import numpy as np

P0 = 1013.25   # mptrac.h:305
H0 = 7.0       # mptrac.h:270

# order = MPHIP_U ... / MPHIP_PS ... of include/mptrac_hip.h
FIELDS_3D = ("u", "v", "w", "t", "lwc", "rwc", "iwc", "swc", "pl", "ul", "vl", "zetal", "zeta_dotl", "h2o",
             "z", "pv", "o3", "cc", "wl")
FIELDS_ML = ("pl", "ul", "vl", "zetal", "zeta_dotl", "wl")     # on model levels [nx][ny][npl]
FIELDS_2D = ("ps", "pbl", "cape", "cin", "pel", "pct", "pcb", "cl", "ess", "nss", "shf",
             "ts", "zs", "us", "vs", "lsm", "sst", "pt", "tt", "zt", "h2ot", "plcl", "plfc", "o3c")
# read by module_meteo only; generated on request (fields=...)
FIELDS_METEO_ONLY = ("z", "pv", "o3", "cc", "ts", "zs", "us", "vs", "lsm", "sst", "pt", "tt", "zt", "h2ot", "plcl",
                     "plfc", "o3c")

GRIDS = {
    # name: (NX without the periodic column, NY, NP)
    "C1": (360, 181, 60),
    "C2": (360, 181, 137),
    "C3": (720, 361, 137),
    "tiny": (36, 19, 20),
}


def pressure_from_z(z):
    """P(z), mptrac.h:1784."""
    return P0 * np.exp(-np.asarray(z, dtype=np.float64) / H0)


class Met:
    """One snapshot: axes (float64) + dict of float32 fields."""

    def __init__(self, time, lon, lat, p, f3, f2, coord_type=0):
        self.time = float(time)
        self.coord_type = int(coord_type)
        self.lon = np.ascontiguousarray(lon, dtype=np.float64)
        self.lat = np.ascontiguousarray(lat, dtype=np.float64)
        self.p = np.ascontiguousarray(p, dtype=np.float64)
        self.nx, self.ny, self.np = len(self.lon), len(self.lat), len(self.p)
        self.f3 = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in f3.items()}
        self.f2 = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in f2.items()}
        self.npl = self.np
        for k, v in self.f3.items():
            if k in FIELDS_ML:
                self.npl = v.shape[2]
                assert v.shape[:2] == (self.nx, self.ny), (k, v.shape)
            else:
                assert v.shape == (self.nx, self.ny, self.np), (k, v.shape)
        for k, v in self.f2.items():
            assert v.shape == (self.nx, self.ny), (k, v.shape)


def make_axes(nx0, ny, npl, lon0=-180.0, lat_reverse=False):
    lon = lon0 + np.arange(nx0 + 1, dtype=np.float64) * (360.0 / nx0)
    lat = -90.0 + np.arange(ny, dtype=np.float64) * (180.0 / (ny - 1))
    if lat_reverse:
        lat = -lat
    z = np.arange(npl, dtype=np.float64) * (60.0 / (npl - 1))
    return lon, lat, pressure_from_z(z)


def synthetic_met(grid="C1", time=0.0, amp=1.0, fields=None, lon0=-180.0, lat_reverse=False, model_levels=None):
    """Analytic snapshot.  ``amp`` scales the wind amplitudes so that two
    snapshots differ and the time interpolation is exercised.  ``model_levels``:
    number of model levels (met_t::npl; default: seven more than pressure levels)."""
    nx0, ny, npl = GRIDS[grid] if isinstance(grid, str) else grid
    lon, lat, p = make_axes(nx0, ny, npl, lon0, lat_reverse)
    want = set(FIELDS_3D + FIELDS_2D) - set(FIELDS_METEO_ONLY) if fields is None else set(fields)
    lam = np.deg2rad(lon)[:, None, None]
    phi = np.deg2rad(lat)[None, :, None]
    k = np.arange(npl, dtype=np.float64)[None, None, :]
    cphi = np.cos(phi)
    f3, f2 = {}, {}
    shape3 = (nx0 + 1, ny, npl)

    def put3(name, expr):
        if name in want:
            f3[name] = np.broadcast_to(expr, shape3).astype(np.float32)

    put3("u", 30.0 * amp * cphi * (1.0 + 0.1 * np.sin(0.3 * k)) + 0.0 * lam)
    put3("v", 5.0 * amp * np.sin(2.0 * lam) * cphi + 0.0 * k)
    put3("w", 1e-3 * amp * np.sin(lam) * cphi * np.sin(np.pi * k / (npl - 1)))
    put3("t", 280.0 - 0.5 * k * (60.0 / npl) + 5.0 * cphi + 2.0 * (amp - 1.0) * np.cos(lam))
    put3("lwc", 1e-5 * amp * np.maximum(0.0, np.sin(lam) * cphi) * (k < npl // 4))
    put3("rwc", 2e-6 * amp * np.maximum(0.0, np.cos(lam) * cphi) * (k < npl // 4))
    put3("iwc", 1e-6 * np.maximum(0.0, -np.sin(lam)) * cphi * (k < npl // 3) + 0.0 * k)
    put3("swc", 0.0 * lam * phi * k)
    put3("h2o", 1e-2 * np.exp(-k * (60.0 / npl) / 2.5) * cphi * (1.0 + 0.2 * amp * np.sin(lam)))
    put3("z", k * (60.0 / (npl - 1)) + 0.1 * amp * np.cos(lam) * cphi)
    put3("pv", 0.5 * amp * np.sin(phi) * np.exp(k * (60.0 / npl) / 10.0) + 0.0 * lam)
    put3("o3", 1e-8 + 8e-6 * np.exp(-((k * (60.0 / npl) - 25.0) / 8.0) ** 2) * (1.0 + 0.1 * amp * np.sin(lam) * cphi))
    put3("cc", np.clip(0.5 * amp * np.sin(lam) * cphi, 0.0, 1.0) * (k < npl // 3))

    if want & set(FIELDS_ML):
        # terrain-following model levels: p = sigma_k * ps (sigma 1 -> ~2e-4), zeta = a potential-temperature-like
        # monotonic function of p that also varies horizontally and in time, winds analytic in (lon, lat, level)
        npl_ml = npl + 7 if model_levels is None else int(model_levels)
        kk = np.arange(npl_ml, dtype=np.float64)[None, None, :]
        sigma = np.exp(-8.5 * kk / (npl_ml - 1))
        ps3 = (1013.25 - 30.0 * np.sin(lam) ** 2 * np.cos(phi))
        pl = sigma * ps3
        zeta = 290.0 * (1000.0 / pl) ** 0.286 * (1.0 + 0.02 * amp * np.cos(lam) * cphi)
        shape_ml = (nx0 + 1, ny, npl_ml)
        ml = {"pl": pl, "zetal": zeta,
              "ul": 25.0 * amp * cphi * (1.0 + 0.2 * np.sin(0.2 * kk)) + 0.0 * lam,
              "vl": 4.0 * amp * np.sin(2.0 * lam) * cphi + 0.0 * kk,
              "zeta_dotl": 2e-3 * amp * np.sin(lam) * cphi * np.sin(np.pi * kk / (npl_ml - 1)),
              "wl": 1.5e-3 * amp * np.sin(lam) * cphi * np.sin(np.pi * kk / (npl_ml - 1)) * (pl / 1000.0)}
        for name, expr in ml.items():
            if name in want:
                f3[name] = np.broadcast_to(expr, shape_ml).astype(np.float32)

    lam2, phi2 = lam[:, :, 0], phi[:, :, 0]
    shape2 = (nx0 + 1, ny)

    def put2(name, expr):
        if name in want:
            f2[name] = np.broadcast_to(expr, shape2).astype(np.float32)

    ps = 1013.25 - 30.0 * np.sin(lam2) ** 2 * np.cos(phi2)
    put2("ps", ps)
    put2("pbl", ps - (80.0 + 40.0 * amp * np.cos(lam2) * np.cos(phi2)))
    put2("cape", 200.0 * amp * (1.0 + np.sin(2.0 * lam2)) * np.cos(phi2))
    put2("cin", 10.0 + 0.0 * lam2 * phi2)
    put2("pel", 300.0 + 100.0 * np.sin(lam2) + 0.0 * phi2)
    put2("pct", 400.0 + 100.0 * np.cos(lam2) + 0.0 * phi2)
    put2("pcb", 800.0 + 0.0 * lam2 * phi2)
    put2("cl", 0.7 + 0.4 * amp * np.sin(lam2) * np.cos(phi2))
    # surface stresses [N/m^2] and sensible heat flux [W/m^2]: all three stability regimes occur
    put2("ess", 0.15 * amp * np.cos(lam2) * np.cos(phi2))
    put2("nss", 0.05 * np.sin(2.0 * lam2) + 0.0 * phi2)
    put2("shf", 60.0 * amp * np.sin(3.0 * lam2) * np.cos(phi2) * (np.abs(np.sin(7.0 * phi2)) > 0.3))
    # module_meteo-only surface fields; sst is undefined over land and plfc where there is no free convection
    # (NaN, as in the reference's data: exercises the nearest-neighbour branch of intpol_met_space_2d)
    lsm = (np.sin(3.0 * lam2) * np.cos(phi2) > 0.2).astype(np.float64)
    put2("ts", 288.0 + 12.0 * np.cos(phi2) + amp * np.sin(lam2))
    put2("zs", 1.5 * lsm * np.sin(lam2) ** 2 * np.cos(phi2))
    put2("us", 6.0 * amp * np.cos(phi2) + 0.0 * lam2)
    put2("vs", 2.0 * amp * np.sin(2.0 * lam2) * np.cos(phi2))
    put2("lsm", lsm)
    put2("sst", np.where(lsm > 0.5, np.nan, 290.0 + 10.0 * np.cos(phi2) + 0.5 * amp * np.cos(lam2)))
    put2("pt", 100.0 + 180.0 * np.abs(np.sin(phi2)) + 5.0 * amp * np.sin(lam2))
    put2("tt", 195.0 + 25.0 * np.abs(np.sin(phi2)) + 0.0 * lam2)
    put2("zt", 17.0 - 8.0 * np.abs(np.sin(phi2)) + 0.1 * amp * np.cos(lam2))
    put2("h2ot", 4e-6 + 2e-6 * amp * np.cos(lam2) * np.cos(phi2))
    put2("plcl", 900.0 + 50.0 * np.sin(lam2) * np.cos(phi2))
    put2("plfc", np.where(np.sin(2.0 * lam2) * np.cos(phi2) > 0.1, 800.0 + 60.0 * amp * np.cos(lam2), np.nan))
    put2("o3c", 300.0 + 60.0 * np.sin(phi2) + 0.0 * lam2)

    # periodic column is an exact copy of column 0 (mptrac.c:11726-11769)
    for d in (f3, f2):
        for a in d.values():
            a[-1] = a[0]
    return Met(time, lon, lat, p, f3, f2)


_LCG_A = 6364136223846793005
_LCG_C = 1442695040888963407
_M64 = 1 << 64


def lcg_skip(seed, k):
    """State of the LCG after k steps from `seed` (affine map composed by
    binary exponentiation, Python integers)."""
    A, C = 1, 0
    a, c = _LCG_A, _LCG_C
    while k:
        if k & 1:
            A, C = (a * A) % _M64, (a * C + c) % _M64
        a, c = (a * a) % _M64, (a * c + c) % _M64
        k >>= 1
    return (A * seed + C) % _M64


def lcg_uniform(n, seed=12345, skip=0):
    """Uniforms number skip .. skip+n-1 of the 64-bit LCG s <- s*a + c (top 53
    bits), vectorised by jump-ahead (all arithmetic wraps mod 2**64)."""
    a = np.uint64(_LCG_A)
    c = np.uint64(_LCG_C)
    start = np.uint64(lcg_skip(seed, skip))
    if n == 0:
        return np.zeros(0)
    with np.errstate(over="ignore"):
        an = np.cumprod(np.full(n, a, dtype=np.uint64))              # a^1 .. a^n
        geo = np.cumsum(np.concatenate(([np.uint64(1)], an[:-1])))    # 1+a+..+a^(k-1)
        s = an * start + c * geo
    return (s >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synthetic_particles(n, seed=12345, time=0.0, quantities=("m", "rp", "rhop"),
                        lon=(-180.0, 180.0), lat=(-85.0, 85.0), z=(0.5, 29.5), first=0):
    """Particles first .. first+n-1 of the seeded global set: lon/lat/z
    uniformly scattered; q[m] in [0.5, 1.5), q[rp]=1 um, q[rhop]=1000 kg/m3."""
    r = lcg_uniform(3 * n, seed, skip=3 * first).reshape(n, 3)
    atm = {
        "time": np.full(n, float(time)),
        "lon": lon[0] + (lon[1] - lon[0]) * r[:, 0],
        "lat": lat[0] + (lat[1] - lat[0]) * r[:, 1],
        "p": pressure_from_z(z[0] + (z[1] - z[0]) * r[:, 2]),
    }
    defaults = {"m": 1.0, "vmr": 1e-9, "rp": 1.0, "rhop": 1000.0, "ens": 0.0}
    q = np.zeros((len(quantities), n))
    for i, name in enumerate(quantities):
        q[i] = defaults.get(name, 0.0)
    if "m" in quantities:   # make masses distinguishable for mixing tests
        q[list(quantities).index("m")] = 0.5 + r[:, 0]
    atm["q"] = q
    for k in ("time", "lon", "lat", "p"):
        atm[k] = np.ascontiguousarray(atm[k], dtype=np.float64)
    return atm
