"""Zonal-mean climatology tables for the tests: the HNO3 file of the reference (data/gozcards_HNO3.nc, the default
of CLIM_HNO3_FILENAME; copied unchanged to tests/golden/ref_data/) read the way read_clim_zm does
(mptrac.c:8747-8843), and an independent numpy statement of clim_zm (mptrac.c:414-466)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HNO3_FILE = os.path.join(HERE, "golden", "ref_data", "gozcards_HNO3.nc")
MONTH_MID = np.array([1209600.00, 3888000.00, 6393600.00, 9072000.00, 11664000.00, 14342400.00, 16934400.00,
                      19612800.00, 22291200.00, 24883200.00, 27561600.00, 30153600.00])


def load_zonal_mean(path=HNO3_FILE, var="HNO3"):
    """(time, p, lat, vmr[12][np][nlat]) with the gaps (negative entries) filled per column: the value of the
    highest level that holds one (the second of the reference's two loops overwrites the first)."""
    from scipy.io import netcdf_file
    f = netcdf_file(path, "r", mmap=False)
    p = np.asarray(f.variables["press"][:], dtype=np.float64)
    lat = np.asarray(f.variables["lat"][:], dtype=np.float64)
    vmr = np.asarray(f.variables[var][:], dtype=np.float64).copy()
    f.close()
    assert vmr.shape == (12, len(p), len(lat)) and p[0] > p[1] and lat[0] < lat[1]
    for it in range(12):
        for iy in range(len(lat)):
            col = vmr[it, :, iy]            # a view: a filled entry counts as valid for the ones above it
            for iz in range(len(p)):
                if col[iz] < 0:
                    valid = np.nonzero(col >= 0)[0]
                    if len(valid):
                        col[iz] = col[valid[-1]]
    return MONTH_MID.copy(), p, lat, vmr


def synthetic_zonal_mean(seed, np_=11, nlat=9, scale=1e-9):
    rng = np.random.default_rng(seed)
    p = 1000.0 * np.exp(-np.arange(np_) * 0.8)
    lat = np.linspace(-80.0, 80.0, nlat)
    return MONTH_MID.copy(), p, lat, scale * rng.uniform(0.1, 1.0, (12, np_, nlat))


def clim_zm(table, t, lat, p):
    """numpy restatement: clamp to the table, linear in pressure, latitude and time of year, never negative."""
    time, pp, ll, vmr = table
    sec = t - int(t / (365.25 * 86400.0)) * (365.25 * 86400.0)
    while sec < 0:
        sec += 365.25 * 86400.0
    p = min(max(p, pp[-1]), pp[0])
    lat = min(max(lat, ll[0]), ll[-1])
    it = int(np.clip(np.searchsorted(time, sec, side="right") - 1, 0, len(time) - 2))
    iy = int(np.clip(int((lat - ll[0]) / (ll[1] - ll[0])), 0, len(ll) - 2))
    iz = int(np.clip(np.searchsorted(-pp, -p, side="left") - 1, 0, len(pp) - 2))

    def lin(x0, y0, x1, y1, x):
        return y0 + (y1 - y0) / (x1 - x0) * (x - x0)
    a = [[lin(pp[iz], vmr[it + dt, iz, iy + dy], pp[iz + 1], vmr[it + dt, iz + 1, iy + dy], p) for dy in (0, 1)]
         for dt in (0, 1)]
    b = [lin(ll[iy], a[dt][0], ll[iy + 1], a[dt][1], lat) for dt in (0, 1)]
    return max(lin(time[it], b[0], time[it + 1], b[1], sec), 0.0)
