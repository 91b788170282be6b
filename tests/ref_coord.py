"""The reference's tests/coord_test (Cartesian / UTM meteo grid, midpoint advection, turbulent and mesoscale
diffusion with the Squares generator, module_meteo output of t, u, v, w) as a golden case: its three
netCDF meteo files and its thirteen golden particle files are kept under tests/golden/ref_coord_test
(data files of the reference's own test, unchanged).  This module turns the netCDF files into met
snapshots the way the reference's reader does for the fields this configuration touches, and parses the
golden tables.

Reference behaviour restated here (file:line in /root/reference/src/mptrac.c):
  * read_met_nc_grid 9638-9822: x / y axes in metres for MET_COORD_TYPE 1, `plev` in Pa -> hPa, time from the
    file name;
  * read_met_nc_3d 10568-10610 / read_met_nc_2d 10255-10293: unpacked floats, value = scl * aux in single
    precision, _FillValue / missing_value / |aux| >= 1e14 -> NaN; scale 0.01f for w (Pa/s -> hPa/s) and sp;
  * read_met_extrapolate 9470-9506: below the lowest level with a non-finite t, u, v or w the column is filled
    downwards from the level above;
  * polar-wind fix, periodic columns, down-sampling, detrending: no-ops for this grid / these defaults.
The boundary-layer pressure of the reference comes out of its meteo preprocessing, which is outside this
repository's scope; with the default diffusivities (TURB_DX_PBL = TURB_DX_TROP, TURB_DZ_PBL = TURB_DZ_TROP = 0)
its value does not enter the tropospheric result, so a finite stand-in (surface pressure - 100 hPa) is used.
"""
import os

import numpy as np
from scipy.io import netcdf_file

from mptrac_amd.synth import Met

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_coord_test")
T0 = 799372800.0          # 2025-05-01 00:00 UTC in the reference's seconds since 2000-01-01 (first golden line)
CTL = dict(advect=2, dt_mod=600.0, dt_met=3600.0, diffusion=1, met_coord_type=1,
           met_utm_ref_lat=48.1507476,   # (MET_UTM_REF_LON 11.5692782 is not used on this path)
           met_dt_out=0.1, t_stop=T0 + 7200.0)
QUANTITIES = ("t", "u", "v", "w")
OUTPUTS = {600 * k: "atm_2025_05_01_%02d_%02d_00.tab" % (k // 6, 10 * (k % 6)) for k in range(13)}   # every 10 min


def _valid(a, var):
    fill = getattr(var, "_FillValue", None)
    miss = getattr(var, "missing_value", None)
    ok = np.abs(a) < np.float32(1e14)
    if fill is not None and fill != 0:
        ok &= a != np.float32(fill)
    if miss is not None and miss != 0:
        ok &= a != np.float32(miss)
    return ok


def load_met(hour):
    f = netcdf_file(os.path.join(HERE, "era5_utm32_2025_05_01_%02d.nc" % hour), "r", mmap=False)
    x = np.array(f.variables["x"][:], dtype=np.float64)
    y = np.array(f.variables["y"][:], dtype=np.float64)
    p = np.array(f.variables["plev"][:], dtype=np.float64) / 100.0
    assert np.all(np.diff(p) < 0), "pressure levels must be descending (mptrac.c:10153-10156)"

    def field3(name, scl):
        var = f.variables[name]
        a = np.array(var[:], dtype=np.float32)[0]                 # (lev, y, x)
        out = np.where(_valid(a, var), np.float32(scl) * a, np.float32(np.nan)).astype(np.float32)
        return np.ascontiguousarray(out.transpose(2, 1, 0))       # [ix][iy][ip]

    t, u, v, w = field3("t", 1.0), field3("u", 1.0), field3("v", 1.0), field3("w", 0.01)
    # read_met_extrapolate
    bad = ~(np.isfinite(t) & np.isfinite(u) & np.isfinite(v) & np.isfinite(w))
    for ix, iy in zip(*np.nonzero(bad.any(axis=2))):
        ip0 = np.nonzero(bad[ix, iy])[0].max()
        for a in (t, u, v, w):
            # a column without any valid level copies element [np] of the zero-initialised met_t array
            a[ix, iy, :ip0 + 1] = a[ix, iy, ip0 + 1] if ip0 + 1 < len(p) else np.float32(0.0)
    var = f.variables["sp"]
    a = np.array(var[:], dtype=np.float32)[0]                     # (y, x)
    ps = np.where(_valid(a, var), np.float32(0.01) * a, np.float32(np.nan)).astype(np.float32).T
    pbl = (ps - np.float32(100.0)).astype(np.float32)
    return Met(T0 + 3600.0 * hour, x, y, p, dict(u=u, v=v, w=w, t=t), dict(ps=ps, pbl=pbl), coord_type=1)


def read_atm_tab(seconds):
    rows = np.array([[float(c) for c in ln.split()] for ln in open(os.path.join(HERE, OUTPUTS[seconds]))
                     if ln.strip() and not ln.startswith("#")])
    return dict(time=rows[:, 0], z=rows[:, 1], x=rows[:, 2], y=rows[:, 3], q=rows[:, 4:8].T)


def initial_particles():
    """The particle file the reference's run started from is the text its own tools wrote with the same
    number formats as the first golden output (altitude %g, the others as printed), so the first golden
    file is the input, digit for digit."""
    g = read_atm_tab(0)
    n = len(g["time"])
    return {"time": g["time"].copy(), "p": 1013.25 * np.exp(-g["z"] / 7.0), "lon": g["x"].copy(), "lat": g["y"].copy(),
            "q": np.zeros((len(QUANTITIES), n))}


# print precision of the golden tables: x, y "%.2f", altitude and the quantities "%g" (six significant digits)
TOL_XY = 0.00501
TOL_REL = 5.01e-6


def run_against_golden(engine, mets):
    """Steps `engine` (the oracle or the HIP back end, same interface) through the two hours of the reference's
    test and returns the largest deviations from the thirteen golden tables (every ten minutes)."""
    import cases
    worst = dict(x=0.0, y=0.0, z=0.0, q=0.0)
    imet, seen = 0, 0
    for t in cases.step_times(engine.ctl):
        if t > mets[imet + 1].time:
            imet += 1
            engine.swap_met(mets[imet + 1])
        engine.run_timestep(t)
        sec = int(round(t - T0))
        if sec in OUTPUTS:
            g, s = read_atm_tab(sec), engine.state()
            assert np.array_equal(s["time"], g["time"])
            z = 7.0 * np.log(1013.25 / s["p"])
            worst["x"] = max(worst["x"], np.abs(s["lon"] - g["x"]).max())
            worst["y"] = max(worst["y"], np.abs(s["lat"] - g["y"]).max())
            worst["z"] = max(worst["z"], (np.abs(z - g["z"]) / np.abs(g["z"])).max())
            worst["q"] = max(worst["q"], (np.abs(s["q"] - g["q"]) / np.maximum(np.abs(g["q"]), 1e-300)).max())
            seen += 1
    assert seen == len(OUTPUTS)
    return worst
