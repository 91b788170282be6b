"""Inputs that reproduce the reference's own test set-ups in memory.

``dd_test_case`` rebuilds the input of the reference's tests/dd_test/run.sh:
the solid-body-rotation wind field of the ``wind`` tool (formulae of
src/wind.c:124-169 with the parameters of run.sh:100-120, 368-392:
360 x 181 x 60 grid, WIND_U0 = WIND_U1 = 50 m/s, WIND_ALPHA = 90,
WIND_LAT_REVERSE = 1, W0 = 0) and the 144 start positions of the golden
00:00 file.  The reference reads the field back from netCDF and (in a non-DD
build) appends the periodic longitude column (src/mptrac.c:11714-11771); the
same is done here.
"""
import os

import numpy as np

from mptrac_amd.synth import Met, pressure_from_z

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def wind_tool_met(time, nx=360, ny=181, nz=60, z0=0.0, z1=60.0, u0=50.0, u1=50.0, alpha=90.0,
                  lat_reverse=True):
    lon = 360.0 / nx * np.arange(nx, dtype=np.float64)
    lat = 180.0 / (ny - 1) * np.arange(ny, dtype=np.float64) - 90.0
    if lat_reverse:
        lat = -lat
    iz = np.arange(nz, dtype=np.float64)
    p = pressure_from_z(z0 + (z1 - z0) / (nz - 1.0 - 0.0) * (iz - 0.0))
    rad = np.pi / 180.0
    la = (lat * rad)[None, :, None]
    lo = (lon * rad)[:, None, None]
    speed = (u0 + (u1 - u0) / (nz - 1.0) * iz)[None, None, :]
    u = speed * (np.cos(la) * np.cos(alpha * rad) + np.sin(la) * np.cos(lo) * np.sin(alpha * rad))
    v = -speed * np.sin(lo) * np.sin(alpha * rad) + 0.0 * la
    w = np.zeros((nx, ny, nz))
    t = np.full((nx, ny, nz), 280.0)
    ps = np.full((nx, ny), 1013.25)
    pbl = np.full((nx, ny), float(pressure_from_z(1.0)))

    def periodic(a):
        return np.concatenate([a, a[:1]], axis=0)

    lonp = np.concatenate([lon, [lon[-1] + lon[1] - lon[0]]])
    f3 = {"u": periodic(np.broadcast_to(u, (nx, ny, nz)).astype(np.float32)),
          "v": periodic(np.broadcast_to(v, (nx, ny, nz)).astype(np.float32)),
          "w": periodic(w.astype(np.float32)), "t": periodic(t.astype(np.float32))}
    f2 = {"ps": periodic(ps.astype(np.float32)), "pbl": periodic(pbl.astype(np.float32))}
    return Met(time, lonp, lat, p, f3, f2)


def read_tab(path):
    rows = [ln.split() for ln in open(path) if ln.strip() and not ln.startswith("#")]
    return np.array(rows, dtype=np.float64)


def dd_test_case():
    """Returns (ctl keywords, particle dict, t_start, golden {hour: table})."""
    gold = {h: read_tab(os.path.join(GOLD, "ref_dd_test", f"atm_2022_06_02_{h:02d}_00_00.tab"))
            for h in range(7)}
    g0 = gold[0]
    n = len(g0)
    atm = {"time": g0[:, 0].copy(), "p": pressure_from_z(g0[:, 1]), "lon": g0[:, 2].copy(),
           "lat": g0[:, 3].copy(), "q": np.ascontiguousarray(g0[:, 4:9].T.copy())}
    # tests/dd_test/data.ref/config.ctl
    ctl = dict(advect=2, advect_vert_coord=0, turb_dx_trop=0.0, turb_dx_strat=0.0, turb_dz_trop=0.0,
               turb_dz_strat=0.0, turb_mesox=0.0, turb_mesoz=0.0, direction=1,
               tdec_trop=259200.0, tdec_strat=259200.0, dt_mod=600.0, dt_met=3600.0,
               t_stop=707464800.00, nq=5, qnt_m=2)
    assert n == 144
    return ctl, atm, 707443200.00, gold


def fmt_g(x):
    """printf('%g') as used by write_atm_asc (src/mptrac.c:12840-12850)."""
    return float("%g" % x)
