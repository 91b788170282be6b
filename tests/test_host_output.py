"""CPU tests of the host layer's outputs besides the ASCII / binary particle and grid files: the analysis writers
(CSI, ensembles, profiles, samples, station, VTK) against the golden files of the reference's tests/trac_test,
and the netCDF particle / grid files (classic format, own writer) against scipy's netCDF reader.

The writers are host code that works on downloaded particles -- no device is involved; the library is only
loaded."""
import os
import subprocess

import numpy as np
import pytest

from hostfiles import compile_c_test

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "golden", "ref_trac_test")
T_OBS = 360720000.0        # 2011-06-07 00:00 UTC: the time of the golden particle file and of every observation

# tests/trac_test/run.sh: the thirteen quantities (the two chemistry tracers, which this build would refuse to
# fill, are carried under neutral names -- the writers only print them) and the keys the writers read
QUANTITIES = ["t", "u", "v", "w", "zg", "pv", "ps", "pt", "m", "stat", "ens", "qa", "qb"]
GOLDEN_NAMES = {"qa": "Cccl3f", "qb": "Cx"}
KEYS = {"DT_MOD": "300.0", "SPECIES": "SO2", "OH_CHEM_REACTION": "0", "CSI_OBSMIN": "1e-5", "CSI_MODMIN": "1e-5",
        "SAMPLE_DZ": "100", "STAT_LON": "-22", "STAT_LAT": "-40", "VTK_STRIDE": "100"}


def _run(out_dir, extra, atm=os.path.join(REF, "atm_pl_2011_06_07_00_00_00.tab"), quantities=QUANTITIES, t=T_OBS):
    exe = compile_c_test("writers")
    args = [exe, atm, str(out_dir), repr(t), "NQ", str(len(quantities))]
    for i, q in enumerate(quantities):
        args += [f"QNT_NAME[{i}]", q]
        if q in GOLDEN_NAMES:
            args += [f"QNT_UNIT[{i}]", "ppv"]
    for k, v in {**KEYS, **extra}.items():
        args += [k, v]
    res = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "RESULT done" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
    return res.stdout


def _rows(path):
    """(header lines, numeric rows) of a text table; blank lines are kept as empty rows."""
    head, rows = [], []
    for line in open(path):
        if line.startswith("#"):
            head.append(line.rstrip("\n"))
        else:
            rows.append([float(x) for x in line.split()])
    return head, rows


def _golden_header(head):
    for ours, theirs in GOLDEN_NAMES.items():
        head = [h.replace(f"= {ours} ", f"= {theirs} ") for h in head]
    return head


def _close(a, b, rel):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.all(both_nan | (np.abs(a - b) <= rel * np.maximum(np.abs(b), 1e-300)) | (a == b))


def test_ensemble_file_matches_the_reference_golden(tmp_path):
    """write_ens on the golden particles = the golden ensemble file: same header, one row (quirk Q8: the
    accumulators are addressed by the slot of the quantity `ens`), every value to the precision the six printed
    digits of the input allow."""
    _run(tmp_path, {"ENS_BASENAME": "ens"})
    head, rows = _rows(tmp_path / "ens.tab")
    ghead, grows = _rows(os.path.join(REF, "ens_pl_2011_06_07_00_00_00.tab"))
    assert _golden_header(head) == ghead
    rows, grows = [r for r in rows if r], [r for r in grows if r]
    assert len(rows) == len(grows) == 1 and len(rows[0]) == len(grows[0]) == 4 + 2 * 13 + 1
    assert rows[0][0] == grows[0][0] and rows[0][-1] == grows[0][-1] == 10000
    assert _close(rows[0], grows[0], 2e-5), (rows[0], grows[0])


def test_vtk_file_matches_the_reference_golden(tmp_path):
    """write_vtk on the golden particles (every 100th) = the golden VTK file, line for line (%g of the six-digit
    inputs reproduces them; only the altitude goes through Z(P(z)))."""
    _run(tmp_path, {"VTK_BASENAME": "cloud"})
    ours = open(tmp_path / "cloud.vtk").read().split("\n")
    gold = open(os.path.join(REF, "atm_pl_00003.vtk")).read().split("\n")
    for a, b in GOLDEN_NAMES.items():
        ours = [ln.replace(f"SCALARS {a} ", f"SCALARS {b} ") for ln in ours]
    assert len(ours) == len(gold)
    for a, b in zip(ours, gold):
        if a == b:
            continue
        fa, fb = [float(x) for x in a.split()], [float(x) for x in b.split()]      # (a last digit of the altitude)
        assert len(fa) == len(fb) and _close(fa, fb, 2e-5), (a, b)
    assert sum(a == b for a, b in zip(ours, gold)) > 0.95 * len(gold)


def test_csi_file_matches_the_reference_golden(tmp_path):
    """write_csi at the observation time = the golden row: contingency counts exactly, scores and error
    statistics (own Pearson / Spearman / error norms instead of GSL's) to input precision."""
    _run(tmp_path, {"CSI_BASENAME": "csi", "CSI_OBSFILE": os.path.join(REF, "obs.tab")})
    head, rows = _rows(tmp_path / "csi.tab")
    ghead, grows = _rows(os.path.join(REF, "csi_pl.tab"))
    assert head == ghead
    rows, grows = [r for r in rows if r], [r for r in grows if r]
    assert len(grows) == 1 and len(rows) == 1
    r, g = rows[0], grows[0]
    assert r[:7] == g[:7] and r[-1] == g[-1], (r, g)      # time, member, hits, misses, false alarms, n_obs, n_for, points
    assert _close(r[7:-1], g[7:-1], 5e-5), (r, g)


def test_sample_file_matches_the_reference_golden_where_no_meteo_data_enter(tmp_path):
    """write_sample: one row per observation -- particle counts and column densities of the golden file (they
    depend on the particles alone); the volume mixing ratio needs the run's temperature field, so it is only
    compared where it is zero."""
    _run(tmp_path, {"SAMPLE_BASENAME": "sample", "SAMPLE_OBSFILE": os.path.join(REF, "obs.tab")})
    head, rows = _rows(tmp_path / "sample.tab")
    ghead, grows = _rows(os.path.join(REF, "sample_pl.tab"))
    assert head == ghead
    rows, grows = np.array([r for r in rows if r]), np.array([r for r in grows if r])
    assert rows.shape == grows.shape and rows.shape[0] == 546
    assert np.array_equal(rows[:, [0, 1, 2, 3, 5, 6, 9]], grows[:, [0, 1, 2, 3, 5, 6, 9]])
    assert _close(rows[:, 4], grows[:, 4], 1e-6) and _close(rows[:, 7], grows[:, 7], 2e-5)
    assert grows[:, 6].sum() > 50 and np.array_equal(rows[:, 8] == 0, grows[:, 8] == 0)
    with_mass = grows[:, 8] > 0      # constant 250 K instead of the run's temperatures: the right magnitude
    assert np.all(np.abs(rows[with_mass, 8] / grows[with_mass, 8] - 1) < 0.3)


def test_profile_file_matches_the_reference_golden_where_no_meteo_data_enter(tmp_path):
    """write_prof: the same profiles (columns with observations and particle mass) at the same levels with the
    same observation means and counts as the golden file; zero mixing ratio where the golden one is zero."""
    _run(tmp_path, {"PROF_BASENAME": "prof", "PROF_OBSFILE": os.path.join(REF, "obs.tab")})
    head, rows = _rows(tmp_path / "prof.tab")
    ghead, grows = _rows(os.path.join(REF, "prof_pl.tab"))
    assert head == ghead
    assert [bool(r) for r in rows] == [bool(r) for r in grows]          # the same blocks
    rows, grows = np.array([r for r in rows if r]), np.array([r for r in grows if r])
    assert rows.shape == grows.shape and rows.shape[0] == 3000
    assert np.array_equal(rows[:, [0, 1, 2, 3, 4, 9, 10]], grows[:, [0, 1, 2, 3, 4, 9, 10]])
    assert np.array_equal(rows[:, 6] == 0, grows[:, 6] == 0)
    assert np.all(rows[:, 5] == 250.0) and np.all(rows[:, 7] == 1e-5) and np.all(rows[:, 8] == 1e-6)


def test_station_file_lists_every_particle_once(tmp_path):
    """write_station: the particles within STAT_R of the station, each with all quantities in the particle
    file's own digits, and with a `stat` quantity each particle at most once."""
    out = _run(tmp_path, {"STAT_BASENAME": "station", "STAT_R": "800"})
    assert "RESULT done 10000" in out
    head, rows = _rows(tmp_path / "station.tab")
    rows = np.array([r for r in rows if r])
    ahead, arows = _rows(os.path.join(REF, "atm_pl_2011_06_07_00_00_00.tab"))
    atm = np.array([r for r in arows if r])
    assert head[4:] == ahead[4:] or _golden_header(head)[4:] == ahead[4:]
    # great-circle chord distance to the station, as the writer measures it
    def xyz(lon, lat):
        lam, phi = np.deg2rad(lon), np.deg2rad(lat)
        return 6367.421 * np.stack([np.cos(phi) * np.cos(lam), np.cos(phi) * np.sin(lam), np.sin(phi)], axis=-1)
    d2 = ((xyz(atm[:, 2], atm[:, 3]) - xyz(-22.0, -40.0)) ** 2).sum(axis=1)
    inside = (d2 <= 800.0 ** 2) & (atm[:, 4 + QUANTITIES.index("stat")] == 0)      # (not listed by the run before)
    assert (d2 <= 800.0 ** 2).sum() > inside.sum() > 0 and inside.sum() == len(rows)
    want = atm[inside].copy()
    want[:, 4 + QUANTITIES.index("stat")] = 1.0        # the flag is set before the line is printed (as in the reference)
    assert _close(rows[:, 2:], want[:, 2:], 1e-12) and _close(rows[:, 1], want[:, 1], 2e-6)


@pytest.mark.parametrize("atm_type_out", [2, 4])
def test_netcdf_particle_files_round_trip_and_are_readable_by_scipy(tmp_path, atm_type_out):
    """ATM_TYPE_OUT 2 (netCDF: dimension obs; time, press, lon, lat, one variable per quantity with long_name and
    units) and 4 (CLaMS position file: NPARTS; LAT, LON, PRESS, ZETA, quantities [time][NPARTS]): written by the
    host layer's own classic-netCDF writer, read back by its reader bit for bit, and readable by an independent
    implementation of the format (scipy.io.netcdf_file) with the same values."""
    from scipy.io import netcdf_file
    quantities = ["m", "zeta_d", "ens"]
    # a small text particle file
    rng = np.random.default_rng(5)
    n = 1234
    cols = np.column_stack([np.full(n, 1000.0), rng.uniform(0.5, 30, n), rng.uniform(-180, 180, n),
                            rng.uniform(-89, 89, n), rng.uniform(0, 5, n), rng.uniform(300, 2000, n),
                            rng.integers(0, 4, n).astype(float)])
    src = tmp_path / "in.tab"
    np.savetxt(src, cols, fmt="%.17g")
    out = _run(tmp_path, {"ATM_BASENAME": "out.nc", "ATM_TYPE_OUT": str(atm_type_out)}, atm=str(src),
               quantities=quantities, t=1000.0)
    assert "RESULT roundtrip identical" in out
    f = netcdf_file(str(tmp_path / "out.nc"), "r", mmap=False)
    names = {2: ("obs", "lon", "lat", "press"), 4: ("NPARTS", "LON", "LAT", "PRESS")}[atm_type_out]
    assert f.dimensions[names[0]] == n
    assert np.array_equal(f.variables[names[1]][:], cols[:, 2]) and np.array_equal(f.variables[names[2]][:], cols[:, 3])
    press = 1013.25 * np.exp(-cols[:, 1] / 7.0)
    assert np.allclose(f.variables[names[3]][:], press, rtol=1e-15)
    m = f.variables["m"]
    assert np.array_equal(np.asarray(m[:]).reshape(-1), cols[:, 4]) and m.units == b"kg"
    if atm_type_out == 2:
        assert m.long_name == b"mass" and f.featureType == b"point"
        assert np.array_equal(f.variables["time"][:], cols[:, 0])
    else:
        assert f.model == b"MPTRAC" and m.shape == (1, n)
        assert np.array_equal(f.variables["ZETA"][:], cols[:, 5])
    f.close()


def test_clams_trajectory_file_grows_by_one_record_per_output(tmp_path):
    """ATM_TYPE_OUT 3: traj_fix_3d_<start>_<stop>.nc with an unlimited time dimension -- one record per call --
    and, at the stop time, init_fix_<stop>.nc (a position file)."""
    from scipy.io import netcdf_file
    rng = np.random.default_rng(7)
    n = 300
    cols = np.column_stack([np.full(n, 86400.0), rng.uniform(1, 20, n), rng.uniform(-180, 180, n),
                            rng.uniform(-80, 80, n), rng.uniform(0, 5, n), rng.uniform(300, 900, n)])
    src = tmp_path / "in.tab"
    np.savetxt(src, cols, fmt="%.17g")
    _run(tmp_path, {"ATM_BASENAME": "ignored", "ATM_TYPE_OUT": "3"}, atm=str(src), quantities=["m", "zeta_d"], t=86400.0)
    # 2000-01-02 00:00 is start and stop of the one-call run
    traj = netcdf_file(str(tmp_path / "traj_fix_3d_00010200_00010200.nc"), "r", mmap=False)
    assert traj.dimensions["time"] is None and traj.dimensions["NPARTS"] == n and traj.dimensions["TMDT"] == 7
    assert traj.variables["LAT"].shape == (1, n) and np.array_equal(traj.variables["LAT"][0], cols[:, 3])
    assert np.array_equal(traj.variables["m"][0], cols[:, 4]) and traj.variables["time"][0] == 86400.0
    traj.close()
    init = netcdf_file(str(tmp_path / "init_fix_00010200.nc"), "r", mmap=False)
    assert np.array_equal(init.variables["LON"][:], cols[:, 2]) and np.array_equal(init.variables["ZETA"][:], cols[:, 5])
    init.close()


@pytest.mark.parametrize("coord_type", [0, 1])
def test_meteo_snapshot_as_netcdf(tmp_path, coord_type):
    """mptrac_write_met with MET_TYPE 0: the reference's variable names, dimension order ([time][lev][lat][lon]),
    units and scalings (pressures in Pa, w in Pa/s, humidity and ozone as mass mixing ratios, surface height as
    geopotential), readable by scipy; on a Cartesian grid the host layer's own reader returns the snapshot
    (to the 1e-6 the float scalings allow).  (tests/c/met_nc.c)"""
    from scipy.io import netcdf_file
    exe = compile_c_test("met_nc")
    res = subprocess.run([exe, str(tmp_path), str(coord_type)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT written")][-1].split()
    t, ps23, w123, h2o640 = (float(x) for x in line[2:])
    if coord_type == 1:
        assert "RESULT readback same" in res.stdout
    f = netcdf_file(str(tmp_path / "met_2001_02_03_04.nc"), "r", mmap=False)
    xname, yname = ("lon", "lat") if coord_type == 0 else ("x", "y")
    assert f.dimensions[xname] == 7 and f.dimensions[yname] == 5 and f.dimensions["lev"] == 4 and f.dimensions["time"] == 1
    assert f.variables["time"][0] == t and np.array_equal(f.variables["lev"][:], [100000.0, 80000.0, 60000.0, 40000.0])
    assert f.variables["sp"].shape == (1, 5, 7) and f.variables["w"].shape == (1, 4, 5, 7)
    assert f.variables["sp"].units == b"Pa" and f.variables["q"].long_name == b"Specific humidity"
    assert f.variables["sp"][0, 3, 2] == np.float32(100.0) * np.float32(ps23)
    assert f.variables["w"][0, 3, 2, 1] == np.float32(100.0) * np.float32(w123)
    assert f.variables["q"][0, 0, 4, 6] == np.float32(18.01528 / 28.9644) * np.float32(h2o640)
    assert len(f.variables) == 4 + 24 + 11
    f.close()


INTEROP = os.path.join(HERE, "golden", "ref_interoper_test")


def _atm_conv(args, zeta_coordinate=False):
    # the atm_conv tool of this build; the readers' diabatic set-up (which a run would refuse) through the test
    # program tests/c/atm_conv.c, the same code with one switch
    from mptrac_amd import build
    build.build_host()
    exe = compile_c_test("atm_conv") if zeta_coordinate else build.ATM_CONV_BIN
    env = dict(os.environ, ATM_CONV_ZETA_COORDINATE="1") if zeta_coordinate else dict(os.environ)
    res = subprocess.run([exe, "-"] + [str(a) for a in args], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0 and ("RESULT converted" in res.stdout or not zeta_coordinate), res.stdout[-3000:] + res.stderr[-2000:]
    return res.stdout


def test_clams_position_file_round_trip_of_the_reference_interoper_test(tmp_path):
    """tests/interoper_test/run.sh:19-20 of the reference: its particle file without quantities -> CLaMS position
    file -> text file, byte for byte the reference's golden atm_output.tab (times collapse to the first particle's,
    pressures survive the file, six printed digits)."""
    out = _atm_conv([os.path.join(INTEROP, "atm_input.tab"), 0, tmp_path / "atm_output.nc", 4])
    assert "ZETA of the position file is not a vertical coordinate" in out
    _atm_conv([tmp_path / "atm_output.nc", 4, tmp_path / "atm_output.tab", 0])
    with open(tmp_path / "atm_output.tab", "rb") as a, open(os.path.join(INTEROP, "atm_output.tab"), "rb") as b:
        assert a.read() == b.read()
    from scipy.io import netcdf_file
    f = netcdf_file(str(tmp_path / "atm_output.nc"), "r", mmap=False)
    assert f.dimensions["NPARTS"] == 10000 and np.array_equal(f.variables["ZETA"][:], f.variables["LAT"][:])
    f.close()


def test_clams_init_file_of_the_reference_interoper_test(tmp_path):
    """The CLaMS init file the reference's diabatic test starts from (tests/interoper_test/data.ref/init, classic
    netCDF: TIME_INIT, LAT, LON, ZETA, no PRESS) read with ATM_TYPE 3 and ZETA as the vertical coordinate: times,
    longitudes, latitudes and zeta are the columns the reference printed at the start time of that run
    (atm_2016_07_01_00_00_00.tab: no parcel has moved at t = T_START; its altitudes come from the meteo data)."""
    quantities = ["theta", "pv", "m", "zeta", "zeta_d", "ps", "p"]
    args = [os.path.join(INTEROP, "pos_glo_16070100.nc"), 3, tmp_path / "pos.tab", 0, "NQ", len(quantities)]
    for i, q in enumerate(quantities):
        args += [f"QNT_NAME[{i}]", q]
    _atm_conv(args, zeta_coordinate=True)
    mine = np.loadtxt(tmp_path / "pos.tab")
    gold = np.loadtxt(os.path.join(INTEROP, "atm_2016_07_01_00_00_00.tab"))
    assert mine.shape == gold.shape == (5662, 4 + len(quantities))
    for col in (0, 2, 3, 4 + quantities.index("m"), 4 + quantities.index("zeta")):
        assert np.array_equal(mine[:, col], gold[:, col]), col


@pytest.mark.parametrize("atm_type", [1, 2], ids=["binary", "netcdf"])
def test_particle_file_conversions_of_the_reference_atm_test(tmp_path, atm_type):
    """tests/atm_test/run.sh:37-43 of the reference: its particle file (10000 parcels: aoa, m, vmr) converted to the
    binary / netCDF format and back to text is the file itself, byte for byte."""
    gold = os.path.join(HERE, "golden", "ref_atm_test", "atm_2000_01_01_00_00_00.tab")
    qnt = ["NQ", 3, "QNT_NAME[0]", "aoa", "QNT_NAME[1]", "m", "QNT_NAME[2]", "vmr"]
    packed = tmp_path / ("atm.bin" if atm_type == 1 else "atm.nc")
    _atm_conv([gold, 0, packed, atm_type] + qnt)
    _atm_conv([packed, atm_type, tmp_path / "back.tab", 0] + qnt)
    with open(tmp_path / "back.tab", "rb") as a, open(gold, "rb") as b:
        assert a.read() == b.read()


def _clim_tables(tmp_path, keys):
    exe = compile_c_test("clim_zm")
    args = [exe, str(tmp_path / "clim.txt")]
    for k, v in keys.items():
        args += [k, str(v)]
    res = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "RESULT done" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
    lines = open(tmp_path / "clim.txt").read().splitlines()
    out = {}
    for k in range(25, len(lines), 3):      # the five time series behind the five tables
        out[lines[k].split()[0]] = (np.array(lines[k + 1].split(), dtype=np.float64),
                                    np.array(lines[k + 2].split(), dtype=np.float64))
    for k in range(0, 25, 5):
        name, nt, npr, nlat = lines[k].split()
        nt, npr, nlat = int(nt), int(npr), int(nlat)
        axes = [np.array(lines[k + 1 + j].split(), dtype=np.float64) for j in range(3)]
        out[name] = (axes[0], axes[1], axes[2], np.array(lines[k + 4].split(), dtype=np.float64).reshape(nt, npr, nlat))
    return out, res.stdout


def test_zonal_mean_climatology_reader_on_the_reference_hno3_file(tmp_path):
    """mptrac_read_clim with a quantity that needs the HNO3 climatology: the reference's data/gozcards_HNO3.nc
    (classic netCDF, single precision, 2164 gaps) gives the table an independent reading of the file gives
    (tests/refclim.py: monthly mid-points, gaps filled from the highest valid level of their column); tables no
    quantity asks for are not read, a missing file is a warning."""
    import refclim
    keys = {"NQ": 2, "QNT_NAME[0]": "tnat", "QNT_NAME[1]": "h2o", "CLIM_HNO3_FILENAME": refclim.HNO3_FILE}
    tabs, _ = _clim_tables(tmp_path, keys)
    want = refclim.load_zonal_mean()
    for got, ref in zip(tabs["hno3"], want):
        assert np.array_equal(got, ref)
    assert all(tabs[k][3].size == 0 for k in ("oh", "h2o2", "ho2", "o1d"))
    keys.update({"NQ": 3, "QNT_NAME[2]": "ho2", "CLIM_HO2_FILENAME": str(tmp_path / "nothing.nc")})
    tabs, log = _clim_tables(tmp_path, keys)
    assert "HO2 climatology data are missing" in log and tabs["ho2"][3].size == 0 and tabs["hno3"][3].size == 12 * 25 * 18


def test_oh_climatology_with_diurnal_correction(tmp_path):
    """OH_CHEM_BETA > 0 (clim_oh_diurnal_correction, mptrac.c:122-152): the table is divided by the mean over 360
    longitudes of exp(-beta / max(cos(sza), cos 85 deg)) at each month and latitude.  Table: a classic netCDF file
    made here with the variable names of the reference's radical climatology."""
    from scipy.io import netcdf_file
    import refclim
    _, p, lat, vmr = refclim.synthetic_zonal_mean(21, scale=1e-13)
    path = str(tmp_path / "radicals.nc")
    f = netcdf_file(path, "w")
    for name, n in (("time", 12), ("press", len(p)), ("lat", len(lat))):
        f.createDimension(name, n)
    f.createVariable("press", "f", ("press",))[:] = p
    f.createVariable("lat", "i", ("lat",))[:] = lat.astype(np.int32)
    f.createVariable("OH", "f", ("time", "press", "lat"))[:] = vmr
    f.close()
    keys = {"NQ": 1, "QNT_NAME[0]": "oh", "CLIM_OH_FILENAME": path}
    plain, _ = _clim_tables(tmp_path, keys)
    assert np.array_equal(plain["oh"][3], vmr.astype(np.float32).astype(np.float64))
    assert np.array_equal(plain["oh"][1], p.astype(np.float32).astype(np.float64)) and np.array_equal(plain["oh"][2], lat)
    beta = 0.6
    scaled, _ = _clim_tables(tmp_path, dict(keys, OH_CHEM_BETA=beta))
    from oracle import binding as B
    L = B.lib()
    thresh = np.cos(np.deg2rad(85.0))
    for it in (0, 5, 11):
        for iy in (0, 4, 8):
            c = np.array([L.orc_cos_sza(refclim.MONTH_MID[it], float(lon), float(lat[iy])) for lon in range(-180, 180)])
            factor = np.mean(np.exp(-beta / np.maximum(c, thresh)))
            assert np.allclose(scaled["oh"][3][it, :, iy], plain["oh"][3][it, :, iy] / factor, rtol=1e-13, atol=0)


def test_trace_gas_time_series_reader_on_the_reference_files(tmp_path):
    """mptrac_read_clim with trace-gas quantities: the reference's data/noaa_gml_sf6.tab and noaa_gml_n2o.tab
    (text, "year vmr") become (seconds since 2000, vmr) series; a series whose gas is not carried is not read, "-"
    switches one off."""
    ref = os.path.join(HERE, "golden", "ref_data")
    keys = {"NQ": 2, "QNT_NAME[0]": "Csf6", "QNT_NAME[1]": "Cn2o", "CLIM_SF6_TIMESERIES": os.path.join(ref, "noaa_gml_sf6.tab"),
            "CLIM_N2O_TIMESERIES": os.path.join(ref, "noaa_gml_n2o.tab")}
    tabs, _ = _clim_tables(tmp_path, keys)
    for name, fn in (("sf6", "noaa_gml_sf6.tab"), ("n2o", "noaa_gml_n2o.tab")):
        raw = np.loadtxt(os.path.join(ref, fn))
        assert np.array_equal(tabs[name][0], (raw[:, 0] - 2000.0) * 365.25 * 86400.0) and np.array_equal(tabs[name][1], raw[:, 1])
        assert len(raw) > 100
    assert all(len(tabs[k][0]) == 0 for k in ("ccl4", "ccl3f", "ccl2f2"))
    tabs, _ = _clim_tables(tmp_path, dict(keys, CLIM_N2O_TIMESERIES="-"))
    assert len(tabs["n2o"][0]) == 0 and len(tabs["sf6"][0]) > 100


def test_netcdf4_particle_files_of_the_reference_dd_test(tmp_path):
    """The nine init.nc files of the reference's tests/dd_test (netCDF-4 / HDF5: superblock version 2, links and
    attributes in the object headers, contiguous little-endian doubles -- written by the reference's CLaMS
    writer) read with ATM_TYPE 4 by the host layer's own HDF5 reader: converted to text, the 144 parcels are the
    rows of the golden particle file of the start time (the run has not moved them yet; columns time, altitude,
    longitude, latitude, idx, zeta -- absent from the file under that name, hence 0 as in the golden -- and m)."""
    gold = np.loadtxt(os.path.join(HERE, "golden", "ref_dd_test", "atm_2022_06_02_00_00_00.tab"))
    rows = []
    for k in range(9):
        out = tmp_path / f"atm_{k}.tab"
        _atm_conv([os.path.join(HERE, "golden", "ref_dd_test", "init", f"data.{k}.nc"), 4, out, 0, "NQ", 3,
                   "QNT_NAME[0]", "idx", "QNT_UNIT[0]", "-", "QNT_NAME[1]", "zeta", "QNT_NAME[2]", "m"])
        rows.append(np.loadtxt(out).reshape(-1, 7))
    mine = np.concatenate(rows)
    assert mine.shape == (144, 7) and gold.shape[0] == 144
    mine = mine[np.argsort(mine[:, 4])]
    gold = gold[np.argsort(gold[:, 4])]
    assert np.array_equal(mine, gold[:, :7])


def test_netcdf4_climatology_file_of_the_reference(tmp_path):
    """data/cams_H2O2.nc of the reference (netCDF-4 / HDF5 with a version-0 superblock; the default of
    CLIM_H2O2_FILENAME) through mptrac_read_clim: 12 months x 25 levels x 241 latitudes of mixing ratios on
    the axes the file declares (1000 ... 1 hPa, -90 ... 90 degrees in steps of 0.75)."""
    keys = {"NQ": 1, "QNT_NAME[0]": "h2o2", "CLIM_H2O2_FILENAME": os.path.join(HERE, "golden", "ref_data", "cams_H2O2.nc")}
    tabs, _ = _clim_tables(tmp_path, keys)
    time, p, lat, vmr = tabs["h2o2"]
    assert vmr.shape == (12, 25, 241) and p[0] == 1000.0 and p[-1] == 1.0 and np.all(np.diff(p) < 0)
    assert np.array_equal(lat, -90.0 + 0.75 * np.arange(241))
    assert vmr.min() >= 0 and (vmr == 0).sum() < 50 and vmr.max() < 1e-8 and np.isfinite(vmr).all()   # (zeros: polar night)
    # (double precision in the file: 12 * 25 * 241 * 8 bytes of its 588720 are this array)
    # smooth in latitude: neighbouring columns differ by far less than the field varies
    assert np.abs(np.diff(vmr, axis=2)).max() < 0.2 * vmr.max()


def test_met_conv_netcdf_to_binary_and_back(tmp_path):
    """The met_conv tool of this build on a meteo file of the reference's coord_test (classic netCDF, UTM grid):
    netCDF -> MET_TYPE 1 binary -> netCDF; the level fields the model reads come back as they are in the original
    (read with scipy) wherever that holds data, the surface fields to single precision (Pa -> hPa -> Pa; missing values as NaN)."""
    import subprocess
    from scipy.io import netcdf_file
    from mptrac_amd import build
    build.build_host()
    src = os.path.join(HERE, "golden", "ref_coord_test", "era5_utm32_2025_05_01_00.nc")
    keys = ["MET_COORD_TYPE", "1", "MET_UTM_REF_LON", "11.5692782", "MET_UTM_REF_LAT", "48.1507476", "MET_CAPE", "0"]
    binf, back = str(tmp_path / "era5_utm32_2025_05_01_00.bin"), str(tmp_path / "back_2025_05_01_00.nc")
    for a, ta, b, tb in ((src, 0, binf, 1), (binf, 1, back, 0)):
        r = subprocess.run([build.MET_CONV_BIN, "-", a, str(ta), b, str(tb)] + keys, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    f, g = netcdf_file(src, "r", mmap=False), netcdf_file(back, "r", mmap=False)
    valid = np.asarray(f.variables["t"][:]) > -1e30          # below the ground the file holds a missing value ...
    assert 0 < (~valid).sum() < valid.size // 2
    for name in ("t", "u", "v"):
        a, b = np.asarray(g.variables[name][:]), np.asarray(f.variables[name][:])
        assert np.array_equal(a[valid], b[valid]), name
        assert np.isfinite(a).all() and np.abs(a).max() < 1e4, name   # ... which the reader replaces (read_met_extrapolate)
    sp0, sp1 = np.asarray(f.variables["sp"][:]), np.asarray(g.variables["sp"][:])
    assert np.allclose(sp1[sp0 > -1e30], sp0[sp0 > -1e30], rtol=3e-7, atol=0) and np.isnan(sp1[sp0 <= -1e30]).all()
    assert np.allclose(np.asarray(g.variables["w"][:])[valid], np.asarray(f.variables["w"][:])[valid], rtol=3e-7, atol=1e-12)
    assert np.array_equal(g.variables["x"][:], f.variables["x"][:]) and np.allclose(g.variables["lev"][:], f.variables["plev"][:], rtol=1e-15)
    f.close()
    g.close()


def test_met_bin_reader_threads_give_the_same_file(tmp_path):
    """MET_TYPE 1 level fields are read by several threads side by side (one pread per longitude slab,
    MPTRAC_AMD_IO_THREADS; mptrac_amd/host/mptrac.c: met_slab_main): a file read with 1, 3 and 8 threads and written
    back is the same file, byte for byte -- with the limits the reference applies on reading (negative humidities and
    cloud fractions above one are cut, mptrac.c:9172-9177) -- and a file cut short is refused by every reader."""
    import subprocess
    import hostfiles as hf
    from mptrac_amd import build
    from mptrac_amd.synth import synthetic_met
    import cases
    build.build_host()
    m = synthetic_met("tiny", 0.0, 1.0, fields=cases.PRESSURE_LEVEL_FIELDS)
    m.f3["h2o"][::3, ::2, ::5] = -1e-6          # the reader's lower limit
    m.f3["lwc"][1::4, :, 2] = -0.5
    src = str(tmp_path / "met_2000_01_01_00.bin")
    hf.write_met_bin(src, m)
    outs = []
    for threads in (1, 3, 8):
        dst = str(tmp_path / ("out%d_2000_01_01_00.bin" % threads))
        r = subprocess.run([build.MET_CONV_BIN, "-", src, "1", dst, "1"], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, MPTRAC_AMD_IO_THREADS=str(threads)))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
        outs.append(open(dst, "rb").read())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) == os.path.getsize(src)
    assert outs[0] != open(src, "rb").read()                 # (the negative values were cut)
    cut = str(tmp_path / "cut_2000_01_01_00.bin")
    open(cut, "wb").write(open(src, "rb").read()[:-(m.nx * m.ny * m.np * 4 * 2 + 100)])
    for threads in (1, 8):
        r = subprocess.run([build.MET_CONV_BIN, "-", cut, "1", str(tmp_path / "x.bin"), "1"], capture_output=True, text=True,
                           timeout=120, env=dict(os.environ, MPTRAC_AMD_IO_THREADS=str(threads)))
        assert r.returncode != 0 and "Error while reading" in r.stdout + r.stderr, (threads, r.stdout[-500:])
