"""The C host layer (mptrac_amd/host: the reference's mptrac_* interface and
the `trac` driver) end to end: files in the reference's formats in, particle
files out, compared with the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import cases
import hostfiles as hf
from mptrac_amd import build
from mptrac_amd.clim import load_clim_tropo
from mptrac_amd.ctl import ctl_from_quantities
from mptrac_amd.synth import FIELDS_METEO_ONLY, synthetic_met, synthetic_particles
from oracle import binding as B

T0 = 707443200.0      # 2022-06-02 00:00 UTC, as tests/dd_test of the reference
QUANT = ("m", "rp", "rhop")
# with module_meteo outputs (default MET_DT_OUT 0.1: every step), as the reference's tests/trac_test sets them
QUANT_METEO = ("m", "rp", "rhop", "t", "u", "zg", "pv", "ps", "pt", "theta", "rh", "sst")


def _setup(tmp, n=3000, hours=2, atm_type=1, pbl=False, meteo=False, extra=None):
    lib, trac = build.build_host()
    metbase = os.path.join(tmp, "met")
    mets = []
    fields = cases.PRESSURE_LEVEL_FIELDS + FIELDS_METEO_ONLY if meteo else None
    QUANT = QUANT_METEO if meteo else globals()["QUANT"]
    for k in range(hours + 1):
        m = synthetic_met("tiny", T0 + 3600.0 * k, 1.0 + 0.1 * k, fields=fields)
        hf.write_met_bin(hf.met_filename(metbase, m.time), m)
        mets.append(m)
    atm = synthetic_particles(n, time=T0, quantities=QUANT)
    if pbl:      # half of the particles inside the boundary layer
        atm["p"][::2] = 1013.25 * np.exp(-(0.02 + 0.9 * (atm["lon"][::2] + 180.0) / 360.0) / 7.0)
    (hf.write_atm_bin if atm_type == 1 else hf.write_atm_asc)(os.path.join(tmp, "atm_in"), atm)
    keys = {"NQ": len(QUANT), "METBASE": metbase, "MET_TYPE": 1, "DT_MET": 3600, "DT_MOD": 180, "ADVECT": 4, "DIFFUSION": 1, "TURB_DZ_TROP": 0.1,
            "CONV_CAPE": 0, "T_STOP": T0 + 3600.0 * hours, "ATM_TYPE": atm_type, "ATM_TYPE_OUT": 1,
            "ATM_BASENAME": "atm", "ATM_DT_OUT": 3600, "GRID_BASENAME": "grid", "GRID_DT_OUT": 3600,
            "GRID_NX": 36, "GRID_NY": 18}
    keys.update({"QNT_NAME[%d]" % i: q for i, q in enumerate(QUANT)})
    if not meteo:
        keys["MET_DT_OUT"] = 0
    if pbl:
        keys.update({"TURB_PBL_SCHEME": 1, "TURB_MESOZ": 0})
    keys.update(extra or {})
    hf.write_ctl(os.path.join(tmp, "trac.ctl"), keys)
    open(os.path.join(tmp, "dirlist"), "w").write(tmp + "\n")
    return trac, mets, atm


def _oracle(mets, atm, hours, pbl=False, meteo=False):
    ctl = dict(advect=4, dt_mod=180.0, dt_met=3600.0, diffusion=1, turb_dz_trop=0.1, conv_cape=0.0,
               t_stop=T0 + 3600.0 * hours, met_dt_out=0.1 if meteo else 0.0,
               **ctl_from_quantities(QUANT_METEO if meteo else QUANT))
    if pbl:
        ctl.update(turb_pbl_scheme=1, turb_mesoz=0.0)
    o = B.Oracle(ctl, load_clim_tropo(), mets[0], mets[1], atm)
    o.timesteps_init()
    imet = 0
    snaps = {}
    for t in cases.step_times(o.ctl):
        while t > o.met[1].time:
            imet += 1
            o.swap_met(mets[imet + 1])
        o.run_timestep(t)
        if (t - T0) % 3600.0 == 0:
            snaps[t] = o.state()
    return snaps


def test_trac_refuses_to_run_without_a_device(tmp_path):
    """No CPU fallback: on a box without a GPU the driver stops with an error."""
    from test_abi import _have_gpu
    if _have_gpu():
        pytest.skip("a HIP device is present")
    trac, mets, atm = _setup(str(tmp_path), n=10, hours=1)
    r = subprocess.run([trac, os.path.join(str(tmp_path), "dirlist"), "trac.ctl", "atm_in"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode != 0 and "HIP device" in out, out[-2000:]


@pytest.mark.parametrize("args,msg", [(("DEPO_BASENAME", "depo"), "DEPO_BASENAME is not implemented"),
                                      (("TRACER_CHEM", "1"), "TRACER_CHEM is not implemented"),
                                      (("GRID_GPFILE", "plot.gp"), "GRID_GPFILE is not implemented"),
                                      (("GRID_NC_QUANT[1]", "3"), "quantisation of netCDF output"),
                                      (("GRID_TYPE", "2"), "Set GRID_TYPE to 0 or 1"),
                                      (("ADVECT_VERT_COORD", "2"), "requires meteo data on model levels"),
                                      (("RNG_TYPE", "0"), "RNG_TYPE 1")])
def test_trac_rejects_what_it_does_not_implement(tmp_path, args, msg):
    """Control keys of the reference this host layer has no code for stop the run with a message instead of
    being ignored (the reference ignores unknown keys, so a silent drop would look like a normal run)."""
    trac, mets, atm = _setup(str(tmp_path), n=10, hours=1)
    r = subprocess.run([trac, os.path.join(str(tmp_path), "dirlist"), "trac.ctl", "atm_in", *args],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode != 0 and msg in out, out[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("atm_type,pbl,meteo", [(1, False, False), (0, False, False), (1, True, False), (1, False, True)])
def test_trac_end_to_end_matches_oracle(tmp_path, atm_type, pbl, meteo):
    tmp = str(tmp_path)
    nq = len(QUANT_METEO if meteo else QUANT)
    trac, mets, atm = _setup(tmp, n=3000, hours=2, atm_type=atm_type, pbl=pbl, meteo=meteo)
    r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    if atm_type == 0:
        # ASCII input carries altitude; the driver converts with P(z) like the reference
        atm = dict(atm, p=1013.25 * np.exp(-(7.0 * np.log(1013.25 / atm["p"])) / 7.0))
    snaps = _oracle(mets, atm, 2, pbl=pbl, meteo=meteo)
    for hour in (0, 1, 2):
        t = T0 + 3600.0 * hour
        f = os.path.join(tmp, "atm_2022_06_02_%02d_00_00.bin" % hour)
        got = hf.read_atm_bin(f, nq)
        ref = snaps[t]
        tol = 1e-10 if atm_type == 1 else 1e-9     # ASCII input: repr() round trip of z -> p
        assert np.array_equal(got["time"], ref["time"])
        for k in ("lon", "lat", "p"):
            assert cases.rel_err(got[k], ref[k]) <= tol, (hour, k, cases.rel_err(got[k], ref[k]))
        assert cases.rel_err(got["q"], ref["q"]) <= tol
        g = os.path.join(tmp, "grid_2022_06_02_%02d_00_00.tab" % hour)
        rows = [ln.split() for ln in open(g) if ln.strip() and not ln.startswith("#")]
        assert len(rows) == 36 * 18 and sum(int(r[8]) for r in rows) == 3000


@pytest.mark.gpu
def test_trac_step_queue_is_not_observable(tmp_path):
    """The host layer holds back the time steps mptrac_run_timestep is given while they follow each other and hands
    them to the device as one mphip_run_timesteps call when anything else needs it (output, a new meteo file, the
    end of the run): the driver's loop stays the reference's, one call per step.  Every output file is the same,
    byte for byte, with the queue (default), with short queues and without it (HIP_STEP_BATCH 1)."""
    import hashlib
    digests = {}
    for batch in ("default", "4", "1"):
        tmp = str(tmp_path / ("batch_" + batch))
        os.makedirs(tmp)
        trac, mets, atm = _setup(tmp, n=3000, hours=2, extra={"ATM_DT_OUT": 1800})
        env = dict(os.environ)
        if batch != "default":
            env["HIP_STEP_BATCH"] = batch
        r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, env=env)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        files = sorted(f for f in os.listdir(tmp) if f.startswith(("atm_2022", "grid_2022")))
        assert len(files) == 5 + 3, files      # particles every half hour, the grid every hour
        digests[batch] = [(f, hashlib.sha1(open(os.path.join(tmp, f), "rb").read()).hexdigest()) for f in files]
    assert digests["4"] == digests["default"] and digests["1"] == digests["default"]


@pytest.mark.gpu
def test_trac_with_meteo_read_ahead(tmp_path):
    """HIP_MET_PREFETCH 1: the next meteo file is read by a thread and uploaded on the copy stream while
    the interval's time steps run; results as without it (3 h, 4 files, two hand-overs from the read-ahead)."""
    tmp = str(tmp_path)
    trac, mets, atm = _setup(tmp, n=3000, hours=3, extra={"HIP_MET_PREFETCH": 1})
    r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert out.count("Meteo data from the read-ahead") == 2, out[-3000:]
    snaps = _oracle(mets, atm, 3)
    for hour in (1, 2, 3):
        got = hf.read_atm_bin(os.path.join(tmp, "atm_2022_06_02_%02d_00_00.bin" % hour), len(QUANT))
        ref = snaps[T0 + 3600.0 * hour]
        assert np.array_equal(got["time"], ref["time"])
        for k in ("lon", "lat", "p"):
            assert cases.rel_err(got[k], ref[k]) <= 1e-10, (hour, k, cases.rel_err(got[k], ref[k]))
        assert cases.rel_err(got["q"], ref["q"]) <= 1e-10


@pytest.mark.gpu
def test_trac_writes_the_reference_coord_test_files_byte_for_byte(tmp_path):
    """The reference's tests/coord_test through the drop-in `trac` driver on the GPU: the control file and
    command line of the reference's run.sh (its netCDF meteo files handed over as MET_TYPE 1 binaries, the
    format this host layer reads), its input particle file, and the thirteen golden output files compared as
    text -- identical bytes."""
    import shutil
    import ref_coord as R
    tmp = str(tmp_path)
    _, trac = build.build_host()
    metbase = os.path.join(tmp, "era5_utm32")
    for h in range(3):
        m = R.load_met(h)
        hf.write_met_bin(hf.met_filename(metbase, m.time), m)
    # the particle file the reference's tools wrote carries the same text as its first output
    shutil.copy(os.path.join(R.HERE, R.OUTPUTS[0]), os.path.join(tmp, "atm_split.tab"))
    keys = {"NQ": 4, "QNT_NAME[0]": "t", "QNT_NAME[1]": "u", "QNT_NAME[2]": "v", "QNT_NAME[3]": "w",
            "METBASE": metbase, "MET_TYPE": 1, "TRACER_CHEM": 0, "DIFFUSION": 1, "DT_MET": 3600.0,
            "T_STOP": R.T0 + 7200.0}
    hf.write_ctl(os.path.join(tmp, "trac.ctl"), keys)
    open(os.path.join(tmp, "dirlist"), "w").write(tmp + "\n")
    r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_split.tab", "ATM_BASENAME", "atm",
                        "MET_CAPE", "0", "DT_MOD", "600", "ATM_DT_OUT", "600", "MET_COORD_TYPE", "1",
                        "MET_UTM_REF_LON", "11.5692782", "MET_UTM_REF_LAT", "48.1507476"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    for name in R.OUTPUTS.values():
        got = open(os.path.join(tmp, name)).read()
        ref = open(os.path.join(R.HERE, name)).read()
        if got != ref:
            gl, rl = got.splitlines(), ref.splitlines()
            bad = [i for i in range(min(len(gl), len(rl))) if gl[i] != rl[i]]
            raise AssertionError("%s: %d of %d lines differ, first: %r vs %r" % (
                name, len(bad) + abs(len(gl) - len(rl)), len(rl), gl[bad[0]] if bad else None, rl[bad[0]] if bad else None))


def _as_netcdf4(src, dst, style="old"):
    """the variables of a classic netCDF file in an HDF5 file of the old on-disk style (tests/h5write.py): the
    dimensions as dimension scales, everything with more than one axis chunked, shuffled and deflated"""
    from scipy.io import netcdf_file
    import h5write
    f = netcdf_file(src, "r", mmap=False)
    w = h5write.Writer(style)
    for name, var in f.variables.items():
        if not var.shape:      # (a scalar variable: nothing the reader of the meteo files needs)
            continue
        data = np.array(var[:])
        attrs = [(k, np.asarray(v)) for k, v in var._attributes.items() if not isinstance(v, (bytes, str))]
        if data.ndim <= 1 and name in f.dimensions:
            h5write.dimension_scale(w, name, data.astype(data.dtype.newbyteorder("<")))
        elif data.ndim <= 1:
            w.dataset(name, data.astype(data.dtype.newbyteorder("<")), attrs=attrs)
        else:
            chunks = tuple(max(1, (n + 1) // 2) for n in data.shape)
            w.dataset(name, data.astype(data.dtype.newbyteorder("<")), chunks=chunks, shuffle=True, deflate=3, attrs=attrs)
    f.close()
    w.close(dst)


@pytest.mark.gpu
@pytest.mark.parametrize("container", ["classic", "netcdf4", "netcdf4_new_style"])
def test_trac_runs_the_reference_coord_test_command_line_on_its_netcdf_files(tmp_path, container):
    """tests/coord_test/run.sh of the reference as it stands: control file, command line (MET_TYPE left at its
    default 0 = netCDF) and the three classic-netCDF meteo files -- read by the host layer's own reader, no
    conversion step -- give the thirteen golden particle files byte for byte.  `netcdf4`: the same variables
    repacked into HDF5 files (chunked, shuffled, deflated; old and new on-disk style, tests/h5write.py) and read by
    the host layer's HDF5 reader: the same bytes come out."""
    import shutil
    import ref_coord as R
    tmp = str(tmp_path)
    _, trac = build.build_host()
    shutil.copy(os.path.join(R.HERE, R.OUTPUTS[0]), os.path.join(tmp, "atm_split.tab"))
    metbase = os.path.join(R.HERE, "era5_utm32")
    if container != "classic":
        metbase = os.path.join(tmp, "era5_utm32")
        for h in range(3):
            name = "era5_utm32_2025_05_01_%02d.nc" % h
            _as_netcdf4(os.path.join(R.HERE, name), os.path.join(tmp, name), "new" if container.endswith("new_style") else "old")
            assert open(os.path.join(tmp, name), "rb").read(4) == b"\x89HDF"
    keys = {"NQ": 4, "QNT_NAME[0]": "t", "QNT_NAME[1]": "u", "QNT_NAME[2]": "v", "QNT_NAME[3]": "w",
            "METBASE": metbase, "TRACER_CHEM": 0, "DIFFUSION": 1, "DT_MET": 3600.0,
            "T_STOP": R.T0 + 7200.0}
    hf.write_ctl(os.path.join(tmp, "trac.ctl"), keys)
    open(os.path.join(tmp, "dirlist"), "w").write(tmp + "\n")
    r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_split.tab", "ATM_BASENAME", "atm",
                        "MET_CAPE", "0", "DT_MOD", "600", "ATM_DT_OUT", "600", "MET_COORD_TYPE", "1",
                        "MET_UTM_REF_LON", "11.5692782", "MET_UTM_REF_LAT", "48.1507476"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert "era5_utm32_2025_05_01_01.nc" in out
    for name in R.OUTPUTS.values():
        got = open(os.path.join(tmp, name)).read()
        ref = open(os.path.join(R.HERE, name)).read()
        if got != ref:
            gl, rl = got.splitlines(), ref.splitlines()
            bad = [i for i in range(min(len(gl), len(rl))) if gl[i] != rl[i]]
            raise AssertionError("%s: %d of %d lines differ, first: %r vs %r" % (
                name, len(bad) + abs(len(gl) - len(rl)), len(rl), gl[bad[0]] if bad else None, rl[bad[0]] if bad else None))


@pytest.mark.gpu
def test_trac_runs_the_reference_dd_test_on_global_netcdf_meteo_files(tmp_path):
    """tests/dd_test of the reference through the drop-in driver with MET_TYPE 0 files on a global longitude /
    latitude grid: the wind tool's solid-body rotation written as netCDF (tests/c/wind_met.c -> mptrac_write_met),
    read back by the host layer's reader -- latitudes from north to south, polar-wind fix, periodic longitude column
    (read_met_polar_winds, read_met_periodic) --, six hours of midpoint advection of the 144 golden parcels: the seven
    hourly particle files carry the golden rows (time, altitude, longitude, latitude, idx, zeta, m), every printed digit."""
    tmp = str(tmp_path)
    _, trac = build.build_host()
    exe = hf.compile_c_test("wind_met")
    metbase = os.path.join(tmp, "wind")
    r = subprocess.run([exe, metbase, repr(T0), "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RESULT done" in r.stdout, r.stdout[-2000:]
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_dd_test")
    gold = {h: [ln.strip() for ln in open(os.path.join(gold_dir, "atm_2022_06_02_%02d_00_00.tab" % h)) if ln.strip()]
            for h in range(7)}
    open(os.path.join(tmp, "init.tab"), "w").write("\n".join(gold[0]) + "\n")
    # tests/dd_test/data.ref/config.ctl (the two domain-decomposition quantities under neutral names)
    keys = {"NQ": 5, "QNT_NAME[0]": "idx", "QNT_UNIT[0]": "-", "QNT_NAME[1]": "zeta", "QNT_NAME[2]": "m",
            "QNT_NAME[3]": "sub_a", "QNT_UNIT[3]": "-", "QNT_NAME[4]": "sub_b", "QNT_UNIT[4]": "-",
            "METBASE": metbase, "MET_TYPE": 0, "MET_DT_OUT": 0, "ADVECT": 2, "ADVECT_VERT_COORD": 0,
            "TURB_DX_TROP": 0, "TURB_DX_STRAT": 0, "TURB_DZ_TROP": 0, "TURB_DZ_STRAT": 0.0, "TURB_MESOX": 0.0,
            "TURB_MESOZ": 0.0, "DIRECTION": 1, "TDEC_TROP": 259200, "TDEC_STRAT": 259200, "DT_MOD": 600, "DT_MET": 3600,
            "T_START": repr(T0), "T_STOP": repr(T0 + 6 * 3600.0), "ATM_DT_OUT": 3600, "ATM_BASENAME": "atm"}
    hf.write_ctl(os.path.join(tmp, "trac.ctl"), keys)
    open(os.path.join(tmp, "dirlist"), "w").write(tmp + "\n")
    r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "init.tab"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    for h in range(7):
        rows = [ln.strip() for ln in open(os.path.join(tmp, "atm_2022_06_02_%02d_00_00.tab" % h))
                if ln.strip() and not ln.startswith("#")]
        # (the last two columns are the subdomain bookkeeping of the reference's domain decomposition, which moves
        # parcels between ranks: not part of this build)
        bad = [i for i in range(144) if rows[i].split()[:7] != gold[h][i].split()[:7]]
        assert len(rows) == 144 and not bad, (h, len(bad), rows[bad[0]], gold[h][bad[0]])


def _atm_test_run(tmp, extra_args=(), case="ref_atm_test", atm_file="atm_2000_01_01_00_00_00.tab",
                  quantities=("aoa", "m", "vmr"), t0=0.0):
    """`trac` on a golden particle file of the reference (default: tests/atm_test, 10000 parcels with aoa, m, vmr); the
    outputs at t0 are written after the first call of the time step, which moves nothing (dt = 0)."""
    import shutil
    _, trac = build.build_host()
    metbase = os.path.join(tmp, "met")
    for k in range(2):
        m = synthetic_met("tiny", t0 + 3600.0 * k, 1.0 + 0.1 * k)
        hf.write_met_bin(hf.met_filename(metbase, m.time), m)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", case)
    shutil.copy(os.path.join(gold, atm_file), os.path.join(tmp, "atm_in.tab"))
    keys = {"NQ": len(quantities), "METBASE": metbase, "MET_TYPE": 1,
            "DT_MET": 3600, "DT_MOD": 180, "T_STOP": repr(t0 + 180), "MET_DT_OUT": 0, "ATM_BASENAME": "atm",
            "ATM_DT_OUT": 86400, "GRID_BASENAME": "grid", "GRID_DT_OUT": 86400, "GRID_NX": 72, "GRID_NY": 36}
    for i, q in enumerate(quantities):
        keys[f"QNT_NAME[{i}]"] = q
    hf.write_ctl(os.path.join(tmp, "trac.ctl"), keys)
    open(os.path.join(tmp, "dirlist"), "w").write(tmp + "\n")
    r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in.tab", *extra_args],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    return gold


def _same_text(got_path, ref_path):
    got, ref = open(got_path).read(), open(ref_path).read()
    if got != ref:
        gl, rl = got.splitlines(), ref.splitlines()
        bad = [i for i in range(min(len(gl), len(rl))) if gl[i] != rl[i]]
        raise AssertionError("%s: %d lines differ, first: %r vs %r" % (os.path.basename(ref_path), len(bad),
                                                                       gl[bad[0]], rl[bad[0]]))


@pytest.mark.gpu
def test_trac_grid_and_particle_files_equal_the_reference_atm_test_goldens(tmp_path):
    """Gridded output (binning on the device, post-processing and text layout on the host) and the ASCII
    particle reader / writer against the reference's tests/atm_test: the `atm2grid` golden
    (72 x 36 cells, column density, particle counts, means of aoa / m / vmr) and the particle file itself,
    byte for byte."""
    tmp = str(tmp_path)
    gold = _atm_test_run(tmp)
    for name in ("grid_2000_01_01_00_00_00.tab", "atm_2000_01_01_00_00_00.tab"):
        _same_text(os.path.join(tmp, name), os.path.join(gold, name))


@pytest.mark.gpu
def test_trac_grid_file_equals_the_reference_dt_test_golden(tmp_path):
    """The `atm2grid` golden of the reference's tests/dt_test (run.sh:47-49): 10000 parcels carrying t, u, v, w and no
    mass, GRID_NX 72, GRID_NY 36, MOLMASS set -- column density and mixing ratio are printed as nan, the one
    occupied cell holds the means; gridded file and particle file byte for byte."""
    tmp = str(tmp_path)
    t0 = 360547200.0      # 2011-06-05 00:00 UTC
    gold = _atm_test_run(tmp, ("MOLMASS", "64.066"), case="ref_dt_test", atm_file="atm_pl_2011_06_05_00_00_00.tab",
                         quantities=("t", "u", "v", "w"), t0=t0)
    _same_text(os.path.join(tmp, "grid_2011_06_05_00_00_00.tab"), os.path.join(gold, "grid_2011_06_05_00_00_00.tab"))
    _same_text(os.path.join(tmp, "atm_2011_06_05_00_00_00.tab"), os.path.join(gold, "atm_pl_2011_06_05_00_00_00.tab"))


@pytest.mark.gpu
def test_trac_sparse_box_grid_against_the_reference_trac_test_golden(tmp_path):
    """The gridded output of the reference's tests/trac_test at 2011-06-07 (300 x 90 cells on -90...60 E, -60...-15 N,
    GRID_SPARSE 1, thirteen quantities) from the golden particle file of that time.  The particle file holds six
    digits, so a parcel within 1e-4 degrees of a cell border may change cells and the sums carry the rounding of
    their summands: the rows are compared cell by cell -- nearly all cells with the same count, and in those the
    column density and every mean within 2e-5 (the implicit mixing ratio needs that run's temperatures: skipped;
    a sparse table lists the cells in which it is positive, i.e. the cells that hold mass)."""
    tmp = str(tmp_path)
    quantities = ("t", "u", "v", "w", "zg", "pv", "ps", "pt", "m", "stat", "ens", "qa", "qb")
    extra = ["GRID_LON0", "-90", "GRID_LON1", "60", "GRID_LAT0", "-60", "GRID_LAT1", "-15", "GRID_NX", "300",
             "GRID_NY", "90", "GRID_SPARSE", "1", "SPECIES", "SO2", "OH_CHEM_REACTION", "0", "DT_MOD", "300", "T_STOP", repr(360720000.0 + 300.0)]
    for i, q in enumerate(quantities):
        if q in ("qa", "qb"):
            extra += [f"QNT_UNIT[{i}]", "ppv"]
    gold = _atm_test_run(tmp, extra, case="ref_trac_test", atm_file="atm_pl_2011_06_07_00_00_00.tab",
                         quantities=quantities, t0=360720000.0)

    def rows(path):
        out = {}
        for ln in open(path):
            if ln.strip() and not ln.startswith("#"):
                c = [float(x) for x in ln.split()]
                out[(c[2], c[3])] = c
        return out
    mine = rows(os.path.join(tmp, "grid_2011_06_07_00_00_00.tab"))
    ref = rows(os.path.join(gold, "grid_pl_2011_06_07_00_00_00.tab"))
    assert sum(c[8] for c in mine.values()) == sum(c[8] for c in ref.values())       # every parcel with mass inside the box
    same = [k for k in ref if k in mine and mine[k][8] == ref[k][8]]
    assert len(same) >= 0.995 * len(ref) and len(mine) <= 1.005 * len(ref), (len(same), len(ref), len(mine))
    cols = [0, 1, 4, 5, 6] + list(range(9, 9 + len(quantities)))
    a = np.array([[mine[k][c] for c in cols] for k in same])
    b = np.array([[ref[k][c] for c in cols] for k in same])
    bad = ~np.isclose(a, b, rtol=2e-5, atol=0)
    assert bad.sum() <= 0.002 * bad.size, (bad.sum(), bad.size, a[bad][:5], b[bad][:5])


@pytest.mark.gpu
def test_trac_nat_and_sts_temperatures_from_the_hno3_climatology(tmp_path):
    """module_meteo's climatology quantities through the driver: CLIM_HNO3_FILENAME = the reference's HNO3 file,
    quantities h2o, tnat, tice, tsts on the 10000 parcels of the reference's dt_test particle file.  The particle
    file written at the start time holds T_NAT = nat_temperature(p, h2o, HNO3(t, lat, p)) -- recomputed here from
    the printed columns with the independent reading of the table (tests/refclim.py) -- and T_STS = their mean."""
    import refclim
    from oracle import binding as B
    tmp = str(tmp_path)
    t0 = 360547200.0
    _atm_test_run(tmp, ("MET_DT_OUT", "0.1", "CLIM_HNO3_FILENAME", refclim.HNO3_FILE), case="ref_dt_test",
                  atm_file="atm_pl_2011_06_05_00_00_00.tab", quantities=("h2o", "tnat", "tice", "tsts"), t0=t0)
    rows = np.loadtxt(os.path.join(tmp, "atm_2011_06_05_00_00_00.tab"))
    assert rows.shape == (10000, 8)
    L = B.lib()
    table = refclim.load_zonal_mean()
    press = 1013.25 * np.exp(-rows[:, 1] / 7.0)
    h2o, tnat, tice, tsts = rows[:, 4], rows[:, 5], rows[:, 6], rows[:, 7]
    assert h2o.min() > 0 and 150 < tnat.min() and tnat.max() < 260
    assert np.allclose(tsts, 0.5 * (tice + tnat), rtol=0, atol=1.1e-3)       # six printed digits each
    for ip in range(0, 10000, 97):
        hno3 = refclim.clim_zm(table, rows[ip, 0], rows[ip, 3], press[ip])
        assert abs(L.orc_nat_temperature(press[ip], h2o[ip], hno3) - tnat[ip]) <= 2e-4 * tnat[ip]
        assert abs(L.orc_tice(press[ip], h2o[ip]) - tice[ip]) <= 2e-4 * tice[ip]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["atm_test", "dt_test"])
def test_atm2grid_commands_of_the_reference_tests(tmp_path, case):
    """The `atm2grid` tool of this build (host/atm2grid.c: binning on the device, write_grid on the host) with the
    command lines of the reference's tests/atm_test/run.sh:86-87 and tests/dt_test/run.sh:47-48 on their golden
    particle files: the golden gridded files, byte for byte."""
    import shutil
    from mptrac_amd.build import ATM2GRID_BIN
    build.build_host()
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if case == "atm_test":
        atm, grid, stamp = os.path.join(gold, "ref_atm_test", "atm_2000_01_01_00_00_00.tab"), "grid_2000_01_01_00_00_00.tab", "ref_atm_test"
        args = ["-", None, "GRID_BASENAME", None, "GRID_NX", "72", "GRID_NY", "36", "NQ", "3", "QNT_NAME[0]", "aoa",
                "QNT_NAME[1]", "m", "QNT_NAME[2]", "vmr", "MOLMASS", "64.066"]
    else:
        atm, grid, stamp = os.path.join(gold, "ref_dt_test", "atm_pl_2011_06_05_00_00_00.tab"), "grid_2011_06_05_00_00_00.tab", "ref_dt_test"
        hf.write_ctl(str(tmp_path / "trac.ctl"), {"NQ": 4, "QNT_NAME[0]": "t", "QNT_NAME[1]": "u", "QNT_NAME[2]": "v",
                                                  "QNT_NAME[3]": "w", "DT_MOD": 10.0, "DIFFUSION": 1})
        args = [str(tmp_path / "trac.ctl"), None, "GRID_BASENAME", None, "GRID_NX", "72", "GRID_NY", "36", "MOLMASS", "64.066"]
    local = str(tmp_path / os.path.basename(atm))
    shutil.copy(atm, local)
    args[1], args[3] = local, str(tmp_path / "grid")
    r = subprocess.run([ATM2GRID_BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    _same_text(str(tmp_path / grid), os.path.join(gold, stamp, grid))


@pytest.mark.gpu
def test_trac_grid_implicit_volume_mixing_ratio(tmp_path):
    """MOLMASS set: column 8 of the grid file = MA / MOLMASS * column density / (rho(p, T) * dz), T interpolated
    to the cell centre from the two snapshots (mptrac.c:13885-13900)."""
    tmp = str(tmp_path)
    _atm_test_run(tmp, ("MOLMASS", "64.066"))
    rows = np.array([[float(c) for c in ln.split()] for ln in open(os.path.join(tmp, "grid_2000_01_01_00_00_00.tab"))
                     if ln.strip() and not ln.startswith("#")])
    m0 = synthetic_met("tiny", 0.0, 1.0)
    lon, lat, cd, vmr, cnt = rows[:, 2], rows[:, 3], rows[:, 6], rows[:, 7], rows[:, 8]
    press = 1013.25 * np.exp(-40.0 / 7.0)
    ip = np.searchsorted(-m0.p, -press) - 1
    wp = (m0.p[ip + 1] - press) / (m0.p[ip + 1] - m0.p[ip])
    temp = np.empty(len(rows))
    for i in range(len(rows)):
        ix = min(int((lon[i] - m0.lon[0]) / (m0.lon[1] - m0.lon[0])), m0.nx - 2)
        iy = min(max(np.searchsorted(m0.lat, lat[i]) - 1, 0), m0.ny - 2)
        wx = (m0.lon[ix + 1] - lon[i]) / (m0.lon[ix + 1] - m0.lon[ix])
        wy = (m0.lat[iy + 1] - lat[i]) / (m0.lat[iy + 1] - m0.lat[iy])
        T = m0.f3["t"].astype(np.float64)
        col = lambda a, b: wp * (T[a, b, ip] - T[a, b, ip + 1]) + T[a, b, ip + 1]
        a0 = wy * (col(ix, iy) - col(ix, iy + 1)) + col(ix, iy + 1)
        a1 = wy * (col(ix + 1, iy) - col(ix + 1, iy + 1)) + col(ix + 1, iy + 1)
        temp[i] = wx * (a0 - a1) + a1          # t = met0.time: the time weight is 1
    expect = np.where(cnt > 0, 28.9644 / 64.066 * cd / (100.0 * press / (1e3 * 8.3144598 / 28.9644 * temp) * 90.0 * 1e3), 0.0)
    assert cnt.sum() == 10000 and (vmr[cnt == 0] == 0).all()
    assert np.allclose(vmr, expect, rtol=2e-5, atol=0)       # six printed digits of cd and vmr


@pytest.mark.gpu
def test_trac_grid_with_a_vertical_weighting_function(tmp_path):
    """GRID_KERNEL through the driver: the kernel file is read at the first step (heights ascending, weights scaled
    to a largest weight of one), every particle enters the means with weight(z): the grid file of the reference's
    atm_test particles equals the weighted means computed here from the same particle file."""
    tmp = str(tmp_path)
    with open(os.path.join(tmp, "kernel.tab"), "w") as f:
        f.write("# z [km]  weight\n0 0.4\n5 2.0\n12 1.0\n30 0.2\n")
    gold = _atm_test_run(tmp, ("GRID_KERNEL", os.path.join(tmp, "kernel.tab")))
    rows = np.array([[float(c) for c in ln.split()] for ln in open(os.path.join(tmp, "grid_2000_01_01_00_00_00.tab"))
                     if ln.strip() and not ln.startswith("#")])
    atm = np.array([[float(c) for c in ln.split()] for ln in open(os.path.join(gold, "atm_2000_01_01_00_00_00.tab"))
                    if ln.strip() and not ln.startswith("#")])
    z, lon, lat, q = atm[:, 1], atm[:, 2], atm[:, 3], atm[:, 4:7]
    w = np.interp(z, [0.0, 5.0, 12.0, 30.0], [0.2, 1.0, 0.5, 0.1])       # scaled by the largest weight; constant beyond the ends
    ix, iy = np.floor((lon + 180.0) / 5.0).astype(int), np.floor((lat + 90.0) / 5.0).astype(int)
    inside = (lon >= -180) & (lon < 180) & (lat >= -90) & (lat < 90) & (z >= -5) & (z < 85)
    cnt = np.zeros((72, 36))
    mean = np.zeros((3, 72, 36))
    np.add.at(cnt, (ix[inside], iy[inside]), 1)
    for k in range(3):
        np.add.at(mean[k], (ix[inside], iy[inside]), w[inside] * q[inside, k])
    got_cnt = rows[:, 8].reshape(72, 36)
    assert np.array_equal(got_cnt, cnt) and cnt.sum() == 10000
    for k in range(3):
        got = rows[:, 9 + k].reshape(72, 36)
        want = np.where(cnt > 0, mean[k] / np.maximum(cnt, 1), np.nan)
        ok = (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= 2e-5 * np.abs(want) + 1e-30)      # six printed digits in and out
        assert ok.all(), k
    plain = np.array([[float(c) for c in ln.split()] for ln in open(os.path.join(gold, "grid_2000_01_01_00_00_00.tab"))
                      if ln.strip() and not ln.startswith("#")])
    assert not np.allclose(np.nan_to_num(plain[:, 10]), np.nan_to_num(rows[:, 10]), rtol=1e-3)      # the weights matter


@pytest.mark.gpu
def test_trac_balloon_isosurface_and_boundary_conditions(tmp_path):
    """ISOSURF 4 (the driver reads the BALLOON file at the first step) and BOUND_* keys through `trac`."""
    tmp = str(tmp_path)
    trac, mets, atm = _setup(tmp, n=2000, hours=1)
    ts = [T0 - 600.0 + 600.0 * k for k in range(9)]
    ps = [300.0 - 7.5 * k + 0.4 * (k % 3) for k in range(9)]
    with open(os.path.join(tmp, "balloon.tab"), "w") as f:
        f.write("# time [s]  pressure [hPa]\n")
        for a, b in zip(ts, ps):
            f.write("%.2f %.10g\n" % (a, b))
    extra = {"ISOSURF": 4, "BALLOON": os.path.join(tmp, "balloon.tab"), "BOUND_LAT0": -50, "BOUND_LAT1": 50,
             "BOUND_P0": 1100, "BOUND_P1": 100, "BOUND_MASS": 4.0, "BOUND_MASS_TREND": 1e-9, "T_STOP": T0 + 3600.0}
    with open(os.path.join(tmp, "trac.ctl"), "a") as f:
        for k, v in extra.items():
            f.write(f"{k} = {v}\n")
    r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    ctl = dict(advect=4, dt_mod=180.0, dt_met=3600.0, diffusion=1, turb_dz_trop=0.1, conv_cape=0.0,
               t_stop=T0 + 3600.0, met_dt_out=0.0, isosurf=4, bound_lat0=-50.0, bound_lat1=50.0, bound_p0=1100.0,
               bound_p1=100.0, bound_mass=4.0, bound_mass_trend=1e-9, **ctl_from_quantities(QUANT))
    o = B.Oracle(ctl, load_clim_tropo(), mets[0], mets[1], atm)
    o.timesteps_init()
    o.set_balloon(ts, ps)
    for t in cases.step_times(o.ctl):
        o.run_timestep(t)
    got = hf.read_atm_bin(os.path.join(tmp, "atm_2022_06_02_01_00_00.bin"), 3)
    ref = o.state()
    for k in ("lon", "lat", "p"):
        assert cases.rel_err(got[k], ref[k]) <= 1e-10, (k, cases.rel_err(got[k], ref[k]))
    assert cases.rel_err(got["q"], ref["q"]) <= 1e-10
    assert np.all(got["p"] == got["p"][0]) and np.any(got["q"][0] > 3.9)


@pytest.mark.gpu
def test_trac_netcdf_and_analysis_outputs(tmp_path):
    """The outputs besides ASCII / binary particle and grid files, through the drop-in driver on the GPU: the same
    run once with text / binary files and once with netCDF files (GRID_TYPE 1, ATM_TYPE_OUT 2) plus every analysis
    writer.  The netCDF files (own classic-format writer, read here by scipy) hold exactly the values of the
    text / binary files; the station file lists every particle at most once -- its `stat` flag, set on the host
    copy, is handed back to the device after every step (mphip_update_quantity, with the particles in the internal
    locality order) -- and the final particle file carries as many flags as the station file has rows."""
    from scipy.io import netcdf_file
    quant = ("m", "rp", "rhop", "stat", "ens")
    runs = {}
    for name, extra in (("plain", {}),
                        ("nc", {"GRID_TYPE": 1, "ATM_TYPE_OUT": 2, "ENS_BASENAME": "ens", "ENS_DT_OUT": 3600,
                                "VTK_BASENAME": "cloud", "VTK_DT_OUT": 3600, "VTK_STRIDE": 50, "STAT_BASENAME": "station",
                                "STAT_LON": 10, "STAT_LAT": 20, "STAT_R": 1500, "CSI_BASENAME": "csi", "CSI_DT_OUT": 3600,
                                "CSI_OBSFILE": "obs.tab", "CSI_NX": 36, "CSI_NY": 18, "CSI_MODMIN": 1e-12,
                                "CSI_OBSMIN": 0.5, "SAMPLE_BASENAME": "sample", "SAMPLE_OBSFILE": "obs.tab",
                                "SAMPLE_DX": 800, "PROF_BASENAME": "prof", "PROF_OBSFILE": "obs.tab", "PROF_NX": 36,
                                "PROF_NY": 18, "PROF_NZ": 10, "MOLMASS": 64})):
        tmp = str(tmp_path / name)
        os.makedirs(tmp)
        keys = {"NQ": len(quant), "GRID_STDDEV": 1, "GRID_NZ": 3, "GRID_Z0": 0, "GRID_Z1": 30, "HIP_LOCALITY_SORT_INTERVAL": 3}
        keys.update({"QNT_NAME[%d]" % i: q for i, q in enumerate(quant)})
        keys.update(extra)
        trac, mets, atm = _setup(tmp, n=4000, hours=1, extra=keys)
        # particle file with the two extra quantities (flags zero, four ensemble members)
        atm = synthetic_particles(4000, time=T0, quantities=quant)
        atm["q"][3][:] = 0.0
        atm["q"][4][:] = np.arange(4000) % 4
        hf.write_atm_bin(os.path.join(tmp, "atm_in"), atm)
        with open(os.path.join(tmp, "obs.tab"), "w") as f:      # observations at the last step of the hour
            for lon in range(-175, 180, 10):
                for lat in range(-85, 90, 10):
                    f.write("%.2f 5 %d %d %g\n" % (T0 + 3600.0, lon, lat, float(lon > 0)))
        r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, cwd=tmp)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        runs[name] = tmp
    stamp = "2022_06_02_01_00_00"
    # particles: netCDF == binary
    want = hf.read_atm_bin(os.path.join(runs["plain"], "atm_%s.bin" % stamp), len(quant))
    f = netcdf_file(os.path.join(runs["nc"], "atm_%s.nc" % stamp), "r", mmap=False)
    assert np.array_equal(f.variables["time"][:], want["time"]) and np.array_equal(f.variables["press"][:], want["p"])
    assert np.array_equal(f.variables["lon"][:], want["lon"]) and np.array_equal(f.variables["lat"][:], want["lat"])
    for iq, q in enumerate(quant):
        if q != "stat":
            assert np.array_equal(f.variables[q][:], want["q"][iq]), q
    flags = np.array(f.variables["stat"][:])
    assert f.variables["m"].units == b"kg" and f.variables["ens"].long_name == b"ensemble index"
    f.close()
    # grid: netCDF [time][z][lat][lon] == the rows of the text table
    rows = np.loadtxt(os.path.join(runs["plain"], "grid_%s.tab" % stamp))
    f = netcdf_file(os.path.join(runs["nc"], "grid_%s.nc" % stamp), "r", mmap=False)
    nx, ny, nz = 36, 18, 3
    assert f.variables["cd"].shape == (1, nz, ny, nx) and f.variables["np"].shape == (1, nz, ny, nx)
    table = rows.reshape(nx, ny, nz, -1)          # the text table runs lon, lat, z
    def same(var, col, rel=0.0):
        a = np.asarray(f.variables[var][0], dtype=np.float64).transpose(2, 1, 0)      # -> [lon][lat][z]
        b = table[:, :, :, col]
        ok = (np.isnan(a) & np.isnan(b)) | (np.abs(a - b) <= rel * np.abs(b)) | (a == b)
        assert ok.all(), var
    assert np.array_equal(f.variables["z"][:], table[0, 0, :, 1]) and np.array_equal(f.variables["lon"][:], table[:, 0, 0, 2])
    assert np.allclose(f.variables["area"][:], table[0, :, 0, 4], rtol=1e-5)
    same("np", 8)
    same("cd", 6, 1e-5)                  # (a float variable against six printed digits)
    for iq, q in enumerate(quant):
        if q != "stat":
            same(q + "_mean", 9 + iq, 1e-5)
            same(q + "_stddev", 9 + len(quant) + iq, 1e-4)
    assert f.variables["m_mean"].long_name == b"mass (mean)"
    f.close()
    # station: every particle once; the flags of the final particle file are the listed particles
    rows = np.loadtxt(os.path.join(runs["nc"], "station.tab"), ndmin=2)
    assert len(rows) > 20 and int(flags.sum()) == len(rows)
    assert np.all(rows[:, 4 + quant.index("stat")] == 1)
    # the other writers produced their files: one ensemble row with all particles, a VTK cloud of every 50th
    ens = np.loadtxt(os.path.join(runs["nc"], "ens_%s.tab" % stamp), ndmin=2)
    assert ens.shape == (1, 4 + 2 * len(quant) + 1) and ens[0, -1] == 4000
    vtk = open(os.path.join(runs["nc"], "cloud_00002.vtk")).read()
    assert "POINTS 80 float" in vtk and "SCALARS rhop float 1" in vtk
    csi = np.loadtxt(os.path.join(runs["nc"], "csi.tab"), ndmin=2)
    assert csi.shape[0] == 1 and csi[0, 0] == T0 + 3600.0 and csi[0, 5] > 100
    sample = np.loadtxt(os.path.join(runs["nc"], "sample.tab"), ndmin=2)
    assert sample.shape == (36 * 18, 10) and sample[:, 6].sum() > 1000
    prof = np.loadtxt(os.path.join(runs["nc"], "prof.tab"), ndmin=2)
    assert prof.shape[1] == 11 and prof.shape[0] % 10 == 0 and prof.shape[0] >= 10 and np.all(prof[:, 5] > 150)
