"""Writers for the reference's on-disk formats used by the host-side driver
tests: raw binary meteo files (MET_TYPE 1, version 104; field order of the
reference's read_met_bin, src/mptrac.c:8887-9043), particle files (ASCII and
binary version 100, src/mptrac.c:8380-8475) and control files."""
import os
import struct
import time

import numpy as np

SURF_ORDER = ["ps", "ts", "zs", "us", "vs", "ess", "nss", "shf", "lsm", "sst", "pbl", "pt", "tt", "zt", "h2ot",
              "pct", "pcb", "cl", "plcl", "plfc", "pel", "cape", "cin", "o3c"]
LEVEL_ORDER = ["z", "t", "u", "v", "w", "pv", "h2o", "o3", "lwc", "rwc", "iwc", "swc", "cc"]


def met_filename(metbase, t):
    tm = time.gmtime(946684800 + int(t))      # seconds since 2000-01-01
    return "%s_%d_%02d_%02d_%02d.bin" % (metbase, tm.tm_year, tm.tm_mon, tm.tm_mday, tm.tm_hour)


def write_met_bin(path, met):
    with open(path, "wb") as f:
        f.write(struct.pack("<iid", 1, 104, met.time))
        f.write(struct.pack("<iii", met.nx, met.ny, met.np))
        f.write(met.lon.tobytes() + met.lat.tobytes() + met.p.tobytes())
        zero2 = np.zeros((met.nx, met.ny), dtype=np.float32)
        zero3 = np.zeros((met.nx, met.ny, met.np), dtype=np.float32)
        for k in SURF_ORDER:
            f.write(met.f2.get(k, zero2).tobytes())
        for k in LEVEL_ORDER:
            f.write(met.f3.get(k, zero3).tobytes())
        f.write(struct.pack("<i", 999))


def write_atm_asc(path, atm):
    z = 7.0 * np.log(1013.25 / atm["p"])
    with open(path, "w") as f:
        f.write("# $1 = time [s]\n# $2 = altitude [km]\n# $3 = longitude [deg]\n# $4 = latitude [deg]\n\n")
        for i in range(len(z)):
            row = [atm["time"][i], z[i], atm["lon"][i], atm["lat"][i]] + [q[i] for q in atm["q"]]
            f.write(" ".join(repr(float(v)) for v in row) + "\n")


def write_atm_bin(path, atm):
    n = len(atm["time"])
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", 100, n))
        for k in ("time", "p", "lon", "lat"):
            f.write(np.ascontiguousarray(atm[k], dtype=np.float64).tobytes())
        for q in atm["q"]:
            f.write(np.ascontiguousarray(q, dtype=np.float64).tobytes())
        f.write(struct.pack("<i", 999))


def read_atm_bin(path, nq):
    raw = open(path, "rb").read()
    version, n = struct.unpack_from("<ii", raw, 0)
    assert version == 100
    a = np.frombuffer(raw, dtype=np.float64, count=(4 + nq) * n, offset=8).reshape(4 + nq, n)
    assert struct.unpack_from("<i", raw, 8 + 8 * (4 + nq) * n)[0] == 999
    return {"time": a[0].copy(), "p": a[1].copy(), "lon": a[2].copy(), "lat": a[3].copy(), "q": a[4:].copy()}


def write_ctl(path, keys):
    with open(path, "w") as f:
        for k, v in keys.items():
            f.write(f"{k} = {v}\n")


def compile_c_test(name):
    """gcc tests/c/<name>.c against the host layer's header and library -> path of the program."""
    import subprocess
    from mptrac_amd import build
    lib, _ = build.build_host()
    here = os.path.dirname(os.path.abspath(__file__))
    out_dir = os.path.join(here, "c", "build")
    os.makedirs(out_dir, exist_ok=True)
    src, exe = os.path.join(here, "c", name + ".c"), os.path.join(out_dir, name)
    if os.environ.get("MPTRAC_TEST_SANITIZE"):
        # the test and the host layer's sources as one program under the address and undefined-behaviour sanitizers
        exe += "_san"
        host = [os.path.join(build.HOST_DIR, f) for f in ("mptrac.c", "ctlfile.c", "nc_classic.c", "nc_hdf5.c",
                                                           "rendezvous.c", "output.c")]
        defs = [f"-D{k}={v}" for k, v in build.HOST_DIMS.items()]
        defs.append('-DMPTRAC_AMD_DATA_DIR="%s"' % os.path.join(os.path.dirname(build.HOST_DIR), "data"))
        subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", "-Wall", "-fsanitize=address,undefined",
                               "-fno-omit-frame-pointer", "-mcmodel=medium", *defs, "-I", build.HOST_DIR,
                               "-I", os.path.join(os.path.dirname(here), "include"), "-o", exe, src, *host,
                               "-L" + build.LIBDIR, "-lmptrac_hip", "-Wl,-rpath," + build.LIBDIR,
                               "-Wl,-rpath,/opt/rocm/lib", "-lm", "-lpthread", "-lz"])
        return exe
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        defs = [f"-D{k}={v}" for k, v in build.HOST_DIMS.items()]
        subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-Wall", "-mcmodel=medium", *defs, "-I", build.HOST_DIR,
                               "-o", exe, src, "-L" + build.LIBDIR, "-lmptrac", "-lmptrac_hip",
                               "-Wl,-rpath," + build.LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return exe
