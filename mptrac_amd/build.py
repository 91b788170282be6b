"""Build recipe of the in-tree native libraries (gfx950 only).

  lib/libmptrac_hip.so         the HIP back end + C ABI (csrc/mphip_api.hip)
  lib/libmptrac_hip_exact.so   the same sources with the reference's roundings (EXACT_FLAGS): same ABI, same kernels

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with
the tree to the GPU box.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
HIP_LIB = os.path.join(LIBDIR, "libmptrac_hip.so")
# The reference-rounding build: every division of the reference an IEEE division (the default multiplies by reciprocals
# of grid constants rounded once on the host and uses a rcp + Newton quotient), the C library's square root, cosine and
# sine (restated in csrc/mphip_libm.h like exp / log / pow), and no contraction of a multiply and an add into a fused
# multiply-add (the reference's default CPU build, gcc -O3 without -march, has none; the fused operations inside the
# library functions are the C library's own and stay).  What this build computes is the oracle's bits: positions,
# quantities and cache->uvwp (tests/test_gpu_exact_library.py, tools/gpu_bit_census.py).
EXACT_LIB = os.path.join(LIBDIR, "libmptrac_hip_exact.so")
EXACT_FLAGS = ["-DMPHIP_EXACT_DIV=1", "-ffp-contract=off"]

# -disable-machine-licm: the machine-level loop-invariant code motion parks the ~60 double constants of
# the polynomial kernels (sincosf, log, exp, cos) in VGPR pairs for the whole particle loop; without it the
# fused step kernel needs 144 instead of 207 VGPRs and runs three waves per SIMD without scratch
# -fvisibility=hidden: the C ABI of include/mptrac_hip.h is what the library exports (mphip_device.hpp)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics",
               "-mllvm", "-disable-machine-licm", "-Wl,--version-script=" + os.path.join(CSRC, "abi.map")]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def hip_sources():
    src = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h", ".map"))]
    src.append(os.path.join(os.path.dirname(HERE), "include", "mptrac_hip.h"))
    return src


def build_variant(name, extra_flags, verbose=False):
    """Experimental build lib/libmptrac_hip_<name>.so (tuning sweeps; select
    it at run time with MPHIP_LIB=<path>)."""
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, f"libmptrac_hip_{name}.so")
    cmd = [_hipcc(), *HIPCC_FLAGS, *extra_flags, "-o", out, os.path.join(CSRC, "mphip_api.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    verify_async_loads(out, verbose)
    return out


def _build_lib(target, flags, force, verbose):
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _stale(target, hip_sources()):
        if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
            if os.path.exists(target):
                return target       # prebuilt library shipped with the tree
        cmd = [_hipcc(), *HIPCC_FLAGS, *flags, "-o", target, os.path.join(CSRC, "mphip_api.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        verify_async_loads(target, verbose)
    return target


def exact_requested():
    return os.environ.get("MPTRAC_AMD_EXACT", "0") not in ("", "0")


def build_hip(force=False, verbose=False, extra_flags=()):
    """The library the package loads: MPHIP_LIB=<path> (tuning builds), MPTRAC_AMD_EXACT=1 (the reference-rounding
    build), else lib/libmptrac_hip.so."""
    if os.environ.get("MPHIP_LIB"):
        return os.environ["MPHIP_LIB"]
    if exact_requested():
        return build_hip_exact(force, verbose)
    return _build_lib(HIP_LIB, list(extra_flags), force, verbose)


def build_hip_exact(force=False, verbose=False):
    return _build_lib(EXACT_LIB, EXACT_FLAGS, force, verbose)


def build_hip_both(force=False, verbose=False):
    """Both libraries, compiled side by side (each is one translation unit: ~3 minutes of one core)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as pool:
        jobs = [pool.submit(_build_lib, HIP_LIB, [], force, verbose), pool.submit(build_hip_exact, force, verbose)]
        return [j.result() for j in jobs]


def verify_async_loads(lib, verbose=False):
    """The corner caches issue gathers as inline assembly and wait for them later (csrc/mphip_device.hpp:
    load_wind_cached / wind_cache_wait); a library in which the compiler touched such a register before
    the wait would compute with stale data, so it is not kept."""
    from . import check_async_loads as chk
    try:
        hazards, kernels, nloads = chk.check(lib)
    except chk.ToolMissing as exc:      # (a box without the LLVM tools: a prebuilt library was checked where it was built)
        # never silent: a library that was compiled here and could not be checked is reported as unverified
        import sys
        print(f"mptrac_amd.build: machine-code check of {os.path.basename(lib)} NOT run ({exc}); the library is unverified",
              file=sys.stderr)
        return
    if hazards:
        bad = lib + ".rejected"
        os.replace(lib, bad)
        raise RuntimeError(f"{len(hazards)} use-before-wait hazards in {os.path.basename(lib)} (kept as {bad}), e.g. {hazards[0]}")
    if verbose:
        print(f"{os.path.basename(lib)}: {kernels} kernels, {nloads} loads, no register touched before its load was waited for")


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(LIBDIR, "libmptrac.so")
TRAC_BIN = os.path.join(LIBDIR, "trac")
ATM2GRID_BIN = os.path.join(LIBDIR, "atm2grid")
ATM_CONV_BIN = os.path.join(LIBDIR, "atm_conv")
MET_CONV_BIN = os.path.join(LIBDIR, "met_conv")
# small test extents by default; production builds pass the reference's -DNP=... -DEX=... values
HOST_DIMS = {"NP": 200000, "NQ": 15, "EX": 364, "EY": 186, "EP": 64}


def build_host(force=False, verbose=False, dims=None, outdir=None):
    """Host-side C library (the reference's mptrac_* interface on the C ABI)
    and the trac driver.  Plain gcc; links against libmptrac_hip.so.
    `outdir`: a second build beside the default one (production extents: tools/gpu_trac_dropin.py)."""
    os.makedirs(LIBDIR, exist_ok=True)
    dims = dict(HOST_DIMS, **(dims or {}))
    out = outdir or LIBDIR
    os.makedirs(out, exist_ok=True)
    host_lib, trac_bin = os.path.join(out, "libmptrac.so"), os.path.join(out, "trac")
    src = [os.path.join(HOST_DIR, f) for f in ("mptrac.c", "mptrac.h", "trac.c", "atm2grid.c", "atm_conv.c", "met_conv.c", "ctlfile.c", "nc_classic.c",
                                               "nc_classic.h", "nc_internal.h", "nc_hdf5.c", "rendezvous.c", "output.c")]
    lib_src = [os.path.join(HOST_DIR, f) for f in ("mptrac.c", "ctlfile.c", "nc_classic.c", "nc_hdf5.c", "rendezvous.c", "output.c")]
    if not (force or _stale(host_lib, src) or _stale(trac_bin, src)):
        return host_lib, trac_bin
    defs = [f"-D{k}={v}" for k, v in dims.items()]
    defs.append('-DMPTRAC_AMD_DATA_DIR="%s"' % os.path.join(HERE, "data"))
    common = ["gcc", "-O2", "-g", "-std=gnu99", "-Wall", "-W", "-Wno-format-security", "-fPIC", "-mcmodel=medium",
              *defs]
    rpath = ["-L" + LIBDIR, "-lmptrac_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-lpthread", "-lz"]
    cmds = [common + ["-shared", "-o", host_lib] + lib_src + rpath,
            common + ["-o", trac_bin, os.path.join(HOST_DIR, "trac.c")] + lib_src + rpath]
    if outdir is None:
        cmds += [common + ["-o", ATM2GRID_BIN, os.path.join(HOST_DIR, "atm2grid.c")] + lib_src + rpath,
                 common + ["-o", ATM_CONV_BIN, os.path.join(HOST_DIR, "atm_conv.c")] + lib_src + rpath,
                 common + ["-o", MET_CONV_BIN, os.path.join(HOST_DIR, "met_conv.c")] + lib_src + rpath]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return host_lib, trac_bin


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
