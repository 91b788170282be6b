"""The device's exp / log / pow (mptrac_amd/csrc/mphip_libm.h) are a restatement of the C library's -- here the same
header is compiled for the CPU (tests/c/libm_cpu.c) and compared bit by bit with the running libm, the one the oracle
and the reference's CPU build call (src/mptrac.c:4531-4546, 5822).  The device itself is compared with the library in
the GPU suite (test_gpu_parity.py::test_exp_log_pow_are_bit_identical_to_libm)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import libm_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mptrac_amd", "csrc")
_dp = C.POINTER(C.c_double)


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(ROOT, "tests", "c", "build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libm_cpu.so")
    # contraction off: every fused operation of the algorithms is an explicit fma in the header, nothing else may fuse
    # (-mfma only makes those fma() calls one instruction; without it they go through the library's fma(), same bits)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-std=gnu99", "-Wall", "-Wextra", "-I", CSRC,
           "-o", so, os.path.join(ROOT, "tests", "c", "libm_cpu.c"), "-lm"]
    if _has_fma():
        cmd.insert(1, "-mfma")
    subprocess.check_call(cmd)
    L = C.CDLL(so)
    for f in (L.cmp_exp, L.cmp_log, L.cmp_pow, L.cmp_cos, L.cmp_sin):
        f.restype = C.c_size_t
    return L


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _run(L, name, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    first = C.c_size_t(2 ** 63)
    if y is None:
        bad = getattr(L, "cmp_" + name)(_ptr(x), C.c_size_t(len(x)), C.byref(first))
        where = "x = %s" % (float(x[first.value]).hex() if bad else "")
    else:
        y = np.ascontiguousarray(y, dtype=np.float64)
        bad = L.cmp_pow(_ptr(x), _ptr(y), C.c_size_t(len(x)), C.byref(first))
        where = "x = %s, y = %s" % ((float(x[first.value]).hex(), float(y[first.value]).hex()) if bad else ("", ""))
    return bad, where


def test_the_ifunc_variant_matched_is_the_one_this_host_runs():
    """The restatement follows the FMA variants of glibc's exp / log / pow; a host without FMA + AVX2 would run the
    sse2 variants, whose last bits differ (and so would the oracle's) -- say so instead of failing obscurely."""
    flags = open("/proc/cpuinfo").read()
    if not (" fma " in flags and " avx2 " in flags):
        pytest.skip("host CPU without FMA + AVX2: glibc selects other variants here")


def test_restated_exp_log_pow_have_the_librarys_bits(lib):
    assert lib.check_constants() == 0
    rng = np.random.default_rng(20260930)
    n = 5_000_000
    total = 0
    for fn, sets in (("exp", libm_args.exp_sets), ("log", libm_args.log_sets)):
        for name, x in sets(rng, n):
            bad, where = _run(lib, fn, x)
            assert bad == 0, (fn, name, bad, where)
            total += len(x)
    for name, (x, y) in libm_args.pow_sets(rng, n):
        bad, where = _run(lib, "pow", x, y)
        assert bad == 0, ("pow", name, bad, where)
        total += len(x)
    assert total > 1e8


def test_restated_cos_sin_have_the_librarys_bits(lib):
    """glibc's cos / sin (s_sin.c) inside |x| < 2.426 -- DX2DEG's cos(latitude) of the reference-rounding build and
    ZETA's sin: the restatement of mphip_libm.h against the running library, 3 x 10^7 arguments each."""
    rng = np.random.default_rng(20261001)
    total = 0
    for name, x in libm_args.sincos_sets(rng, 4_000_000):
        for fn in ("cos", "sin"):
            bad, where = _run(lib, fn, x)
            assert bad == 0, (fn, name, bad, where)
        total += len(x)
    assert total > 2.5e7


def test_the_check_has_teeth(lib):
    """The same comparison against arguments shifted by one ulp finds differences -- the counters do count."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-10.0, 10.0, 100000)
    out = np.empty_like(x)
    lib.rst_exp(_ptr(x), C.c_size_t(len(x)), _ptr(out))
    ref = np.exp(np.nextafter(x, np.inf))
    assert np.count_nonzero(out != ref) > 1000


def test_polynomial_literals_of_the_header_are_the_librarys(lib):
    """mphip_libm.h carries the polynomial coefficients as literals; they must be the values tools/gen_libm_tables.py
    read from the library (mphip_libmtab.h)."""
    text = open(os.path.join(CSRC, "mphip_libm.h")).read()
    tab = open(os.path.join(CSRC, "mphip_libmtab.h")).read()
    hexf = r"-?0x[01]\.[0-9a-f]+p[+-]?\d+"
    literals = {float.fromhex(m) for m in re.findall(hexf, text)}
    for array in ("mphip_libm_exp_k", "mphip_libm_ln2", "mphip_libm_log_a", "mphip_libm_log_b", "mphip_libm_pow_a",
                  "mphip_libm_sincos_k"):
        body = re.search(array + r"\[\d+\] = \{(.*?)\};", tab, re.S).group(1)
        values = [float.fromhex(m) for m in re.findall(hexf, body)]
        assert values, array
        for v in values:
            assert v in literals, (array, v.hex())


def test_committed_tables_are_the_running_librarys():
    """The generated header against the libm.so.6 of the machine the tests run on (skipped where it cannot be read)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_libm_tables as g
    try:
        found = g.locate(g.rodata(g.find_libm()))
    except SystemExit as exc:
        pytest.skip(str(exc))
    tab = open(os.path.join(CSRC, "mphip_libmtab.h")).read()

    def init(name):
        body = tab[tab.index("#define " + name):]
        body = body[:body.index("\n\n")]
        return re.findall(r"-?0x[0-9a-f.]+(?:p[+-]?\d+)?(?:ULL)?", body)

    log_tab = [float.fromhex(v) for v in init("MPHIP_LIBM_LOG_TAB_INIT")]
    assert log_tab == found["log"]["tab"]
    exp_tab = [int(v[:-3], 16) for v in init("MPHIP_LIBM_EXP_TAB_INIT")]
    assert exp_tab == found["exp"]["tab"]
    pow_tab = [float.fromhex(v) for v in init("MPHIP_LIBM_POW_TAB_INIT")]
    lib_pow = found["pow"]["tab"]
    assert pow_tab == [lib_pow[4 * i + k] for i in range(128) for k in (0, 2, 3)]
    assert [float.fromhex(v) for v in init("MPHIP_LIBM_SINCOS_TAB_INIT")] == found["sincos"]["tab"]
