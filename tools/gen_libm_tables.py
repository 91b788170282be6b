#!/usr/bin/env python3
"""Generates mptrac_amd/csrc/mphip_libmtab.h: the constants of the C library's double-precision exp / log / pow.

The reference's CPU build links glibc's libm; since glibc 2.28 its exp, log and pow are the ARM optimized-routines
algorithms (Szabolcs Nagy, 2018; published under MIT / Apache-2.0 WITH LLVM-exception in ARM-software/optimized-routines,
math/exp.c, log.c, pow.c): a 128-entry table reduction and a short polynomial.  The device restates those
algorithms (mptrac_amd/csrc/mphip_device.hpp: libm_exp / libm_log / libm_pow) so that its results are the C library's
bits; the tables -- 2^(k/128), 1/c and log c -- cannot be re-derived from first principles (the 1/c and log c of
log / pow were picked by the authors' search so that k ln2_hi + log c is exact), so they are READ from the libm.so.6
this image ships, located by their leading constants and checked for internal consistency.  Nothing else is taken
from the library; the polynomial coefficients sit in the same data blocks and are read with them.

sin / cos (used by the reference-rounding build only: DX2DEG's cos(latitude), ZETA's sin) are glibc's own
sysdeps/ieee754/dbl-64/s_sin.c (IBM Accurate Mathematical Library, LGPL; algorithm published there and in
"usncs.h" / "sincostab.c"): for |x| < 2.426 a 440-entry table {sin, its tail, cos, its tail} at k / 128 and two short
polynomials.  The table is read from the library like the others (located by its first entry {0, 0, 1, 0} and checked
against sin / cos of k / 128); the eleven polynomial constants, pi/2 in two parts and the shifter 1.5 x 2^45 are the
published ones, and each is checked to be present in the library's read-only data.

Layouts (the published headers math_config.h / glibc sysdeps/ieee754/dbl-64/math_config.h):
  exp_data      { invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5], uint64 tab[2 * 128] }
  log_data      { ln2hi, ln2lo, poly[5], poly1[11], {invc, logc} tab[128], {chi, clo} tab2[128] }
  pow_log_data  { ln2hi, ln2lo, poly[7], {invc, pad, logc, logctail} tab[128] }

Usage: tools/gen_libm_tables.py [path/to/libm.so.6]      (re-run only if the matched glibc changes; the header is committed)
"""
import math
import os
import struct
import sys

N = 128
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mptrac_amd", "csrc", "mphip_libmtab.h")


def find_libm():
    for cand in ("/lib/x86_64-linux-gnu/libm.so.6", "/usr/lib/x86_64-linux-gnu/libm.so.6", "/lib64/libm.so.6", "/usr/lib64/libm.so.6"):
        if os.path.exists(cand):
            return cand
    raise SystemExit("libm.so.6 not found; pass its path")


def rodata(path):
    """(bytes, virtual address) of every allocated read-only PROGBITS section of an ELF64 little-endian file."""
    blob = open(path, "rb").read()
    assert blob[:4] == b"\x7fELF" and blob[4] == 2 and blob[5] == 1, "ELF64 LE expected"
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", blob, 0x3A)
    out = []
    for i in range(shnum):
        _, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
        if typ == 1 and (flags & 2) and not (flags & 1) and not (flags & 4):    # PROGBITS, ALLOC, not WRITE, not EXEC
            out.append((blob[off:off + size], addr))
    return out


def doubles(buf, off, n):
    return list(struct.unpack_from("<%dd" % n, buf, off))


def find_all(buf, pattern):
    pos, hits = 0, []
    while True:
        pos = buf.find(pattern, pos)
        if pos < 0:
            return hits
        if pos % 8 == 0:
            hits.append(pos)
        pos += 8


def locate(sections):
    found = {}
    ln2hi, ln2lo = float.fromhex("0x1.62e42fefa3800p-1"), float.fromhex("0x1.ef35793c76730p-45")
    sig_exp = struct.pack("<2d", float.fromhex("0x1.71547652b82fep0") * N, float.fromhex("0x1.8p52"))
    sig_log = struct.pack("<2d", ln2hi, ln2lo)
    for buf, addr in sections:
        for off in find_all(buf, sig_exp):
            head = doubles(buf, off, 14)
            tab = list(struct.unpack_from("<%dQ" % (2 * N), buf, off + 14 * 8))
            # 2^(k/N) = asdouble(tab[2k+1] + (k << 45)) (1 + asdouble(tab[2k]))
            ok = all(abs(struct.unpack("<d", struct.pack("<Q", tab[2 * k + 1] + (k << 45)))[0] - 2.0 ** (k / N)) < 1e-15 for k in range(N))
            if ok and "exp" not in found:
                found["exp"] = dict(addr=addr + off, head=head, tab=tab)
        for off in find_all(buf, sig_log):
            # try the log layout, then the pow layout; accept the one whose table is consistent
            try:
                poly, poly1 = doubles(buf, off + 16, 5), doubles(buf, off + 16 + 40, 11)
                tab = doubles(buf, off + 16 + 40 + 88, 2 * N)
                if all(0.7 < tab[2 * i] < 1.5 and abs(tab[2 * i + 1] + math.log(tab[2 * i])) < 1e-9 for i in range(N)) and "log" not in found:
                    found["log"] = dict(addr=addr + off, poly=poly, poly1=poly1, tab=tab)
                    continue
            except struct.error:
                pass
            try:
                poly = doubles(buf, off + 16, 7)
                tab = doubles(buf, off + 16 + 56, 4 * N)
                if all(0.7 < tab[4 * i] < 1.5 and tab[4 * i + 1] == 0.0 and abs(tab[4 * i + 2] + math.log(tab[4 * i])) < 1e-9
                       and abs(tab[4 * i + 3]) < 1e-12 for i in range(N)) and "pow" not in found:
                    found["pow"] = dict(addr=addr + off, poly=poly, tab=tab)
            except struct.error:
                pass
    sig_sincos = struct.pack("<4d", 0.0, 0.0, 1.0, 0.0)
    for buf, addr in sections:
        for off in find_all(buf, sig_sincos):
            try:
                tab = doubles(buf, off, 440)
            except struct.error:
                continue
            if all(abs(tab[4 * k] - math.sin(k / 128)) < 1e-15 and abs(tab[4 * k + 2] - math.cos(k / 128)) < 1e-15
                   and abs(tab[4 * k + 1]) < 2e-16 and abs(tab[4 * k + 3]) < 2e-16 for k in range(110)) and "sincos" not in found:
                found["sincos"] = dict(addr=addr + off, tab=tab)
    if "sincos" in found:
        every = b"".join(buf for buf, _ in sections)
        for name, v in SINCOS_K:
            if not find_all(every, struct.pack("<d", float.fromhex(v))):
                raise SystemExit("sin / cos constant %s = %s is not in the library's read-only data" % (name, v))
    missing = {"exp", "log", "pow", "sincos"} - set(found)
    if missing:
        raise SystemExit("not found in the library's read-only data: %s" % sorted(missing))
    found["ln2"] = (ln2hi, ln2lo)
    return found


# s_sin.c / usncs.h: sn3, sn5, cs2, cs4, cs6 (table path), s1..s5 (Taylor path), hp0 + hp1 = pi / 2, big = 1.5 x 2^45, 0.126
SINCOS_K = [("sn3", "-0x1.5555555555515p-3"), ("sn5", "0x1.11110e829872fp-7"), ("cs2", "0x1.0000000000000p-1"),
            ("cs4", "-0x1.5555555555535p-5"), ("cs6", "0x1.6c16bedd9e239p-10"), ("s1", "-0x1.5555555555555p-3"),
            ("s2", "0x1.1111111110ecep-7"), ("s3", "-0x1.a01a019db08b8p-13"), ("s4", "0x1.71de27b9a7ed9p-19"),
            ("s5", "-0x1.addffc2fcdf59p-26"), ("hp0", "0x1.921fb54442d18p+0"), ("hp1", "0x1.1a62633145c07p-54"),
            ("big", "0x1.8000000000000p+45"), ("taylor_below", "0x1.020c49ba5e354p-3")]


def hexd(v):
    return float.hex(v)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else find_libm()
    f = locate(rodata(path))
    ver = os.popen("ldd --version 2>/dev/null | head -1").read().strip()
    e, lg, pw = f["exp"], f["log"], f["pow"]
    with open(OUT, "w") as o:
        w = o.write
        w("/* generated by tools/gen_libm_tables.py -- do not edit.\n"
          " * Constants of the C library's double exp / log / pow (ARM optimized-routines algorithms, glibc >= 2.28),\n"
          " * read from %s (%s).\n"
          " * Plain C so that the device code (C++) and the CPU restatement under tests/c/ share one file. */\n" % (path, ver))
        w("#ifndef MPHIP_LIBMTAB_H\n#define MPHIP_LIBMTAB_H\n#include <stdint.h>\n\n")
        w("#define MPHIP_LIBM_N 128\n\n")
        h = e["head"]
        w("/* exp: InvLn2N, Shift, NegLn2hiN, NegLn2loN, C2..C5 */\n")
        w("static const double mphip_libm_exp_k[8] = {\n  %s\n};\n" % ",\n  ".join(hexd(v) for v in h[:8]))
        w("/* log and pow: Ln2hi, Ln2lo */\n")
        w("static const double mphip_libm_ln2[2] = { %s, %s };\n" % (hexd(f["ln2"][0]), hexd(f["ln2"][1])))
        w("/* log: A0..A4 (away from 1), B0..B10 (near 1) */\n")
        w("static const double mphip_libm_log_a[5] = {\n  %s\n};\n" % ",\n  ".join(hexd(v) for v in lg["poly"]))
        w("static const double mphip_libm_log_b[11] = {\n  %s\n};\n" % ",\n  ".join(hexd(v) for v in lg["poly1"]))
        w("/* pow: A0..A6 of its log */\n")
        w("static const double mphip_libm_pow_a[7] = {\n  %s\n};\n\n" % ",\n  ".join(hexd(v) for v in pw["poly"]))
        w("/* The three tables as initialiser lists (the device builds one object of them, mphip_device.hpp: LibmBlob).\n"
          " * log: {invc, logc} x 128 */\n")
        w("#define MPHIP_LIBM_LOG_TAB_INIT \\\n")
        w(", \\\n".join("  %s, %s" % (hexd(lg["tab"][2 * i]), hexd(lg["tab"][2 * i + 1])) for i in range(N)) + "\n\n")
        w("/* exp: {bits of the tail, bits of 2^(k/128) - (k << 45)} x 128 */\n")
        w("#define MPHIP_LIBM_EXP_TAB_INIT \\\n")
        w(", \\\n".join("  0x%016xULL, 0x%016xULL" % (e["tab"][2 * k], e["tab"][2 * k + 1]) for k in range(N)) + "\n\n")
        w("/* pow: {invc, logc, logctail} x 128 (the library's unused pad member dropped) */\n")
        w("#define MPHIP_LIBM_POW_TAB_INIT \\\n")
        w(", \\\n".join("  %s, %s, %s" % (hexd(pw["tab"][4 * i]), hexd(pw["tab"][4 * i + 2]), hexd(pw["tab"][4 * i + 3])) for i in range(N)) + "\n\n")
        sc = f["sincos"]
        w("/* sin / cos: sn3, sn5, cs2, cs4, cs6, s1..s5, hp0, hp1, big, 0.126 */\n")
        w("static const double mphip_libm_sincos_k[%d] = {\n  %s\n};\n" % (len(SINCOS_K), ",\n  ".join(v for _, v in SINCOS_K)))
        w("/* sin / cos: {sin, tail, cos, tail} of k / 128, k = 0 .. 109 */\n")
        w("#define MPHIP_LIBM_SINCOS_TAB_INIT \\\n")
        w(", \\\n".join("  %s, %s, %s, %s" % tuple(hexd(v) for v in sc["tab"][4 * k:4 * k + 4]) for k in range(110)) + "\n\n")
        w("#ifndef __HIPCC__\n")
        w("static const double mphip_libm_log_tab[2 * MPHIP_LIBM_N] = { MPHIP_LIBM_LOG_TAB_INIT };\n")
        w("static const uint64_t mphip_libm_exp_tab[2 * MPHIP_LIBM_N] = { MPHIP_LIBM_EXP_TAB_INIT };\n")
        w("static const double mphip_libm_pow_tab[3 * MPHIP_LIBM_N] = { MPHIP_LIBM_POW_TAB_INIT };\n")
        w("static const double mphip_libm_sincos_tab[440] = { MPHIP_LIBM_SINCOS_TAB_INIT };\n")
        w("#endif\n\n#endif\n")
    print("wrote %s: exp_data @%#x, log_data @%#x, pow_log_data @%#x, sincostab @%#x of %s"
          % (OUT, e["addr"], lg["addr"], pw["addr"], f["sincos"]["addr"], path))


if __name__ == "__main__":
    main()
