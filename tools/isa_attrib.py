#!/usr/bin/env python3
"""Static instruction mix of one kernel, attributed to source functions.

  hipcc ... -gline-tables-only --cuda-device-only -S mphip_api.hip -o api.s
  tools/isa_attrib.py api.s step_kernelILj255E [--by line|func] [--blocks]

Every instruction is assigned to the innermost source line of its last `.loc`
(file, line) and from there to the enclosing __device__ function of
mphip_device.hpp / mphip_kernels.hpp.  The counts are static (each instruction
once); the fused step kernel is almost straight-line code per particle, so
they track the dynamic mix the PMC counters report (tools/profile_valu_mix.sh).
"""
import argparse
import collections
import os
import re
import sys

CLASSES = [
    ("fp64", re.compile(r"^v_(add|mul|fma|fmac|min|max|fract|floor|ceil|rndne|trunc|ldexp|frexp_mant|div_scale|div_fmas|div_fixup)_f64")),
    ("fp64t", re.compile(r"^v_(rcp|rsq|sqrt)_f64")),
    ("cmp64", re.compile(r"^v_cmp[x]?_\w+_(f64|u64|i64)|^v_cmp_class_f64")),
    ("cvt", re.compile(r"^v_cvt_")),
    ("fp32", re.compile(r"^v_(pk_)?(add|sub|mul|fma|mac|fmac|min|max|rcp|rsq|sqrt|exp|log|fract|floor|ldexp|mad|med3)_f32|^v_subrev_f32|^v_fmaak|^v_fmamk")),
    ("int64", re.compile(r"^v_(lshlrev|lshrrev|ashrrev)_[bi]64|^v_mad_[ui]64|^v_lshl_add_u64|^v_mul_hi_u32|^v_mul_lo_u32|^v_add_co|^v_addc_co|^v_subb?_co|^v_subrev_co")),
    ("sel", re.compile(r"^v_cndmask")),
    ("cmp32", re.compile(r"^v_cmp")),
    ("mov", re.compile(r"^v_mov|^v_accvgpr|^v_readlane|^v_writelane|^v_readfirstlane|^v_swap")),
    ("int32", re.compile(r"^v_")),
    ("vmem", re.compile(r"^(global|buffer|flat|scratch)_")),
    ("lds", re.compile(r"^ds_")),
    ("smem", re.compile(r"^s_(load|buffer_load)")),
    ("wait", re.compile(r"^s_waitcnt|^s_nop|^s_barrier")),
    ("branch", re.compile(r"^s_(c?branch|setpc|swappc|endpgm)")),
    ("salu", re.compile(r"^s_")),
]
VALU = ("fp64", "fp64t", "cmp64", "cvt", "fp32", "int64", "sel", "cmp32", "mov", "int32")


def classify(mn):
    for name, rx in CLASSES:
        if rx.match(mn):
            return name
    return "other"


def function_table(path):
    """[(first_line, name)] of the function definitions in a source file."""
    out = []
    rx = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:__device__|__global__)[^;{]*?\b([A-Za-z_]\w*)\s*\(")
    try:
        lines = open(path).read().split("\n")
    except OSError:
        return out
    for i, l in enumerate(lines, 1):
        m = rx.match(l)
        if m and "operator" not in l:
            out.append((i, m.group(1)))
        elif "operator()" in l and "__device__" in l:
            out.append((i, "hook"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--by", default="func", choices=("func", "line"))
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()

    files = {}
    tables = {}
    cur = None
    inside = False
    counts = collections.defaultdict(collections.Counter)
    total = collections.Counter()
    rx_file = re.compile(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"')
    rx_loc = re.compile(r"^\s*\.loc\s+(\d+)\s+(\d+)")
    for raw in open(args.asm):
        m = rx_file.match(raw)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3))
            continue
        if "Begin function" in raw:
            inside = args.kernel in raw
            continue
        if not inside:
            continue
        if raw.startswith(".Lfunc_end") or ".end_amdhsa_kernel" in raw:
            inside = False
            continue
        m = rx_loc.match(raw)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        s = raw.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        mn = s.split()[0]
        if not re.match(r"^[a-z_0-9]+$", mn):
            continue
        cls = classify(mn)
        key = "?"
        if cur:
            f = files.get(cur[0], "?")
            base = os.path.basename(f)
            if args.by == "line":
                key = f"{base}:{cur[1]}"
            else:
                if f not in tables:
                    tables[f] = function_table(f)
                name = None
                for first, n in tables[f]:
                    if first <= cur[1]:
                        name = n
                    else:
                        break
                key = f"{base.replace('mphip_', '').replace('.hpp', '')}:{name}" if name else base
        counts[key][cls] += 1
        total[cls] += 1

    cols = [c for c, _ in CLASSES]
    def row(name, c):
        valu = sum(c[k] for k in VALU)
        return f"{name[:44]:44s} {valu:6d} " + " ".join(f"{c[k]:6d}" for k in cols)
    print(f"{'where':44s} {'VALU':>6s} " + " ".join(f"{k:>6s}" for k in cols))
    print(row("TOTAL", total))
    for name, c in sorted(counts.items(), key=lambda kv: -sum(kv[1][k] for k in VALU))[: args.top]:
        print(row(name, c))


if __name__ == "__main__":
    main()
