#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch bytes / LDS of every kernel of a built library (from the code object's metadata).
  python tools/kernel_regs.py [lib.so] [name filter]"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(lib):
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], cwd=tmp, capture_output=True, check=True)
        co = glob.glob(os.path.join(tmp, "*gfx950*"))[0]
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        def f(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, blk)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", f("name")], capture_output=True, text=True).stdout.strip()
        out.append((name, int(f("vgpr_count")), int(f("sgpr_count")), int(f("private_segment_fixed_size")),
                    int(f("group_segment_fixed_size"))))
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "mptrac_amd", "lib", "libmptrac_hip.so")
    flt = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for name, v, s, scr, lds in sorted(kernels(lib)):
        if all(f in name for f in flt):
            print(f"{v:4d} vgpr {s:4d} sgpr {scr:5d} scratch {lds:6d} lds  {name[:110]}")
