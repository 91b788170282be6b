#!/usr/bin/env python3
"""Registers, scratch and LDS of the kernels in a built library (from the code object's metadata notes):
  tools/kernel_resources.py [lib.so] [name substring ...]
A lean step kernel must show <= 128 VGPRs (four waves per SIMD) and no scratch."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def resources(lib):
    with tempfile.TemporaryDirectory() as tmp:
        import glob
        import shutil
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], cwd=tmp, capture_output=True)
        co = glob.glob(os.path.join(tmp, "*gfx950*"))[0]      # the embedded code object, extracted next to the copy
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = []
    for block in notes.split("- .agpr_count:")[1:]:
        def field(name):
            m = re.search(r"\." + name + r":\s+(\S+)", block)
            return m.group(1) if m else "?"
        out.append({"name": field("name"), "vgpr": field("vgpr_count"), "agpr": block.split()[0], "sgpr": field("sgpr_count"),
                    "scratch": field("private_segment_fixed_size"), "lds": field("group_segment_fixed_size"),
                    "spill_v": field("vgpr_spill_count"), "spill_s": field("sgpr_spill_count")})
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(ROOT, "mptrac_amd", "lib", "libmptrac_hip.so")
    for r in resources(lib):
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        if not args or any(a in name for a in args):
            print(f'{name[:70]:70s} vgpr {r["vgpr"]:>4s} agpr {r["agpr"]:>3s} sgpr {r["sgpr"]:>4s} scratch {r["scratch"]:>5s} '
                  f'lds {r["lds"]:>6s} spills v{r["spill_v"]} s{r["spill_s"]}')
