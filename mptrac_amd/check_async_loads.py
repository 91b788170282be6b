#!/usr/bin/env python3
"""No instruction may touch the destination of a load that has not been waited for.

The wind / model-level corner caches issue their gathers as inline assembly and wait for them in a second
asm statement, so that the work between the two overlaps the gather (mphip_device.hpp: load_wind_cached,
wind_cache_wait, load_ml_cached).  The compiler does not know about those loads: if the register allocator
moved, copied or spilled one of the destination registers between the load and the wait, the kernel would
read stale data -- silently, and only in the build where it happens.  This check reads the machine code of the
built library and fails when that happened:

  for every VMEM load, its destination registers are "in flight" until an s_waitcnt vmcnt(N) retires it
  (loads and stores retire in order on gfx9); an instruction that reads or writes an in-flight register
  before that is reported.

The scan is linear over each kernel (state dropped after an unconditional branch), which is exact for the
forward, structured code these kernels compile to; compiler-generated loads pass by construction, so a
report always points at one of the asm sequences.

  python -m mptrac_amd.check_async_loads [lib.so]        exit status 1 when a hazard is found
(mptrac_amd.build.build_hip runs it on every library it compiles and refuses a library with a hazard)
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(HERE, "lib", "libmptrac_hip.so")

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
WAIT = re.compile(r"vmcnt\((\d+)\)")
VMEM_LOAD = re.compile(r"^(global_load|buffer_load|scratch_load|flat_load)")
VMEM_STORE = re.compile(r"^(global_store|buffer_store|scratch_store|flat_store|global_atomic|buffer_atomic|flat_atomic)")


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], cwd=tmp, capture_output=True, check=True)
        co = glob.glob(os.path.join(tmp, "*gfx950*"))[0]
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True,
                              text=True, check=True).stdout


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, lines):
    """lines: [(address, instruction text)]; returns a list of hazard descriptions"""
    hazards = []
    inflight = []          # VMEM operations not yet retired, oldest first: (address, text, set of destination VGPRs)
    for addr, ins in lines:
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = WAIT.search(ins)
            if m:
                keep = int(m.group(1))
                inflight = inflight[len(inflight) - keep:] if keep else []
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            inflight = []
            continue
        touched = regs_of(ins.split("//")[0])
        if VMEM_LOAD.match(op):     # a later load may overwrite the destination of an earlier one: they return in order
            touched = regs_of(ins.split(None, 1)[1].split(",", 1)[1])
        for a0, t0, dst in inflight:
            if dst & touched:
                hazards.append(f"{name}: {addr} `{ins.strip()}` touches v{sorted(dst & touched)} of the load at {a0} `{t0.strip()}` "
                               "before it was waited for")
        if VMEM_LOAD.match(op):
            first = ins.split(None, 1)[1].split(",")[0]
            inflight.append((addr, ins, regs_of(first)))
        elif VMEM_STORE.match(op):
            returns = "sc0" in ins and "atomic" in op      # an atomic with return value writes its first operand
            inflight.append((addr, ins, regs_of(ins.split(None, 1)[1].split(",")[0]) if returns else set()))
    return hazards


def check(lib=DEFAULT_LIB):
    text = disassemble(lib)
    hazards, kernels, nloads = [], 0, 0
    name, lines = None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            if name and lines:
                hazards += check_kernel(name, lines)
                kernels += 1
            name, lines = m.group(1), []
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and name:
            lines.append((m.group(2), m.group(1)))
            nloads += bool(VMEM_LOAD.match(m.group(1)))
    if name and lines:
        hazards += check_kernel(name, lines)
        kernels += 1
    return hazards, kernels, nloads


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    hazards, kernels, nloads = check(lib)
    per_kernel = {}
    for h in hazards:
        per_kernel.setdefault(h.split(":")[0], []).append(h)
    for name, hs in per_kernel.items():
        print(f"{name}: {len(hs)} hazards, the first ones:")
        for h in hs[:4]:
            print("   " + h.split(": ", 1)[1])
    print(f"{os.path.basename(lib)}: {kernels} kernels, {nloads} loads, {len(hazards)} use-before-wait hazards")
    sys.exit(1 if hazards else 0)
