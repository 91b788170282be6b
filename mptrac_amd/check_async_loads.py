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

The in-flight set is followed along every path of a kernel's control flow (both sides of a conditional branch, the
target of an unconditional one, loop back-edges included); compiler-generated loads pass by construction, so a
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


class ToolMissing(RuntimeError):
    """llvm-objdump (or the code object inside the library) is not where this image keeps it"""


def disassemble(lib):
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        raise ToolMissing(f"{objdump} not found: the machine-code check of {os.path.basename(lib)} cannot run here")
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], cwd=tmp, capture_output=True, check=True)
        found = glob.glob(os.path.join(tmp, "*gfx950*"))
        if not found:
            raise ToolMissing(f"no gfx950 code object found inside {os.path.basename(lib)}")
        co = found[0]
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


BRANCH = re.compile(r"^s_c?branch")


def _decode(lines):
    """Per instruction: (address, text, kind, simm16 of a branch, registers it touches, destination registers of a load)"""
    out = []
    for addr, ins in lines:
        op = ins.split()[0]
        kind, imm, touched, dst = "other", 0, set(), set()
        if op == "s_waitcnt":
            m = WAIT.search(ins)
            kind, imm = ("wait", int(m.group(1))) if m else ("other", 0)
        elif op == "s_branch":
            kind, imm = "jump", int(ins.split()[1])
        elif op.startswith("s_cbranch"):
            kind, imm = "branch", int(ins.split()[1])
        elif op in ("s_endpgm", "s_setpc_b64"):
            kind = "end"
        elif op == "s_swappc_b64":      # a call: the callee starts with s_waitcnt vmcnt(0) (AMDGPU calling convention)
            kind, imm = "wait", 0
        else:
            touched = regs_of(ins.split("//")[0])
            if VMEM_LOAD.match(op):   # a later load may overwrite the destination of an earlier one: they return in order
                touched = regs_of(ins.split(None, 1)[1].split(",", 1)[1])
                kind, dst = "vmem", regs_of(ins.split(None, 1)[1].split(",")[0])
            elif VMEM_STORE.match(op):
                returns = "sc0" in ins and "atomic" in op      # an atomic with return value writes its first operand
                kind, dst = "vmem", (regs_of(ins.split(None, 1)[1].split(",")[0]) if returns else set())
        out.append((addr, ins, kind, imm, touched, dst))
    return out


def check_kernel(name, lines):
    """lines: [(address, instruction text)]; returns a list of hazard descriptions.

    Data-flow over the kernel's control flow.  VMEM operations retire in order, so after `s_waitcnt vmcnt(N)` a load is
    still in flight exactly if fewer than N operations were issued behind it.  The state at an instruction is therefore
    {load: smallest number of VMEM operations issued behind it on any path that reaches the instruction}; a join takes
    the union with the smaller counts, an issue raises every count by one, a wait drops the loads with count >= N.
    Blocks are re-visited until nothing changes, so a load that is still in flight at the back-edge of a loop (the
    multi-step and Runge-Kutta loops) is checked against the instructions at the loop head as well."""
    code = _decode(lines)
    n = len(code)
    index = {int(a, 16): i for i, (a, *_rest) in enumerate(code)}
    CAP = 64                                   # (vmcnt holds 6 bits: any wait retires a load with 64 or more behind it)
    entry = {0: {}}                            # block leader -> state at its entry
    work = [0]
    hazards, seen_hazard = [], set()

    def flow_into(target, state):
        old = entry.get(target)
        if old is None:
            entry[target] = dict(state)
            work.append(target)
            return
        changed = False
        for k, v in state.items():
            if k not in old or v < old[k]:
                old[k] = v
                changed = True
        if changed:
            work.append(target)

    leaders = {0}
    for i, (addr, ins, kind, imm, touched, dst) in enumerate(code):
        if kind in ("jump", "branch"):
            off = imm - 65536 if imm >= 32768 else imm
            t = index.get(int(addr, 16) + 4 + 4 * off)
            if t is not None:
                leaders.add(t)
            if i + 1 < n:
                leaders.add(i + 1)
    while work:
        i = work.pop()
        state = dict(entry[i])
        first = True
        while i < n:
            if not first and i in leaders:     # the next block: hand the state over and stop here
                flow_into(i, state)
                break
            first = False
            addr, ins, kind, imm, touched, dst = code[i]
            if kind == "wait":
                state = {k: v for k, v in state.items() if v < imm} if imm else {}
            elif kind in ("jump", "branch"):
                off = imm - 65536 if imm >= 32768 else imm
                t = index.get(int(addr, 16) + 4 + 4 * off)
                if t is not None:
                    flow_into(t, state)
                if kind == "jump":
                    break
            elif kind == "end":
                break
            else:
                if touched:
                    for j in state:
                        if code[j][5] & touched and (j, i) not in seen_hazard:
                            seen_hazard.add((j, i))
                            hazards.append(f"{name}: {addr} `{ins.strip()}` touches v{sorted(code[j][5] & touched)} of the load "
                                           f"at {code[j][0]} `{code[j][1].strip()}` before it was waited for")
                if kind == "vmem":
                    state = {k: min(v + 1, CAP) for k, v in state.items()}
                    if dst:
                        state[i] = 0
            i += 1
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
