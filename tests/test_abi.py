"""The C-ABI library builds for gfx950, loads on a CPU-only box, exports every
symbol include/mptrac_hip.h declares, and refuses to run without a device
(no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from mptrac_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mptrac_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(mphip_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_library_exports_every_declared_symbol():
    L = hip.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mptrac_hip.h but not exported"


def test_both_builds_export_the_c_abi_and_nothing_else():
    """libmptrac_hip.so and libmptrac_hip_exact.so (the reference-rounding build, mptrac_amd/build.py): the same dynamic
    symbols -- every declaration of the header, and nothing beside mphip_* (csrc/abi.map: two builds in one process,
    LD_PRELOAD in front of the C driver, must not interpose on each other's kernel handles)."""
    import subprocess
    from mptrac_amd import build as b
    b.build_hip_both()
    exported = []
    for lib in (b.HIP_LIB, b.EXACT_LIB):
        out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
        names = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
        assert names and all(n.startswith("mphip_") for n in names), [n for n in names if not n.startswith("mphip_")][:5]
        assert set(declared_symbols()) <= set(names)
        exported.append(names)
    assert exported[0] == exported[1]
    L = C.CDLL(b.EXACT_LIB)
    L.mphip_version.restype = C.c_char_p
    assert b"reference rounding" in L.mphip_version() and b"reference rounding" not in hip.load().mphip_version()


def test_struct_layouts_match_the_python_mirrors():
    L = hip.load()
    assert L.mphip_sizeof_ctl() == C.sizeof(hip.MphipCtl)
    assert L.mphip_sizeof_met() == C.sizeof(hip.MphipMet)


def test_header_cites_the_reference_interface():
    txt = open(os.path.join(ROOT, "include", "mptrac_hip.h")).read()
    for ref in ("mptrac_update_device", "mptrac_update_host", "mptrac_run_timestep", "mptrac_alloc",
                "mptrac_get_met", "mptrac.c:7851"):
        assert ref in txt


def _have_gpu():
    L = hip.load()
    h = C.c_void_p()
    rc = L.mphip_create(C.byref(h), 0)
    if rc == 0:
        L.mphip_destroy(h)
    return rc == 0


def test_no_cpu_fallback_without_a_device():
    if _have_gpu():
        pytest.skip("a HIP device is present")
    from cases import make_case
    ctl, clim, m0, m1, atm = make_case("advect", n=10, grid="tiny")
    with pytest.raises(hip.MphipError):
        hip.Simulation(ctl, clim, m0, m1, atm)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    touch oracle/."""
    pkg = os.path.join(ROOT, "mptrac_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"mptrac_oracle", r"orc_[a-z_]+\(", r"oracle/"):
                    assert not re.search(pat, txt, flags=re.M), f"{f} reaches into the oracle ({pat})"


def test_integration_glue_names_only_members_the_reference_has():
    """integration/mptrac_hip_glue.c (route A of INTEGRATION.md) is written against the reference's own
    structs; every ctl_t / met_t / clim_t / cache_t / atm_t member it names must exist in the reference's
    src/mptrac.h and every member of mphip_ctl_t must be filled.  Runs where the reference tree is present."""
    import subprocess
    import sys
    if not os.path.exists("/root/reference/src/mptrac.h"):
        pytest.skip("reference tree not present on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "integration", "check_glue_fields.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()


def test_integration_glue_type_checks_against_the_reference_header():
    """Route A is more than a claim: integration/mptrac_hip_glue.c compiles (gcc -fsyntax-only) in one unit with
    the reference's own src/mptrac.h -- every member access, argument and pointer type is checked by the
    compiler.  integration/typecheck_glue.py explains the empty stand-ins for the absent GSL / netCDF headers
    (nothing is built).  Runs where the reference tree is present."""
    import subprocess
    import sys
    if not os.path.exists("/root/reference/src/mptrac.h"):
        pytest.skip("reference tree not present on this machine")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "integration", "typecheck_glue.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()


def test_async_load_check_sees_a_register_touched_before_its_wait():
    """The checker itself: a spill of an in-flight register is a hazard, a later load into it is not."""
    from mptrac_amd import check_async_loads as chk
    spill = [("10", "global_load_dwordx4 v[30:33], v[18:19], off offset:32"),
             ("18", "scratch_store_dwordx4 off, v[30:33], off offset:240"),
             ("20", "s_waitcnt vmcnt(0)")]
    assert len(chk.check_kernel("k", spill)) == 1
    fine = [("10", "global_load_dwordx4 v[30:33], v[18:19], off"),
            ("14", "v_add_f64 v[2:3], v[4:5], v[6:7]"),
            ("18", "global_load_dwordx2 v[30:31], v[8:9], off"),        # loads return in order
            ("1c", "s_waitcnt vmcnt(0)"),
            ("20", "v_mov_b32_e32 v1, v30")]
    assert chk.check_kernel("k", fine) == []
    partial = [("10", "global_load_dword v1, v[18:19], off"),
               ("14", "global_load_dword v2, v[18:19], off offset:4"),
               ("18", "s_waitcnt vmcnt(1)"),
               ("1c", "v_mov_b32_e32 v3, v1"),          # the older load has returned
               ("20", "v_mov_b32_e32 v4, v2")]          # the younger one has not
    assert len(chk.check_kernel("k", partial)) == 1
    # a load still in flight at the back-edge of a loop meets the instruction at the loop head (address 0x10: the
    # branch at 0x20 jumps back by four instructions); with a wait in front of the back-edge it does not
    loop = [("10", "v_mov_b32_e32 v5, v1"),
            ("14", "v_add_u32_e32 v7, v7, v8"),
            ("18", "global_load_dword v1, v[18:19], off"),
            ("1c", "s_nop 0"),
            ("20", "s_cbranch_scc1 65531"),
            ("24", "s_waitcnt vmcnt(0)"),
            ("28", "s_endpgm")]
    assert len(chk.check_kernel("k", loop)) == 1
    loop[3] = ("1c", "s_waitcnt vmcnt(0)")
    assert chk.check_kernel("k", loop) == []
    # ... and on the taken side of a forward branch that skips the wait of the other side
    skip = [("10", "global_load_dword v1, v[18:19], off"),
            ("14", "s_cbranch_vccz 1"),                    # -> 0x1c
            ("18", "s_waitcnt vmcnt(0)"),
            ("1c", "v_mov_b32_e32 v3, v1"),
            ("20", "s_endpgm")]
    assert len(chk.check_kernel("k", skip)) == 1


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs llvm-objdump")
def test_built_library_touches_no_register_before_its_load_was_waited_for():
    """csrc/mphip_device.hpp issues the corner gathers as inline assembly and waits for them in a later statement;
    the machine code of the library that ships must not touch those registers in between."""
    from mptrac_amd import build as b
    from mptrac_amd import check_async_loads as chk
    for lib in (hip.load(build=False)._name, b.build_hip_exact()):      # both builds ship
        hazards, kernels, nloads = chk.check(lib)
        assert kernels > 50 and nloads > 1000
        assert hazards == [], (lib, hazards[:3])
