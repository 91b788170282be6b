"""Build recipe of the in-tree native libraries (gfx950 only).

  lib/libmptrac_hip.so   the HIP back end + C ABI (csrc/mphip_api.hip)

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

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics"]


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
    src = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))]
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
    return out


def build_hip(force=False, verbose=False, extra_flags=()):
    if os.environ.get("MPHIP_LIB"):
        return os.environ["MPHIP_LIB"]
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _stale(HIP_LIB, hip_sources()):
        if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
            if os.path.exists(HIP_LIB):
                return HIP_LIB       # prebuilt library shipped with the tree
        cmd = [_hipcc(), *HIPCC_FLAGS, *extra_flags, "-o", HIP_LIB, os.path.join(CSRC, "mphip_api.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
