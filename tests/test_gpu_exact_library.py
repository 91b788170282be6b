"""The reference-rounding build (lib/libmptrac_hip_exact.so: the same sources compiled with the reference's roundings --
IEEE divisions instead of reciprocal products, the C library's cos / sin, no contraction; mptrac_amd/build.py:EXACT_FLAGS)
against the oracle: not within a tolerance but BIT FOR BIT -- positions, quantities and the single-precision
perturbations of the named cases after their 21 steps (10 000 particles each).  The default build trades those last bits
for speed (0.80 against 1.15 ms per step on workload C3, DESIGN.md sections 2 and 4); this one is what a run that must
reproduce the CPU build's numbers links.
A process loads one of the two libraries, so the comparison runs in a child with MPTRAC_AMD_EXACT=1
(tools/gpu_bit_census.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GROUPS = [["advect", "advect_midpoint", "advect_euler", "turb", "diff", "pbl", "pbl_meso"],
          ["conv_sedi", "conv_thresh", "full", "wet_henry", "bound", "isosurf_theta", "isosurf_rho"],
          ["advect_zeta", "advect_zeta_midpoint", "advect_eta", "zeta_full", "advect_mlp", "mlp_full", "bound_pbl_zeta"],
          ["meteo", "meteo_gated"]]


def _census(names, exact):
    env = dict(os.environ, MPTRAC_AMD_EXACT="1" if exact else "0")
    env.pop("MPHIP_LIB", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_bit_census.py"), "--json", *names],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lib = [ln for ln in res.stdout.splitlines() if ln.startswith("library:")][0]
    assert ("exact" in lib) == exact, lib
    return [json.loads(ln[5:]) for ln in res.stdout.splitlines() if ln.startswith("JSON ")]


@pytest.mark.parametrize("names", GROUPS, ids=["pressure_levels", "modules", "model_levels", "module_meteo"])
def test_reference_rounding_build_has_the_oracles_bits(names):
    rows = _census(names, exact=True)
    assert [r["case"] for r in rows] == names
    for r in rows:
        differing = {k: r[k] for k in ("lon", "lat", "p", "q", "uvwp") if r[k]}
        assert not differing, (r["case"], differing, r.get("q rows"))


def test_the_comparison_sees_the_default_builds_last_bits():
    """The same census on the default build counts values that are not the oracle's bits (a few per cent of the
    longitudes and pressures, all within 1e-15) -- the check above is not vacuous."""
    r, = _census(["advect"], exact=False)
    assert r["lon"] > 0 and r["p"] > 0 and r["uvwp"] == 0 and r["worst"] < 2e-15


def test_the_c_driver_with_the_library_in_front_writes_the_oracles_bits(tmp_path):
    """trac is linked against libmptrac_hip.so by name; started with the reference-rounding build in front of it
    (LD_PRELOAD: same ABI -- the switch a C user has) it announces that build, and the particle files of a two-hour run
    with turbulent diffusion, convection and the boundary-layer closure hold the oracle's positions bit for bit."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostfiles as hf
    import test_host_driver as D
    from mptrac_amd import build as b
    tmp = str(tmp_path)
    trac, mets, atm = D._setup(tmp, n=3000, hours=2, pbl=True)
    res = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, LD_PRELOAD=b.build_hip_exact()))
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    assert "reference rounding" in res.stdout + res.stderr
    snaps = D._oracle(mets, atm, 2, pbl=True)
    for hour in (0, 1, 2):
        got = hf.read_atm_bin(os.path.join(tmp, "atm_2022_06_02_%02d_00_00.bin" % hour), len(D.QUANT))
        ref = snaps[D.T0 + 3600.0 * hour]
        for k in ("time", "lon", "lat", "p", "q"):
            assert np.array_equal(got[k], ref[k]), (hour, k, int(np.count_nonzero(got[k] != ref[k])))
