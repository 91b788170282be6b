#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ tree (tools/profile.sh) into the text
summary that is committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1]
out = []


def find(pattern):
    return sorted(glob.glob(os.path.join(src, pattern), recursive=True))


for f in find("stats/**/*kernel_stats.csv"):
    out.append(f"== kernel stats ({os.path.relpath(f, src)}) ==")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        out.append("  {Name:60.60s} calls={Calls:>6s} total_ns={TotalDurationNs:>14s} avg_ns={AverageNs:>14s} pct={Percentage}".format(**r))
bj = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(bj):
    out.append("== bench line under the profiler ==")
    out.append("  " + open(bj).read().strip())

for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r.get("Kernel_Name", "") or "pack_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.append(f"== PMC {os.path.basename(d)} (step_kernel dispatches, per-dispatch mean) ==")
        for k, v in sorted(acc.items()):
            # rocprofv3 emits one row per dispatch (and per dimension instance); average per dispatch
            out.append(f"  {k:34s} n={len(v):4d} mean={sum(v) / len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
print("\n".join(out))

# HBM traffic of the step kernel per launch (MI355X_MICROARCH.md, HBM section):
# FETCH_SIZE / WRITE_SIZE are in KiB; calibrate on pack_kernel (known bytes).
# The timed steps of bench.py go to the device as one multi-step launch (step_kernel<6710....u>: the kMultiStep
# instantiations, K time steps per particle): its counters divided by K are the per-step figures everything below
# works with.  Runs without such a launch (--multi-step off, older builds): every dispatch is one step.
try:
    _steps = int(json.loads(open(bj).read().strip().splitlines()[-1])["steps"])
except Exception:
    _steps = 1
_pmc_steps = 10      # tools/profile.sh: the PMC passes run `bench.py --steps 10`


def one(d, name, kernel=None):
    for f in glob.glob(os.path.join(src, d, "**/*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name
                and (kernel is None or kernel in r.get("Kernel_Name", ""))]
        multi = [float(r["Counter_Value"]) / _pmc_steps for r in rows if "step_kernel<6710" in r.get("Kernel_Name", "")]
        if multi:
            return multi
        return [float(r["Counter_Value"]) for r in rows]
    return []


fs, ws = one("pmc_fetch", "FETCH_SIZE"), one("pmc_write", "WRITE_SIZE")
# (the meteo pack kernel only: the regex of the calibration passes also matches perm_pack / perm_unpack)
cf, cw = one("pmc_calib_fetch", "FETCH_SIZE", "mphip::pack_kernel("), one("pmc_calib_write", "WRITE_SIZE", "mphip::pack_kernel(")
if fs and ws:
    steady_f = sorted(fs)[len(fs) // 2] * 1024.0      # median dispatch (the dt = 0 first call is smaller)
    steady_w = sorted(ws)[len(ws) // 2] * 1024.0
    info = {"fetch_bytes_raw": steady_f, "write_bytes_raw": steady_w}
    grid_bytes = None
    try:
        cfg = json.loads(open(bj).read().strip().splitlines()[-1])
        nx, ny, npl = cfg["config"]["grid"]
        grid_bytes = nx * ny * npl * 32.0
    except Exception:
        pass
    if cf and grid_bytes:
        # FETCH_SIZE: scale by the factor that makes pack_kernel's coalesced reads come out at their known
        # byte count (~2.0 on gfx950).  WRITE_SIZE: the step kernel's own stores are a known byte count
        # (time, lon, lat, p + uvwp = 44 B per particle, coalesced) and WRITE_SIZE matches it as is, so no
        # factor is applied (pack_kernel's 24-byte record stores are not a usable calibration: partial lines).
        kf = grid_bytes / (max(cf) * 1024.0)
        info.update(calib_fetch_factor=kf, calib_known_bytes=grid_bytes)
        try:
            known_w = 44.0 * cfg["config"]["particles_per_gpu"]
            info.update(write_known_bytes=known_w, write_raw_over_known=steady_w / known_w)
        except Exception:
            pass
        info["traffic_bytes"] = steady_f * kf + steady_w
    else:
        info["traffic_bytes"] = steady_f + steady_w
    # fp64-VALU side of the roofline (SURVEY 8(d) caveat): share of the SIMD cycles in which a VALU
    # instruction is executing = SQ_ACTIVE_INST_VALU [quad-cycles, summed over waves] x 4 /
    # (1024 SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE summed over the 8 XCDs / 8
    av, ga = one("pmc_sq2", "SQ_ACTIVE_INST_VALU"), one("pmc_grbm", "GRBM_GUI_ACTIVE")
    iv = one("pmc_sq1", "SQ_INSTS_VALU")
    if av and ga:
        info["valu_busy_frac"] = max(av) * 4.0 / (1024.0 * max(ga) / 8.0)
    if iv:
        try:
            info["valu_insts_per_64_particle_steps"] = max(iv) / (cfg["config"]["particles_per_gpu"] / 64.0)
        except Exception:
            pass
    print("== HBM traffic of step_kernel per time step (multi-step launch / its steps) ==")
    print("  " + json.dumps(info))
    json.dump(info, open(os.path.join(src, "traffic.json"), "w"))
