#!/usr/bin/env python3
"""What a user of the reference's interface gets at BASELINE configs[2]'s size: the drop-in `trac` binary
(mptrac_amd/host/trac.c, the loop of the reference's src/trac.c:131-163) built with production extents
(-DNP=10000000 -DEX=724 -DEY=364 -DEP=140), on the C3 synthetic meteorology written as MET_TYPE 1 files, 10^7 particles,
three meteo intervals of one hour (61 calls of mptrac_run_timestep, three mptrac_get_met hand-overs read from disk), one
gridded and one particle output per interval.  Prints the driver's own TIMER_TIMESTEPS line -- particle-steps/s over
the whole loop, hand-overs and outputs included -- next to the rate without outputs and the per-kernel times of a
rocprofv3 trace of the same command.  GPU box; files go to /tmp (8 GB), the report to gpurun_out/trac_dropin/.

  tools/gpu_trac_dropin.py [--particles 1e7] [--hours 3] [--no-trace]
"""
import argparse
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import hostfiles as hf  # noqa: E402
import bench  # noqa: E402
from mptrac_amd import build  # noqa: E402
from mptrac_amd.synth import synthetic_met, synthetic_particles  # noqa: E402

T0 = 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--hours", type=int, default=3)
    ap.add_argument("--no-trace", action="store_true")
    args = ap.parse_args()
    n = int(args.particles)
    out_root = os.path.join(ROOT, "gpurun_out", "trac_dropin")
    os.makedirs(out_root, exist_ok=True)
    _, trac = build.build_host(dims={"NP": max(n, 1000), "EX": 724, "EY": 364, "EP": 140}, outdir=os.path.join(build.LIBDIR, "big"))
    tmp = tempfile.mkdtemp(prefix="trac_dropin_", dir="/tmp")
    grid, _, ctl, quantities, fields = bench.WORKLOADS["C3"]
    t_gen = time.time()
    metbase = os.path.join(tmp, "met")
    for k in range(args.hours + 1):
        m = synthetic_met(grid, T0 + 3600.0 * k, 1.0 + 0.25 * k, fields=fields)
        hf.write_met_bin(hf.met_filename(metbase, m.time), m)
    atm = synthetic_particles(n, time=T0, quantities=quantities)
    hf.write_atm_bin(os.path.join(tmp, "atm_in"), atm)
    t_gen = time.time() - t_gen
    keys = {"NQ": len(quantities), "METBASE": metbase, "MET_TYPE": 1, "DT_MET": 3600, "DT_MOD": 180, "ADVECT": 4, "DIFFUSION": 1,
            "CONV_CAPE": 0, "RNG_TYPE": 1, "MET_DT_OUT": 0, "T_STOP": T0 + 3600.0 * args.hours, "ATM_TYPE": 1, "ATM_TYPE_OUT": 1,
            "ATM_BASENAME": "atm", "ATM_DT_OUT": 3600, "GRID_BASENAME": "grid", "GRID_DT_OUT": 3600, "GRID_TYPE": 0,
            "GRID_NX": 360, "GRID_NY": 180, "GRID_NZ": 1}
    keys.update({"QNT_NAME[%d]" % i: q for i, q in enumerate(quantities)})
    report = [f"tools/gpu_trac_dropin.py: {n} particles, {args.hours} h ({20 * args.hours + 1} calls of mptrac_run_timestep), "
              f"C3 meteorology 721 x 361 x 137 as MET_TYPE 1 files (inputs generated in {t_gen:.0f} s)"]
    env = dict(os.environ, TMPDIR="/tmp")
    for name, extra in (("outputs every hour (1 grid + 1 particle file per interval)", {}),
                        ("no outputs", {"ATM_BASENAME": "-", "GRID_BASENAME": "-"}),
                        ("no outputs, next meteo file read beside the time steps (HIP_MET_PREFETCH 1)",
                         {"ATM_BASENAME": "-", "GRID_BASENAME": "-", "HIP_MET_PREFETCH": 1})):
        hf.write_ctl(os.path.join(tmp, "trac.ctl"), dict(keys, **extra))
        open(os.path.join(tmp, "dirlist"), "w").write(tmp + "\n")
        for rep in range(2):      # (the second run reads the meteo files from the page cache)
            t0 = time.time()
            r = subprocess.run([trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], cwd=tmp, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            wall = time.time() - t0
            text = r.stdout.decode()
            assert r.returncode == 0, text[-3000:]
            line = [ln for ln in text.splitlines() if "TIMER_TIMESTEPS" in ln][-1].strip()
            report.append(f"{name}, run {rep + 1}: {line}   (process wall {wall:.1f} s)")
            print(report[-1], flush=True)
            if rep == 1:      # where the loop's wall time went, by call (trac.c)
                for ln in text.splitlines():
                    if any(k in ln for k in ("TIMER_GET_MET", "TIMER_RUN_TIMESTEP", "TIMER_WRITE_OUTPUT", "TIMER_UPDATE_HOST",
                                             "TIMER_WITHOUT_MET_AND_OUTPUT", "MEMORY_METEO", "MEMORY_ATM")):
                        report.append("        " + ln.strip())
    if not args.no_trace:
        hf.write_ctl(os.path.join(tmp, "trac.ctl"), keys)
        prof = os.path.join(out_root, "trace")
        r = subprocess.run(["timeout", "900", "rocprofv3", "--kernel-trace", "--memory-copy-trace", "--stats", "-f", "csv", "-d", prof, "-o", "t", "--",
                            trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], cwd=tmp, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        text = r.stdout.decode()
        line = [ln for ln in text.splitlines() if "TIMER_TIMESTEPS" in ln]
        report.append("under rocprofv3 --kernel-trace --memory-copy-trace --stats: " + (line[-1].strip() if line else f"rc {r.returncode}"))
        for stats in glob.glob(os.path.join(prof, "**", "*_stats.csv"), recursive=True):
            rows = list(csv.DictReader(open(stats)))
            if not rows or "Name" not in rows[0]:
                continue
            report.append("  " + os.path.basename(stats))
            for row in sorted(rows, key=lambda x: -float(x.get("TotalDurationNs", 0) or 0))[:12]:
                name = re.sub(r"\(.*", "", row["Name"])[:60]
                report.append("    %-60s calls %6s  total %9.2f ms  avg %9.3f ms" % (name, row.get("Calls"), float(row["TotalDurationNs"]) / 1e6,
                                                                                   float(row["AverageNs"]) / 1e6))
    open(os.path.join(out_root, "report.txt"), "w").write("\n".join(report) + "\n")
    print("\n".join(report[-40:]))
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
