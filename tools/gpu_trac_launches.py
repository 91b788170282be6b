#!/usr/bin/env python3
"""Does the C driver share launches?  Runs the `trac` binary on the set-up of tests/test_host_driver.py (3 000 particles,
2 h, particle output every half hour) under `rocprofv3 --kernel-trace --stats` and counts the launches of the step kernel
against the time steps of the run, with the step queue of the host layer (default) and without it (HIP_STEP_BATCH 1).
GPU box; writes to gpurun_out/trac_launches/."""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_host_driver as T  # noqa: E402

out_root = os.path.join(ROOT, "gpurun_out", "trac_launches")
os.makedirs(out_root, exist_ok=True)
for batch in ("default", "1"):
    tmp = tempfile.mkdtemp(prefix="trac_launches_")
    trac, mets, atm = T._setup(tmp, n=3000, hours=2, extra={"ATM_DT_OUT": 1800})
    env = dict(os.environ, TMPDIR="/tmp")
    if batch != "default":
        env["HIP_STEP_BATCH"] = batch
    prof = os.path.join(out_root, "queue_" + batch)
    r = subprocess.run(["timeout", "300", "rocprofv3", "--kernel-trace", "--stats", "-f", "csv", "-d", prof, "-o", "t", "--",
                        trac, os.path.join(tmp, "dirlist"), "trac.ctl", "atm_in"], cwd="/tmp", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    stats = glob.glob(os.path.join(prof, "**", "*kernel_stats.csv"), recursive=True)[0]
    launches = sum(int(row["Calls"]) for row in csv.DictReader(open(stats)) if "step_kernel" in row["Name"])
    steps = 2 * 3600 // 180 + 1          # t_start .. t_stop in steps of DT_MOD = 180 s
    print(f"step queue {batch:>7s}: {launches:3d} launches of the step kernel for {steps} time steps", flush=True)
