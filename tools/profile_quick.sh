#!/bin/bash
# Quick look at what bounds the step kernel: per-module-set times, then VALU / TD / TA busy of the bench.
# Every profiler pass runs under `timeout`.   Usage: tools/profile_quick.sh <tag>
set -u
TAG=${1:-quick}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 120 python $ROOT/tools/gpu_ablate.py
BENCH="python $ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --device-warmup-ms 0"
pmc() {
  local name=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/$name" -o pmc --kernel-include-regex "step_kernel" -- $BENCH > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
pmc q1 TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
pmc q2 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/q*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: max(v) for k, v in acc.items()}
for k in sorted(m):
    print("%-24s %.5g" % (k, m[k]))
if "GRBM_GUI_ACTIVE" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    print("kernel cycles %.4g  TD busy %.2f  TA busy %.2f" % (cyc, m["TD_TD_BUSY_sum"] / 256 / cyc, m["TA_TA_BUSY_sum"] / 256 / cyc))
    if "SQ_ACTIVE_INST_VALU" in m:
        # (the 6-step launch of mphip_run_timesteps: per time step)
    print("VALU busy %.2f   VALU per 64 particle-steps %.0f  VMEM_RD %.1f  SALU %.0f" % (
            m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), m["SQ_INSTS_VALU"] / 156250 / 6, m["SQ_INSTS_VMEM_RD"] / 156250 / 6, m["SQ_INSTS_SALU"] / 156250 / 6))
PY
