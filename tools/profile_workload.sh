#!/bin/bash
# Kernel time and the basic SQ / TA / TD counters of the step kernel for one bench workload.
# Usage: tools/profile_workload.sh <tag> <workload> [kernel regex, default step_kernel] [further bench.py arguments]
set -u
TAG=${1:-wl}; WL=${2:-C3}; KRE=${3:-step_kernel}; MORE=${4:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload $WL --steps 6 --warmup 1 --no-cpu-baseline --device-warmup-ms 0 $MORE"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o s -- $BENCH > "$OUT/stats.log" 2>&1
pmc() {
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/$name" -o pmc --kernel-include-regex "$KRE" -- $BENCH > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
pmc q1 TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
pmc q2 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for r in list(csv.DictReader(open(out + "/stats/s_kernel_stats.csv")))[:6]:
    print("%-60s calls %5s  avg %9.1f us  total %6.2f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
acc = collections.defaultdict(list)
for f in glob.glob(out + "/q*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: max(v) for k, v in acc.items()}
if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    print("largest dispatch of the selected kernel: cycles %.4g  TD busy %.2f  TA busy %.2f  VALU busy %.2f  VALU/batch %.0f  VMEM_RD/batch %.1f  SALU/batch %.0f" % (
        cyc, m["TD_TD_BUSY_sum"] / 256 / cyc, m["TA_TA_BUSY_sum"] / 256 / cyc, m["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc),
        m["SQ_INSTS_VALU"] / 156250, m["SQ_INSTS_VMEM_RD"] / 156250, m["SQ_INSTS_SALU"] / 156250))
PY
