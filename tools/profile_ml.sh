#!/bin/bash
# SQ counters of the model-level step kernel (workload C3z): how busy the VALU is, what the waves wait for.
# Usage: tools/profile_ml.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r04z}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload C3z --steps 10 --warmup 2 --no-cpu-baseline --device-warmup-ms 0 $*"
pmc() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc --kernel-include-regex "step_kernel" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} {tot[k] / n[k]:.4g} per launch ({n[k]} launches)")
PY
