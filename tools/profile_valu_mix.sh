#!/bin/bash
# Dynamic instruction mix of step_kernel (PMC passes only, one per counter group).
# Usage: tools/profile_valu_mix.sh <tag> [bench args]      output: gpurun_out/prof_<tag>/valu_mix.txt
set -u
TAG=${1:-mix}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --device-warmup-ms 0 $*"
pmc() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc --kernel-include-regex "step_kernel" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
}
pmc m1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
pmc m2 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT
pmc m3 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH
pmc m4 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_VSKIPPED
python - "$OUT" <<'PY' | tee "$OUT/valu_mix.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = max(acc[k]) / 6.0  # the 6-step launch of `bench.py --steps 6` (mphip_run_timesteps), per time step
    print("%-28s per time step %.4g   per 64 particle-steps %.1f" % (k, v, v / (1e7 / 64)))
PY
