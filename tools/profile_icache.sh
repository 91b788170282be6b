#!/bin/bash
# Instruction-cache behaviour of the step kernel (PMC passes only).
# Usage: tools/profile_icache.sh <tag> [bench args]
set -u
TAG=${1:-ic}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQC_[A-Z0-9_]+" | sort -u | tr "\n" " " > "$OUT/sqc_counters.txt"
BENCH="python $ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --device-warmup-ms 0 $*"
pmc() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc --kernel-include-regex "step_kernel" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
}
pmc i1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pmc i2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc i3 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM
python - "$OUT" <<'PY' | tee "$OUT/icache.txt"
import csv, glob, sys, collections
out = sys.argv[1]
print(open(out + "/sqc_counters.txt").read())
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_i*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("%-30s per launch (steady) %.5g" % (k, max(acc[k])))
PY
