#!/bin/bash
# Per-kernel times of a bench run (rocprofv3 --kernel-trace --stats only).  Usage: tools/profile_stats.sh <tag> [bench args]
set -u
TAG=${1:-st}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT" -o s -- python $ROOT/bench.py --no-cpu-baseline --steps 40 --warmup 3 "$@" > "$OUT/run.log" 2>&1
grep "^{" "$OUT/run.log" | cut -c1-170
python - "$OUT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + "/s_kernel_stats.csv")):
    print("%-60s calls %5s  avg %9.1f us  total %6.2f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
