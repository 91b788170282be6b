#!/bin/bash
# rocprofv3 over bench workload C3m (C3 + module_meteo every step): per-kernel stats and the HBM counters
# of meteo_kernel.  Usage: tools/profile_meteo.sh <tag>     outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01m}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload C3m --steps 10 --warmup 2 --no-cpu-baseline --device-warmup-ms 0 $*"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
grep "^{" "$OUT/stats.log" | tail -1 > "$OUT/bench_under_profiler.json"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d "$OUT/pmc_$c" -o pmc --kernel-include-regex "meteo_kernel" -- $BENCH > "$OUT/pmc_$c.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, statistics, sys
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %6s  avg %10.1f us  total %6.2f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    if vals:
        print(c, "meteo_kernel median per launch [KiB]:", statistics.median(vals), "launches", len(vals))
PY
