#!/bin/bash
# HBM traffic of EVERY kernel of a bench workload (GPU box): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (counters only together with --kernel-trace), summed per time step by tools/traffic_all.py.
#   tools/profile_traffic.sh <tag> <workload> [more workloads]     -> gpurun_out/traffic_<tag>_<workload>/{fetch,write}
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  OUT=$ROOT/gpurun_out/traffic_${TAG}_$W
  mkdir -p $OUT
  BENCH="python $ROOT/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-multi-gpu-probe --device-warmup-ms 0"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/fetch -o pmc -- $BENCH > $OUT/fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/write -o pmc -- $BENCH > $OUT/write.log 2>&1
  python $ROOT/tools/traffic_all.py $OUT $W > $OUT/summary.txt 2>&1
  tail -4 $OUT/summary.txt
done
