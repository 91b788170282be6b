#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box):
#   1. kernel trace + stats                      -> per-kernel time
#   2..n. PMC passes (separate runs, counters only with --kernel-trace)
# Usage: tools/profile.sh <tag> [bench args...]     outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --device-warmup-ms 0 $*"   # (counters per launch: no scratch launches)
# the kernel-stats pass is the DEFAULT bench command (the driver's), device warm-up included: its average
# step_kernel<255u> duration is the one bench.py's roofline.kernel_ms has to agree with
BENCH_STATS="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline $*"

timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o stats -- $BENCH_STATS > "$OUT/stats.log" 2>&1
grep "^{" "$OUT/stats.log" | tail -1 > "$OUT/bench_under_profiler.json"

pmc() {  # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc --kernel-include-regex "step_kernel" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
# calibration of FETCH_SIZE / WRITE_SIZE on a kernel with a known byte count:
# pack_kernel reads 32 B and writes 32 B per grid point, fully coalesced
calib() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc --kernel-include-regex "mphip::pack_kernel" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
}
calib calib_fetch FETCH_SIZE
calib calib_write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
ls -R "$OUT" | head -60
