#!/bin/bash
# Memory-pipeline and instruction-mix PMC passes over the fused step kernel.
# Usage: tools/profile_mem.sh <tag> [bench args...]
set -u
TAG=${1:-mem}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --device-warmup-ms 0 $*"
pmc() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc --kernel-include-regex "step_kernel" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "pass $name failed"
}
pmc ta1 TA_TA_BUSY_sum TA_BUSY_avr
pmc ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pmc ta3 TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pmc td TD_TD_BUSY_sum TD_TC_STALL_sum
pmc tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pmc tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pmc tcp3 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pmc sqa SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64
pmc sqb SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_SMEM SQ_INSTS_BRANCH
pmc sqc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pmc grbm GRBM_GUI_ACTIVE
