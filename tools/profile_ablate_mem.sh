#!/bin/bash
# Memory-pipeline counters per module set (tools/gpu_ablate.py under rocprofv3 --pmc; PMC passes only)
set -u
TAG=${1:-ablmem}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pmc() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/$name" -o pmc --kernel-include-regex "step_kernel" -- python $ROOT/tools/gpu_ablate.py > "$OUT/$name.log" 2>&1
}
pmc m1 TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
pmc m2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pmc m3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum
pmc m4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum FETCH_SIZE
pmc m5 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TC_STALL_sum
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/m*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("<")[1].split(">")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc, key=lambda x: int(x.rstrip("u"))):
    row = {c: sorted(v)[len(v) // 2] for c, v in acc[k].items()}
    print("mask", k)
    for c, v in sorted(row.items()):
        print("   %-40s %.5g" % (c, v))
PY
