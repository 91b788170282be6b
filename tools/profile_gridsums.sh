#!/bin/bash
# Counters of the kernels of one gridded output of C3 (crowded-cell path): HBM bytes, TA / TD busy.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_gridsums
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --device-warmup-ms 0"
KRE="cell_sum_chains|cell_slot_pairs|box_index|sort_scatter|cell_bounds"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o s -- $BENCH > "$OUT/stats.log" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $pass -f csv -d "$OUT/$name" -o pmc --kernel-include-regex "$KRE" -- $BENCH > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
done
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-22s n=%d  last=%.5g  max=%.5g" % (c, len(v), v[-1], max(v)))
PY
