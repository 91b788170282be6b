#!/bin/bash
# Instruction counters of every piece_kernel<K> (tools/piece_cost.py) -> gpurun_out/prof_<tag>/piece_cost.txt
set -u
TAG=${1:-pieces}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_BRANCH -f csv -d "$OUT/pmc" -o pmc --kernel-include-regex "piece_kernel" -- python $ROOT/tools/piece_cost.py > "$OUT/run.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o st --kernel-include-regex "piece_kernel" -- python $ROOT/tools/piece_cost.py > "$OUT/run_stats.log" 2>&1
python - "$OUT" "$ROOT" <<'PY' | tee "$OUT/piece_cost.txt"
import csv, glob, sys, collections, re
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root + "/tools")
from piece_cost import PIECES
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = int(re.search(r"piece_kernel<(\d+)>", r["Kernel_Name"]).group(1))
        acc[k][r["Counter_Name"].replace("SQ_", "")].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"piece_kernel<(\d+)>", r["Kernel_Name"])
        if m:
            dur[int(m.group(1))].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
cols = ["INSTS_VALU", "INSTS_SALU", "INSTS_BRANCH", "INSTS_LDS", "INSTS_VMEM_RD", "ACTIVE_INST_VALU", "WAVE_CYCLES", "WAIT_ANY"]
base = {c: min(acc[0][c]) / 156250.0 for c in cols} if 0 in acc else {c: 0 for c in cols}
print("per 64-particle batch, empty piece subtracted (10^7 particles, C3 grid, cell-sorted)")
print("%-44s %8s " % ("piece", "ms") + " ".join("%9s" % c.replace("INSTS_", "").replace("ACTIVE_INST_", "ACT_")[:9] for c in cols))
for k in sorted(acc):
    row = {c: min(acc[k][c]) / 156250.0 - (base[c] if k else 0) for c in cols}
    print("%-44s %8.3f " % (PIECES.get(k, str(k))[:44], min(dur[k]) if dur[k] else float("nan")) + " ".join("%9.0f" % row[c] for c in cols))
PY
