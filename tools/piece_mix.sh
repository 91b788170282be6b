#!/bin/bash
# Dynamic instruction classes of the lean building blocks of the step kernel (piece_kernel<16..26>): which blocks the
# "other" VALU instructions (moves, selects, compares, lane reads ...) belong to.
#   tools/piece_mix.sh <tag>   -> gpurun_out/prof_<tag>/piece_mix.txt
set -u
TAG=${1:-piecemix}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/$name" -o pmc --kernel-include-regex "piece_kernel" -- python $ROOT/tools/piece_cost.py > "$OUT/$name.log" 2>&1
}
pass a SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
pass b SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT
pass c SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS
python - "$OUT" "$ROOT" <<'PY' | tee "$OUT/piece_mix.txt"
import csv, glob, sys, collections, re
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root + "/tools")
from piece_cost import PIECES
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = int(re.search(r"piece_kernel<(\d+)>", r["Kernel_Name"]).group(1))
        acc[k][r["Counter_Name"].replace("SQ_INSTS_", "")].append(float(r["Counter_Value"]))
cols = ["VALU", "VALU_FMA_F64", "VALU_ADD_F64", "VALU_MUL_F64", "VALU_TRANS_F64", "VALU_CVT", "VALU_INT32", "VALU_INT64", "VALU_ADD_F32",
        "VALU_MUL_F32", "VALU_FMA_F32", "VALU_TRANS_F32"]
base = {c: min(acc[0][c]) / 156250.0 for c in cols}
print("dynamic instructions per 64-particle batch, empty piece subtracted; other = VALU - the classes the counters know")
print("%-40s " % "piece" + " ".join("%8s" % c.replace("VALU_", "")[:8] for c in cols) + "    other")
for k in sorted(acc):
    row = {c: min(acc[k][c]) / 156250.0 - (base[c] if k else 0) for c in cols}
    other = row["VALU"] - sum(row[c] for c in cols[1:])
    print("%-40s " % PIECES.get(k, str(k))[:40] + " ".join("%8.0f" % row[c] for c in cols) + " %8.0f" % other)
PY
