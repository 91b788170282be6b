#!/bin/bash
# dynamic instruction counts per module set (tools/gpu_ablate.py under rocprofv3 --pmc)
set -u
TAG=${1:-abl}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -f csv -d "$OUT/pmc" -o pmc --kernel-include-regex "step_kernel" -- python $ROOT/tools/gpu_ablate.py > "$OUT/run.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH -f csv -d "$OUT/pmc2" -o pmc --kernel-include-regex "step_kernel" -- python $ROOT/tools/gpu_ablate.py > "$OUT/run2.log" 2>&1
python - <<PY
import csv, glob, collections
for d in ("pmc", "pmc2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("<")[1].split(">")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc, key=lambda x: int(x.rstrip("u"))):
            row = {c: sorted(v)[len(v) // 2] for c, v in acc[k].items()}      # median dispatch
            print("mask", k, " ".join("%s=%.0f" % (c.replace("SQ_", ""), v / 156250.0) for c, v in sorted(row.items())), "(per 64-particle batch)")
PY
