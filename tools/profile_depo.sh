#!/bin/bash
# SQ counters of depo_kernel in workload C5 (GPU box): tools/profile_depo.sh  -> gpurun_out/prof_depo/
set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof_depo; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload C5 --steps 10 --warmup 2 --no-cpu-baseline --device-warmup-ms 0"
pmc() { local name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc --kernel-include-regex "depo_kernel" -- $BENCH > "$OUT/pmc_$name.log" 2>&1; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
pmc mem FETCH_SIZE WRITE_SIZE
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "depo_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:24s} {tot[k] / n[k]:.4g} per launch ({n[k]} launches)")
v = {k: tot[k] / n[k] for k in tot}
print("VALU busy %.3f   waiting share of wave cycles %.3f   VALU per particle %.1f" % (
    v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * v["GRBM_GUI_ACTIVE"] / 8), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_INSTS_VALU"] * 64 / 1e7))
PY
