#!/bin/bash
# tools/micro/lds_tile_multistep.hip on the GPU box -> gpurun_out/lds_tile_ms/report.txt (x1 under the multi-step launch)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/lds_tile_ms
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o /tmp/lds_tile_ms "$ROOT/tools/micro/lds_tile_multistep.hip" || exit 1
{
  for args in "1e7 20 0.1 1" "1e7 20 0.1 2" "1e7 20 0.03 1" "1e7 1 0.1 1" "1e7 60 0.1 2" "1e8 20 0.1 1"; do
    timeout 300 /tmp/lds_tile_ms $args
  done
} > "$OUT/report.txt" 2>&1
pmc() {
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc -- /tmp/lds_tile_ms 1e7 20 0.1 1 > "$OUT/pmc_$name.log" 2>&1
}
pmc sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT
pmc grbm GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
python3 - "$OUT" >> "$OUT/report.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\ncounters per launch, N = 1e7, 20 steps, drift 0.1, halo 1 (rocprofv3 --pmc, mean over the launches)")
for k in sorted(acc):
    print(" ", k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-30s %.4g" % (c, sum(v) / len(v)))
PY
cat "$OUT/report.txt"
