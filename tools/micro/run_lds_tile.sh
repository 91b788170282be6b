#!/bin/bash
# tools/micro/lds_tile.hip on the GPU box: kernel times at C3's density (fresh and aged order) and at ten times
# that density, and the TA / TD / LDS counters of both kernels.  Output: gpurun_out/lds_tile/report.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/lds_tile
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o /tmp/lds_tile "$ROOT/tools/micro/lds_tile.hip" || exit 1
{
  for args in "1e7 0" "1e7 1" "1e7 3" "3e7 0" "1e8 0"; do
    timeout 600 /tmp/lds_tile $args
  done
} > "$OUT/report.txt" 2>&1
pmc() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc -- /tmp/lds_tile 1e7 0 > "$OUT/pmc_$name.log" 2>&1
}
pmc td TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
pmc sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
pmc grbm GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
python3 - "$OUT" >> "$OUT/report.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "gather_kernel" if "gather_kernel" in r["Kernel_Name"] else ("tile_kernel" if "tile_kernel" in r["Kernel_Name"] else None)
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\ncounters per launch, N = 1e7 fresh order (rocprofv3 --pmc, mean over the launches)")
for k in ("gather_kernel", "tile_kernel"):
    print(" ", k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-30s %.4g" % (c, sum(v) / len(v)))
PY
cat "$OUT/report.txt"
