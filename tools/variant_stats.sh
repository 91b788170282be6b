#!/bin/bash
# Per-kernel times of one bench workload for experimental library builds
# (mptrac_amd/lib/libmptrac_hip_<tag>.so).  Usage: tools/variant_stats.sh <workload> <pattern> tag1 tag2 ...
WL=$1; PAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  OUT=$R/gpurun_out/vs_$tag
  mkdir -p "$OUT"
  MPHIP_LIB=$R/mptrac_amd/lib/libmptrac_hip_$tag.so timeout 200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT" -o s -- \
    python $R/bench.py --no-cpu-baseline --workload $WL --steps 8 --warmup 2 --device-warmup-ms 0 > "$OUT/run.log" 2>&1
  echo "== $tag: $(grep '^{' "$OUT/run.log" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step', round(d['ms_per_step'],4))")"
  python - "$OUT" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1] + "/s_kernel_stats.csv")):
    if re.search(sys.argv[2], r["Name"]):
        print("   %-58s calls %5s  avg %9.1f us" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
