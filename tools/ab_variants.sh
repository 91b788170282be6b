#!/bin/bash
# A/B timing of experimental library builds (mptrac_amd/lib/libmptrac_hip_<tag>.so): bench C3 kernel time.
# Usage: tools/ab_variants.sh tag1 tag2 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for tag in "$@"; do
  for rep in 1 2; do
    MPHIP_LIB=$R/mptrac_amd/lib/libmptrac_hip_$tag.so python $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$tag', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'ms_per_step', round(d['ms_per_step'],4))"
  done
done
