#!/bin/bash
# A/B of library builds on one box: step-kernel median of workload C3 (tools/gpu_step_trace.py), alternating.
# Usage: tools/ab_trace.sh REPS tag1 tag2 ...     ("default" = mptrac_amd/lib/libmptrac_hip.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
REPS=$1; shift
for rep in $(seq $REPS); do
  for tag in "$@"; do
    if [ "$tag" = default ]; then unset MPHIP_LIB; else export MPHIP_LIB=$R/mptrac_amd/lib/libmptrac_hip_$tag.so; fi
    echo -n "$tag: "; timeout 120 python $R/tools/gpu_step_trace.py 70 ${TRACE_ARGS:-} 2>&1 | tail -1
  done
done
