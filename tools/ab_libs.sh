#!/bin/bash
# A/B of several builds of the back end on ONE box (timings differ by several per cent from box to box):
#   tools/ab_libs.sh "<workloads>" <rounds> lib1.so lib2.so ...     ("default" = the library of the tree)
# prints ms per step and step-kernel ms per step of bench.py for every library, alternating between them.
WL=${1:-C3}; ROUNDS=${2:-2}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset MPHIP_LIB; else export MPHIP_LIB=$ROOT/$lib; fi
    for w in $WL; do
      python bench.py --steps 20 --warmup 5 --workload $w --no-cpu-baseline --no-multi-gpu-probe 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %-4s %.4f ms/step  kernel %.4f  one launch per step %.4f' % ('$lib', '$w', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('kernel_ms_one_launch_per_step') or 0))"
    done
  done
done
