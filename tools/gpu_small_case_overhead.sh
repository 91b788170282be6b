#!/bin/bash
# Where a step of the small workloads goes (C1: 10^4 particles, C2: 10^6): per-step wall time against the
# kernel's own duration, with and without kernel arguments in device memory, and the idle time the device
# shows between consecutive step kernels.  Run on the GPU box: bash tools/gpu_small_case_overhead.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in C1 C2; do
  for env in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
    echo "== $w $env"
    env $env python bench.py --workload $w --steps 2000 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f kernel_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
  done
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/small_$w -o s -- \
     python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 500 --warmup 20 --no-cpu-baseline > /dev/null 2>&1)
  python tools/trace_gaps.py gpurun_out/small_$w
done
