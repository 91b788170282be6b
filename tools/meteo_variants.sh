#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 2 3 4; do
  export MPHIP_LIB=$R/mptrac_amd/lib/libmptrac_hip_mw$w.so
  rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/mv_$w -o s -- python $R/bench.py --workload C3m --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/mv_$w.log 2>&1
  echo "variant $w: $(grep -h meteo_kernel $R/gpurun_out/mv_$w/*kernel_stats.csv $R/gpurun_out/mv_$w/*/*kernel_stats.csv 2>/dev/null | head -1 | cut -d, -f1-4)"
done
