set -u
cd $GRAFT_REPO_ROOT
T=${1:-r03}
MODE=${2:-all}      # "core": the bench lines, kernel statistics, instruction mix and HBM counters only
timeout 900 bash tools/profile.sh $T > gpurun_out/${T}_profile.log 2>&1
timeout 300 python tools/summarize_prof.py gpurun_out/prof_$T > gpurun_out/prof_$T/summary.txt 2>&1
timeout 900 bash tools/profile_valu_mix.sh ${T}mix > gpurun_out/${T}_mix.log 2>&1
timeout 900 bash tools/profile_mem.sh ${T}mem > gpurun_out/${T}_mem.log 2>&1
timeout 300 python tools/summarize_prof.py gpurun_out/prof_${T}mem > gpurun_out/prof_${T}mem/summary.txt 2>&1
for w in C5 C3z C3m C2; do
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_$w -o s -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_$w.log 2>&1)
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_$w.json
done
timeout 300 python bench.py > gpurun_out/${T}_bench_C3.json 2> gpurun_out/${T}_bench_C3.err
tail -c 600 gpurun_out/${T}_bench_C3.json
timeout 300 python bench.py --particles 1e8 --steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_C3_1e8.json
timeout 300 python bench.py --workload C3x --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_C3x.json
timeout 300 python bench.py --workload C1 --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_C1.json
timeout 300 python bench.py --particles 1e5 --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_C3_1e5.json
timeout 300 python bench.py --particles 1e6 --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_C3_1e6.json
timeout 300 python bench.py --steps 20 --warmup 5 --multi-step off --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_C3_one_launch_per_step.json
[ $MODE = core ] || timeout 900 python tools/gpu_config_matrix.py > gpurun_out/${T}_config_matrix.txt 2>&1
[ $MODE = core ] || timeout 900 python tools/gpu_config_matrix.py generic_kernel=1 > gpurun_out/${T}_config_matrix_general.txt 2>&1
[ $MODE = core ] || timeout 900 python tools/gpu_bench_sweep.py C3 > gpurun_out/${T}_sustained_480_steps.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_C3_driver_args.json 2> /dev/null
[ $MODE = core ] || timeout 900 bash tools/piece_cost.sh ${T}pieces > gpurun_out/${T}_pieces.log 2>&1
timeout 600 bash tools/profile_ml.sh ${T}z > gpurun_out/${T}_ml_counters.txt 2>&1
[ $MODE = core ] || timeout 900 python tools/gpu_config_matrix.py big_grid=1 > gpurun_out/${T}_config_matrix_big.txt 2>&1
[ $MODE = core ] || timeout 300 python tools/gpu_ml_subsets.py > gpurun_out/${T}_ml_subsets.txt 2>&1
[ $MODE = core ] || timeout 300 python tools/gpu_sparse_schedule.py > gpurun_out/${T}_sparse_schedule.txt 2>&1
# HBM counters of every kernel of the workloads with several kernels per step / other step kernels (bench.py quotes them)
timeout 1800 bash tools/profile_traffic.sh $T C5 C3z C2 C3m C3p C3x > gpurun_out/${T}_traffic_all.log 2>&1
timeout 300 python bench.py --workload C3p --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_C3p.json
[ $MODE = core ] || timeout 300 python tools/gpu_pbl_cost.py > gpurun_out/${T}_pbl_cost.txt 2>&1
