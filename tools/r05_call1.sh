#!/bin/bash
# round 5, first GPU call: the new full-size tests, the nml != npl tests, the zeta27 big grid, baseline bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_call1
mkdir -p $O
free -g > $O/mem.txt; nproc >> $O/mem.txt; rocm-smi --showmeminfo vram >> $O/mem.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu --durations=5 > $O/full_size.log 2>&1; echo "full_size rc=$?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "model_level_count or model_levels" > $O/nml.log 2>&1; echo "nml rc=$?" >> $O/rc.txt
timeout 900 python tools/gpu_big_grid.py zeta27 > $O/zeta27.log 2>&1; echo "zeta27 rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?" >> $O/rc.txt
timeout 600 python bench.py --workload C5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?" >> $O/rc.txt
timeout 600 python bench.py --workload C3z --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3z.json 2> $O/bench_c3z.err; echo "bench c3z rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -5 $O/full_size.log; tail -3 $O/nml.log; tail -3 $O/zeta27.log
