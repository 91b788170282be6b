#!/bin/bash
# round 5, call 2: the rewritten module_diff_pbl (generic + lean instantiations), C3p bench with 4 / 3 waves per SIMD
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_call2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pbl or boundary_layer" > $O/pbl.log 2>&1; echo "pbl rc=$?" >> $O/rc.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > $O/full.log 2>&1; echo "full rc=$?" >> $O/rc.txt
for v in default pbl3; do
  L=""; [ $v != default ] && L="MPHIP_LIB=$PWD/mptrac_amd/lib/libmptrac_hip_$v.so"
  env $L timeout 600 python bench.py --workload C3p --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3p_$v.json 2> $O/bench_c3p_$v.err; echo "c3p $v rc=$?" >> $O/rc.txt
done
timeout 600 python bench.py --workload C3p --steps 20 --warmup 5 --no-cpu-baseline --option generic_kernel=1 > $O/bench_c3p_generic.json 2> $O/bench_c3p_generic.err; echo "c3p generic rc=$?" >> $O/rc.txt
timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/fuzz.log 2>&1; echo "fuzz rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pbl.log; tail -3 $O/full.log; tail -3 $O/fuzz.log
