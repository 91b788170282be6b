#!/bin/bash
# round 5, call 3: module_diff_pbl again (tests, C3p with 4 / 3 waves, generic), the touch-load experiment
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_call3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pbl or boundary_layer" > $O/pbl.log 2>&1; echo "pbl rc=$?" >> $O/rc.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-multi-gpu-probe"
for v in default pbl3; do
  L=""; [ $v != default ] && L="MPHIP_LIB=$PWD/mptrac_amd/lib/libmptrac_hip_$v.so"
  env $L timeout 150 python bench.py --workload C3p $B > $O/bench_c3p_$v.json 2> $O/bench_c3p_$v.err; echo "c3p $v rc=$?" >> $O/rc.txt
done
timeout 150 python bench.py --workload C3p $B --option generic_kernel=1 > $O/bench_c3p_generic.json 2> $O/bench_c3p_generic.err; echo "c3p generic rc=$?" >> $O/rc.txt
for rep in 1 2; do
  for v in default touch touch1; do
    L=""; [ $v != default ] && L="MPHIP_LIB=$PWD/mptrac_amd/lib/libmptrac_hip_$v.so"
    env $L timeout 150 python bench.py $B > $O/bench_c3_${v}_$rep.json 2> $O/bench_c3_${v}_$rep.err; echo "c3 $v $rep rc=$?" >> $O/rc.txt
    env $L timeout 150 python bench.py --workload C5 $B > $O/bench_c5_${v}_$rep.json 2> $O/bench_c5_${v}_$rep.err; echo "c5 $v $rep rc=$?" >> $O/rc.txt
  done
done
MPHIP_LIB=$PWD/mptrac_amd/lib/libmptrac_hip_touch.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lean_instantiations or (run_timestep_20_steps and (conv_sedi or full))" > $O/touch_parity.log 2>&1; echo "touch parity rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pbl.log; tail -3 $O/touch_parity.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_call3/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], "ms/step %.3f" % d["ms_per_step"], "kernel %.3f" % r["step_kernel_ms_per_step"], "single %.3f" % (r["kernel_ms_one_launch_per_step"] or 0))
    except Exception as e:
        print(f, "unreadable", e)
PY
