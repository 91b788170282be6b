#!/bin/bash
# deposition beside module_mixing: parity (every case with mixing + deposition), C5 A/B in one call
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_call5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dist_gpu.py tests/test_bench_contract.py -x -q -m gpu -k "full or mixing or depo or bench or sort_ahead or two_ranks or bound" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/rc.txt
B="--workload C5 --steps 20 --warmup 5 --no-cpu-baseline --no-multi-gpu-probe"
for rep in 1 2; do
  timeout 150 python bench.py $B > $O/c5_split_$rep.json 2>$O/c5_split_$rep.err; echo "split $rep rc=$?" >> $O/rc.txt
  timeout 150 python bench.py $B --option depo_beside_mixing=0 > $O/c5_whole_$rep.json 2>$O/c5_whole_$rep.err; echo "whole $rep rc=$?" >> $O/rc.txt
done
cat $O/rc.txt; tail -3 $O/parity.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_call5/c5_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], "ms/step %.3f" % d["ms_per_step"], "value %.3e" % d["value"], "launches/step", r["step_kernel_launches_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && timeout 200 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_c5trace2 -o t -- python $GRAFT_REPO_ROOT/bench.py $B > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/step_timeline.py gpurun_out/r05_c5trace2 2 2>&1 | tail -40
