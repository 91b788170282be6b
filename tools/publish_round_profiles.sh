#!/bin/bash
# Copies the summaries of a tools/collect_round_profiles.sh run from gpurun_out/ (scratch) into profiles/ (tracked)
# and stamps profiles/pmc_traffic.json with the hash of the kernel sources.  Usage: tools/publish_round_profiles.sh r03
set -eu
T=${1:-r03}
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out
P=$R/profiles
cp $G/prof_$T/summary.txt $P/${T}_c3_summary.txt
cp $(ls $G/prof_$T/stats/*kernel_stats.csv $G/prof_$T/stats/*/*kernel_stats.csv 2>/dev/null | head -1) $P/${T}_c3_kernel_stats.csv
cp $G/prof_${T}mix/valu_mix.txt $P/${T}_c3_instruction_mix.txt
cp $G/prof_${T}mem/summary.txt $P/${T}_c3_mempipe_summary.txt
for w in C5 C3z C3m C2; do
  cp $(ls $G/prof_${T}_$w/*kernel_stats.csv $G/prof_${T}_$w/*/*kernel_stats.csv 2>/dev/null | head -1) $P/${T}_$(echo $w | tr A-Z a-z)_kernel_stats.csv
done
for w in C1 C2 C3 C3_1e5 C3_1e6 C3_1e8 C3_driver_args C3_one_launch_per_step C3m C3x C3z C3p C5; do
  [ -s $G/${T}_bench_$w.json ] && tail -1 $G/${T}_bench_$w.json > $P/${T}_bench_$(echo $w | tr A-Z a-z).json
done
[ -s $G/${T}_config_matrix.txt ] && cp $G/${T}_config_matrix.txt $P/${T}_config_matrix.txt
[ -s $G/${T}_config_matrix_general.txt ] && { echo; echo "# the same control sets through the general instantiation (option generic_kernel 1)"; cat $G/${T}_config_matrix_general.txt; } >> $P/${T}_config_matrix.txt
[ -s $G/${T}_sustained_480_steps.txt ] && cp $G/${T}_sustained_480_steps.txt $P/${T}_sustained_480_steps.txt
[ -s $G/${T}_config_matrix_big.txt ] && { echo; echo "# the same control sets through the instantiations with 64-bit byte offsets (option big_grid 1: what a grid beyond 4 GB of wind records takes)"; cat $G/${T}_config_matrix_big.txt; } >> $P/${T}_config_matrix.txt
[ -s $G/${T}_ml_subsets.txt ] && cp $G/${T}_ml_subsets.txt $P/${T}_model_level_subsets.txt
[ -s $G/${T}_sparse_schedule.txt ] && cp $G/${T}_sparse_schedule.txt $P/${T}_sparse_schedule.txt
[ -s $G/${T}_ml_counters.txt ] && cp $G/${T}_ml_counters.txt $P/${T}_c3z_sq_counters.txt
[ -s $G/prof_${T}pieces/piece_cost.txt ] && cp $G/prof_${T}pieces/piece_cost.txt $P/${T}_piece_costs.txt
python $R/tools/update_pmc_traffic.py $G/prof_$T C3 profiles/${T}_c3_summary.txt > /dev/null
[ -s $G/${T}_pbl_cost.txt ] && cp $G/${T}_pbl_cost.txt $P/${T}_pbl_cost.txt
# the other workloads: HBM bytes per time step over ALL their kernels (tools/profile_traffic.sh, tools/traffic_all.py)
python - <<PY
import json, os
f = "$P/pmc_traffic.json"
d = json.load(open(f))
how = {}
for w in ("C5", "C3z", "C2", "C3m", "C3p", "C3x"):
    t = "$G/traffic_${T}_%s/traffic.json" % w
    s = "$G/traffic_${T}_%s/summary.txt" % w
    if os.path.exists(t):
        d[w] = json.load(open(t))["traffic_bytes_per_step"]
        how[w] = "profiles/${T}_traffic_%s.txt" % w.lower()
        if os.path.exists(s):
            open("$P/${T}_traffic_%s.txt" % w.lower(), "w").write(open(s).read())
d["_how_other_workloads"] = ("FETCH_SIZE x the same calibration factor + WRITE_SIZE summed over EVERY kernel of the timed "
                             "region of bench.py --steps 10 (step kernels, module_sort, module_mixing, deposition, gridded "
                             "output), per time step (tools/profile_traffic.sh, tools/traffic_all.py): " + json.dumps(how))
json.dump(d, open(f, "w"), indent=1)
PY
python $R/tools/alu_model.py $P/${T}_c3_instruction_mix.txt C3 > /dev/null
python - <<PY
import json
d = json.load(open("$P/pmc_traffic.json"))
print("pmc_traffic.json:", d["C3"], d["_build_id"], d.get("_valu_busy_frac", {}).get("C3"), d.get("_fp64_valu_frac", {}).get("C3"))
PY
