# Second pass of a round's collection: the bench lines again, after tools/publish_round_profiles.sh has stamped
# profiles/pmc_traffic.json with the build the counters were taken from (bench.py quotes `traffic` only for a matching
# build, so the first pass -- which ran before the counters existed -- prints null).  The lines are the ones of
# tools/collect_round_profiles.sh.  Usage (on the GPU box): bash tools/rebench_lines.sh r05
set -u
cd $GRAFT_REPO_ROOT
T=${1:-r03}
grep -E '^ *timeout 300 python bench.py' tools/collect_round_profiles.sh | sed "s/^ *//; s/\${T}/$T/g" > /tmp/rebench.sh
for w in C5 C3z C3m C2; do grep '\$w' /tmp/rebench.sh | sed "s/\$w/$w/g"; done > /tmp/rebench_w.sh
grep -v '\$w' /tmp/rebench.sh >> /tmp/rebench_w.sh
bash /tmp/rebench_w.sh
tail -c 300 gpurun_out/${T}_bench_C3.json
