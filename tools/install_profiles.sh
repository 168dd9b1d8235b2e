#!/bin/bash
# Copies what a tools/gpu_round.sh session left under gpurun_out/<tag>/ into profiles/ as <tag>_*, drops the files of the
# tag it replaces, and regenerates what bench.py reads (pmc_traffic.json, the static instruction mixes).
# usage: tools/install_profiles.sh <tag> [<old tag>]
set -eu
TAG=$1; OLD=${2:-}
R=$(cd $(dirname $0)/.. && pwd); cd $R
S=gpurun_out/$TAG
[ -n "$OLD" ] && git rm -q --ignore-unmatch profiles/${OLD}_* || true
for f in $S/bench_*.json; do cp $f profiles/${TAG}_$(basename $f); done
cp $(find $S/kt -name '*kernel_stats.csv' | head -1) profiles/${TAG}_chain_kernel_stats.csv
[ -d $S/kt_fork ] && cp $(find $S/kt_fork -name '*kernel_stats.csv' | head -1) profiles/${TAG}_chain_kernel_stats_forked.csv
[ -f $S/fork_overlap.json ] && cp $S/fork_overlap.json profiles/${TAG}_chain_fork_overlap.json
cp $S/pmc_summary.json profiles/${TAG}_chain_pmc_summary.json
cp $S/trace_gaps.json profiles/${TAG}_chain_trace_gaps.json
cp $S/kernel_bounds.json profiles/${TAG}_kernel_bounds.json
cp $S/kernel_rooflines.json profiles/${TAG}_kernel_rooflines.json
[ -s $S/kernel_rooflines_2160p.json ] && cp $S/kernel_rooflines_2160p.json profiles/${TAG}_kernel_rooflines_2160p.json
if [ -d $S/chain10 ]; then
  cp $S/chain10/kernel_stats.csv profiles/${TAG}_chain_10bit_kernel_stats.csv
  cp $S/chain10/pmc_summary.json profiles/${TAG}_chain_10bit_pmc_summary.json
  cp $S/chain10/kernel_bounds.json profiles/${TAG}_chain_10bit_kernel_bounds.json
fi
[ -f $S/pytest.log ] && cp $S/pytest.log profiles/${TAG}_pytest.log
python tools/pmc_to_traffic.py $S/pmc_summary.json "profiles/${TAG}_chain_pmc_summary.json (tools/gpu_round.sh $TAG)" profiles/pmc_traffic.json
RND=$(echo $TAG | sed -E 's/^(r[0-9]+).*/\1/')              # r5b -> r5: the static mixes are per round, not per session
for K in eedi2 alias nlmeans; do
  python tools/isa_mix.py handbrake_amd/csrc/$K.hip profiles/${RND}_${K}_isa_mix.json > /dev/null
done
git add profiles
ls profiles | grep "^${TAG}_" | wc -l
