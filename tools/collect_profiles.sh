#!/bin/bash
# Everything profiles/ keeps for one round, in one GPU session:
#   bash tools/collect_profiles.sh r02
# -> gpurun_out/<tag>/: the default bench line, per-workload bench lines, rocprofv3 kernel stats
#    (tools/profile_workloads.sh), SQ counters (tools/sq_counters.sh) and HBM traffic
#    (tools/pmc_traffic.sh).  tools/gather_profiles.py then copies the summaries into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$R"
echo "== default bench"; timeout -k 10 500 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "== kernel stats"; bash tools/profile_workloads.sh logreg_mala logreg_hmc gauss_rw changepoint_rw pt_gauss > "$OUT/profile_workloads.log" 2>&1
for w in logreg_mala:10 logreg_hmc:2 gauss_rw:200 changepoint_rw:500; do
  n=${w%%:*}; i=${w##*:}
  echo "== sq counters $n"; bash tools/sq_counters.sh $n $i > "$OUT/sq_$n.json" 2> "$OUT/sq_$n.err"
  echo "== pmc traffic $n"; bash tools/pmc_traffic.sh $n $i > "$OUT/pmc_$n.txt" 2> "$OUT/pmc_$n.err"
done
echo "== done"
