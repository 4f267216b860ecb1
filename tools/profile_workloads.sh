#!/bin/bash
# Bench line + rocprofv3 kernel-trace summary for the given bench.py workloads (on a GPU box):
#   bash tools/profile_workloads.sh logreg_hmc pt_gauss
# Output under gpurun_out/prof/; copy what is to be kept into profiles/ (see profiles/README.md).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out/prof"
for w in "$@"; do
  cd "$R"
  timeout -k 10 400 python3 bench.py --workload "$w" > "gpurun_out/prof/${w}_bench.json" 2> "gpurun_out/prof/${w}_bench.err"
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof/$w" -- \
    python3 "$R/bench.py" --workload "$w" --steps 5 --warmup 1 --no-ess \
    > "$R/gpurun_out/prof/${w}_under.json" 2> "$R/gpurun_out/prof/${w}_under.err"
  echo "done $w"
done
