#!/bin/bash
# Kernel time against iterations per launch (with and without recording) for cfg2 and cfg5: the fixed cost of a
# launch (table staging, state load / store, first recorded state).  Round 2: cfg2 0.5 ms of 31.2 (0.2 without
# recording), cfg5 0.05 ms of 3.1.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for w in gauss_rw changepoint_rw; do
for i in 1 10 100 1000; do
python3 bench.py --workload $w --iters $i --steps 5 --warmup 2 --no-cpu-baseline --no-ess 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$w iters %5d kernel %.3f ms  wall %s' % ($i, d['roofline']['kernel_ms'], d['roofline']['step_wall_ms']))"
python3 bench.py --workload $w --iters $i --steps 5 --warmup 2 --no-cpu-baseline --no-ess --no-record 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('$w iters %5d kernel %.3f ms  (no record)' % ($i, d['roofline']['kernel_ms']))"
done; done
