#!/bin/bash
# Throughput against the number of chains per GPU (bench.py --chains): where each workload saturates the chip.
#   bash tools/chain_sweep.sh > gpurun_out/chain_sweep.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for w in logreg_mala:20 gauss_rw:500 changepoint_rw:1000; do
  n=${w%%:*}; i=${w##*:}
  for c in 1024 4096 16384 32768 65536 131072 262144; do
    if [ $n = logreg_mala ] && [ $c -gt 131072 ]; then continue; fi
    if [ $n = gauss_rw ] && [ $c -gt 131072 ]; then continue; fi   # history of 131 072 x 100 x 500 doubles = 52 GB
    python3 bench.py --workload $n --chains $c --iters $i --steps 3 --warmup 1 --no-cpu-baseline --no-ess 2>/dev/null |
      python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
ms = d['roofline']['kernel_ms']
print('%-15s chains %7d  kernel %8.2f ms per $i iterations = %.4g it/s  roofline frac %.3f' % ('$n', $c, ms, $c * $i / (ms * 1e-3), d['roofline']['frac']))"
  done
done
