#!/bin/bash
# HBM traffic of the run kernel of a bench.py workload from PMC counters, collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (kernel-trace only).  Usage on a GPU box:   bash tools/pmc_traffic.sh <workload> <iters>
# Prints the per-launch means; profiles/pmc_traffic.json records them (FETCH doubled, see _how).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$1; I=$2
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  out="$R/gpurun_out/pmc_${W}_${C}"
  rm -rf "$out"
  timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$out" -- \
    python3 "$R/bench.py" --workload "$W" --iters "$I" --steps 2 --warmup 1 --no-ess --no-cpu-baseline \
    > "$out.json" 2> "$out.err"
  python3 - "$out" "$C" <<'PY'
import csv, glob, sys
out, ctr = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == ctr and "k_run" in r["Kernel_Name"] or
        (r["Counter_Name"] == ctr and "k_cp_rw" in r["Kernel_Name"])]
big = [v for v in vals if v > 0.2 * max(vals)] if vals else []
print(ctr, "launches", len(vals), "mean KB over full-size launches", sum(big) / max(1, len(big)))
PY
done
