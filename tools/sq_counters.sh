#!/bin/bash
# SQ counters of the run kernel of a bench.py workload (two rocprofv3 --pmc passes, kernel-trace
# only).  Usage on a GPU box:  bash tools/sq_counters.sh <workload> <iters>
# Prints one JSON object with the per-launch means and a few derived fractions
# (profiles/r01_sq_counters.json keeps them).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=$1; I=$2
export TMPDIR=/tmp
cd /tmp
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
n=0
for P in "$P1" "$P2"; do
  n=$((n+1)); out="$R/gpurun_out/sq_${W}_$n"; rm -rf "$out"
  timeout -k 10 500 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$out" -- \
    python3 "$R/bench.py" --workload "$W" --iters "$I" --steps 2 --warmup 1 --no-ess --no-cpu-baseline \
    > "$out.json" 2> "$out.err"
done
python3 - "$R/gpurun_out/sq_${W}" "$I" <<'PY'
import csv, glob, json, sys
base, iters = sys.argv[1], int(sys.argv[2])
vals = {}
for f in glob.glob(base + "_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_run" in r["Kernel_Name"] or "k_cp_rw" in r["Kernel_Name"] or "k_cprj" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {}
for k, v in vals.items():
    big = [x for x in v if x > 0.2 * max(v)]
    m[k] = sum(big) / len(big) if big else 0.0
d = {}
if "SQ_WAVE_CYCLES" in m:
    simd_cycles = 4.0 * m["SQ_BUSY_CYCLES"] * 8    # rough: not used below
    d["wait_any_frac_of_wave_cycles"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    d["wait_inst_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    d["active_frac"] = m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]
if "SQ_INSTS_MFMA" in m:
    if m["SQ_INSTS_MFMA"] > 0:
        d["valu_insts_per_mfma"] = m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"]
    d["lds_conflict_frac_of_lds_cycles"] = m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"])
d["iters_per_launch"] = iters
if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
    # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum the 1024 SIMDs
    d["kernel_cycles"] = m["GRBM_GUI_ACTIVE"] / 8.0
    d["mfma_busy_frac_of_simd_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / d["kernel_cycles"]
m["derived"] = d
print(json.dumps(m, indent=1))
PY
