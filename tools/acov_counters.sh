#!/bin/bash
# SQ counters of the AdaptCov run kernel (k_run_rw_tail<8,false,false,true>, moments in the kernel)
# beside the plain walk, cfg2 size, squares schedule: two rocprofv3 --pmc passes, kernel-trace only.
#   bash tools/acov_counters.sh   (on a GPU box; writes gpurun_out/acov_sq.json)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
n=0
for P in "$P1" "$P2"; do
  n=$((n+1)); out="$R/gpurun_out/acov_sq_$n"; rm -rf "$out"
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$out" -- \
    python3 "$R/tools/adaptcov_timing.py" 150 squares > "$out.log" 2> "$out.err"
done
python3 - "$R/gpurun_out/acov_sq" <<'PY'
import csv, glob, json, sys
base = sys.argv[1]
vals = {}
for f in glob.glob(base + "_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_run_rw_tail" not in k:
            continue
        tag = "mom" if "false, false, true" in k or "Lb0ELb0ELb1" in k else "plain"
        vals.setdefault(tag, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {}
for tag, d in vals.items():
    m = {}
    for k, v in d.items():
        big = [x for x in v if x > 0.5 * max(v)]     # the longest launches of the schedule
        m[k] = sum(big) / len(big)
        m[k + "_n"] = len(big)
    if "GRBM_GUI_ACTIVE" in m:
        kc = m["GRBM_GUI_ACTIVE"] / 8.0
        m["kernel_cycles"] = kc
        m["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / kc
        m["wait_any_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"] > 0:
        m["valu_per_mfma"] = m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"]
        m["lds_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"])
    out[tag] = m
json.dump(out, open(base + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
