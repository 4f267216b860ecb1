#!/usr/bin/env python3
"""Copy the summaries of a tools/collect_profiles.sh session from gpurun_out/ into profiles/ (tracked):
    python tools/gather_profiles.py r02
kernel stats CSVs, bench lines (plain and under rocprofv3), SQ counters, and profiles/pmc_traffic.json
(HBM bytes per chain-iteration of the run kernels: FETCH_SIZE doubled as the gfx950 note of
MI355X_MICROARCH.md prescribes, WRITE_SIZE as is, KB x 1024)."""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
ITERS = {"logreg_mala": 10, "logreg_hmc": 2, "gauss_rw": 200, "changepoint_rw": 500}
ALGO = {"logreg_mala": 4024, "logreg_hmc": 4024, "gauss_rw": 808, "changepoint_rw": 104}


def last_json_line(path):
    out = None
    for line in open(path):
        if line.startswith("{"):
            out = line
    return out


for w in ("logreg_mala", "logreg_hmc", "gauss_rw", "changepoint_rw", "pt_gauss"):
    st = glob.glob(os.path.join(G, "prof", w, "**", "*kernel_stats.csv"), recursive=True)
    if st:   # gpurun_out/ accumulates over sessions: the newest run is the one that belongs to this round
        shutil.copy(max(st, key=os.path.getmtime), os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, w)))
    for src, dst in (("%s_bench.json" % w, "%s_%s_bench.json"), ("%s_under.json" % w, "%s_%s_bench_under_rocprof.json")):
        f = os.path.join(G, "prof", src)
        if os.path.exists(f) and last_json_line(f):
            open(os.path.join(P, dst % (tag, w)), "w").write(last_json_line(f))
f = os.path.join(G, tag, "bench_default.json")
if os.path.exists(f) and last_json_line(f):
    open(os.path.join(P, "%s_bench_default.json" % tag), "w").write(last_json_line(f))
sq = {}
for w in ITERS:
    f = os.path.join(G, tag, "sq_%s.json" % w)
    try:
        sq[w] = json.load(open(f))
    except Exception:
        pass
if sq:
    json.dump(sq, open(os.path.join(P, "%s_sq_counters.json" % tag), "w"), indent=1)
pmc_path = os.path.join(P, "pmc_traffic.json")
pmc = json.load(open(pmc_path))
for w, it in ITERS.items():
    f = os.path.join(G, tag, "pmc_%s.txt" % w)
    if not os.path.exists(f):
        continue
    txt = open(f).read()
    fe = re.search(r"FETCH_SIZE launches \d+ mean KB over full-size launches ([\d.]+)", txt)
    wr = re.search(r"WRITE_SIZE launches \d+ mean KB over full-size launches ([\d.]+)", txt)
    if not fe or not wr:
        continue
    fe, wr = float(fe.group(1)) * 1024, float(wr.group(1)) * 1024
    e = pmc.get(w, {})
    e.update({"round": int(tag[1:]), "iters_per_launch": it, "chains": 65536, "fetch_GB_raw": round(fe / 1e9, 4),
              "write_GB": round(wr / 1e9, 4),
              "hbm_bytes_per_chain_iteration": round((2 * fe + wr) / (65536 * it), 1),
              "algorithmic_bytes_per_chain_iteration": ALGO[w]})
    if w in sq and "SQ_INSTS_VALU" in sq[w]:   # (a wavefront holds 16 chains, or 64 in k_cp_rw_ss64)
        e.pop("valu_insts_per_wave_iteration", None)
        e["valu_insts_per_chain_iteration"] = round(sq[w]["SQ_INSTS_VALU"] / (65536.0 * it), 2)
    pmc[w] = e
json.dump(pmc, open(pmc_path, "w"), indent=1)
print("profiles/ updated for", tag)
