#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc --cuda-device-only -S file: where the
MFMAs, the VALU work, the LDS reads and the scratch (spill) traffic sit relative to the loops.
    python tools/asm_blocks.py file.s "<demangled-name substring>" [min_instructions]"""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    names = re.findall(r"^(_Z\S+):\s*;", s, re.M)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
    target = [n for n, d in zip(names, dem) if want in d]
    if len(target) != 1:
        sys.exit("matches: %r" % [d for d in dem if want in d])
    start = s.index(target[0] + ":")
    body = s[start:s.index(".Lfunc_end", start)]
    blocks, order, cur = {}, [], "entry"
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB\S+):", line)
        if m:
            cur = m.group(1)
        t = line.strip().split()
        if not t or t[0][0] in ";." or t[0].endswith(":"):
            continue
        if cur not in blocks:
            blocks[cur] = dict(n=0, mfma=0, valu=0, dpp=0, ds=0, vmem=0, scr_ld=0, scr_st=0, salu=0, wait=0, br=[])
            order.append(cur)
        b, op = blocks[cur], t[0]
        b["n"] += 1
        if op.startswith("v_mfma"):
            b["mfma"] += 1
        elif op.startswith("scratch_load"):
            b["scr_ld"] += 1
        elif op.startswith("scratch_store"):
            b["scr_st"] += 1
        elif op.startswith("ds_"):
            b["ds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_")):
            b["vmem"] += 1
        elif op.startswith("v_"):
            b["valu"] += 1
            if "dpp" in op or "row_" in line:
                b["dpp"] += 1
        elif op == "s_waitcnt":
            b["wait"] += 1
        elif op.startswith("s_"):
            b["salu"] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            b["br"].append(t[1])
    tot = dict(n=0, mfma=0, valu=0, ds=0, scr_ld=0, scr_st=0)
    for name in order:
        b = blocks[name]
        for k in tot:
            tot[k] += b[k]
        if b["n"] >= minn or b["scr_ld"] or b["scr_st"]:
            print("%-12s n %4d mfma %3d valu %4d (dpp %3d) ds %3d vmem %3d scr %2d/%2d salu %3d wait %3d -> %s" % (
                name, b["n"], b["mfma"], b["valu"], b["dpp"], b["ds"], b["vmem"], b["scr_ld"], b["scr_st"],
                b["salu"], b["wait"], ",".join(b["br"])))
    print("total", tot)


if __name__ == "__main__":
    main()
