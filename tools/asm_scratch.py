#!/usr/bin/env python3
"""Scratch (spill) instructions of one kernel with their basic blocks.
    python tools/asm_scratch.py file.s "<demangled-name substring>" """
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
names = re.findall(r"^(_Z\S+):\s*;", s, re.M)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
t = [n for n, d in zip(names, dem) if sys.argv[2] in d]
assert len(t) == 1, t
i = s.index(t[0] + ":")
cur = None
for line in s[i:s.index(".Lfunc_end", i)].split("\n"):
    m = re.match(r"^(\.LBB\S+):", line)
    if m:
        cur = m.group(1)
    if "scratch_" in line:
        print(cur, line.strip().split(";")[0])
