#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a hipcc --cuda-device-only -S assembly file
(the AMDGPU metadata at its end).   python tools/kernel_resources.py file.s [name filter]"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for b in s.split("  - .agpr_count:")[1:]:
        name = demangle(re.search(r"\.name:\s+(\S+)", b).group(1))
        if flt not in name:
            continue
        g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)  # noqa: E731
        print("vgpr %3s agpr %3s spill %3s scratch %4s sgpr %3s lds %6s  %s" % (
            g("vgpr_count"), b.split()[0], g("vgpr_spill_count"), g("private_segment_fixed_size"),
            g("sgpr_count"), g("group_segment_fixed_size"), name[:150]))


if __name__ == "__main__":
    main()
