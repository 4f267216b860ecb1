"""Static check of the hand-written DPP instructions in the built kernels.

`v_fmac_f64_dpp` (rmh_run.hpp, TailAcc) reads its first source through the DPP row broadcast; on
gfx9-family hardware a VGPR written by a VALU instruction must not be read through DPP within the
next two wait states.  The compiler pads such pairs itself but does not look inside inline asm, so
this script compiles the translation unit to assembly and checks every occurrence.

    python tools/check_dpp_hazards.py          # exit status 1 if a hazard is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemann_amd.build import CSRC, FLAGS  # noqa: E402


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(asm_text):
    """(number of DPP fmacs, list of hazard descriptions) of a gfx950 assembly listing."""
    ins = [l.strip() for l in asm_text.splitlines() if re.match(r"^\s+[a-z]", l)]
    total, bad = 0, []
    for k, l in enumerate(ins):
        if not l.startswith("v_fmac_f64_dpp"):
            continue
        total += 1
        src = _regs(l.split()[2].rstrip(","))
        ws, j = 0, k - 1
        while j >= 0 and ws < 2:
            p = ins[j]
            if p.startswith("s_nop"):
                ws += int(p.split()[1]) + 1
            else:
                if p.startswith("v_") and _regs(p.split()[1].rstrip(",")) & src:
                    bad.append("%s  ->  %s" % (p, l))
                ws += 1
            j -= 1
    return total, bad


def main(source="rmh_kernels_mfma.hip"):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call([hipcc] + FLAGS + ["--offload-device-only", "-S", os.path.join(CSRC, source), "-o", out],
                              stderr=subprocess.DEVNULL)
        total, bad = scan(open(out).read())
    print("%d v_fmac_f64_dpp, %d hazards" % (total, len(bad)))
    for b in bad:
        print("HAZARD:", b)
    return total, bad


if __name__ == "__main__":
    sys.exit(1 if main()[1] else 0)
