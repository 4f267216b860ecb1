"""Static check of the hand-written DPP instructions in the built kernels.

`v_fmac_f64_dpp` (rmh_run.hpp TailAcc, rmh_models.hpp GradTail) reads its first source through the
DPP row broadcast; on gfx9-family hardware
  * a VGPR written by a VALU instruction must not be read through DPP within the next two wait
    states, and
  * a VALU write of EXEC must be five wait states away from a DPP instruction.
The compiler pads such pairs itself but does not look inside inline asm, so the build scans the
DISASSEMBLY OF THE OBJECTS IT LINKS (riemann_amd/build.py calls check_objects and fails on a
hit); the CPU test suite runs the same scan.

Conservative where the listing is not straight-line code: a DPP read fewer than two wait states
after a block entry (a label, or the target of a branch in the disassembly) counts as a hazard,
because the predecessor blocks are not followed.  Every register an instruction writes is
considered (both operands of v_permlane16/32_swap, the whole accumulator of an MFMA).

    python -m riemann_amd._dpp_hazards [objects...]    # default: every object of the library
(tools/check_dpp_hazards.py is the same entry point.)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def _regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _parse(text):
    """[(instruction text, address or None, is_block_entry)] of an assembly listing (.s) or an
    llvm-objdump disassembly."""
    out, entry = [], True
    for line in text.splitlines():
        if re.match(r"^\s*(\.L\S+|[0-9a-fA-F]+ <\S+>|[A-Za-z_]\S*):", line):      # label / function start
            entry = True
            continue
        m = re.match(r"^\s+([a-z][^/;]*?)\s*(?://\s*([0-9A-Fa-f]+):.*)?(?:;.*)?$", line)
        if not m or not re.match(r"^[sv]_|^ds_|^buffer_|^global_|^flat_|^scratch_", m.group(1)):
            continue
        out.append([m.group(1).strip(), int(m.group(2), 16) if m.group(2) else None, entry])
        entry = False
    # branch targets of a disassembly (simm16 in dwords from the next instruction)
    addr = {a: i for i, (_, a, _) in enumerate(out) if a is not None}
    for t, a, _ in out:
        m = re.match(r"s_c?branch\S*\s+(\d+)$", t)
        if m and a is not None:
            off = int(m.group(1))
            off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
            if tgt in addr:
                out[addr[tgt]][2] = True
    return out


def _writes(t):
    """VGPRs written by instruction text t; 'exec' in the set if a VALU instruction writes EXEC."""
    ops = t.split(None, 1)
    if len(ops) < 2 or not ops[0].startswith("v_"):
        return set()
    op, args = ops[0], [a.strip() for a in ops[1].split(",")]
    w = set()
    if op.startswith("v_cmpx") or args[0] == "exec":
        w.add("exec")
    if op.startswith(("v_readlane", "v_readfirstlane", "v_cmp")):
        return w
    w |= _regs(args[0])
    if "swap" in op and len(args) > 1:
        w |= _regs(args[1])
    return w


def scan(text):
    """(number of DPP instructions, list of hazard descriptions)."""
    ins = _parse(text)
    total, bad = 0, []
    for k, (t, _, _) in enumerate(ins):
        if not t.startswith("v_") or ("_dpp" not in t.split()[0] and " row_" not in t and " quad_perm" not in t):
            continue
        total += 1
        src = _regs(t.split(None, 1)[1].split(",")[1])        # the DPP source operand
        ws, j = 0, k
        while ws < 5:
            if ins[j][2] and ws < 2:    # control may arrive here from a block that is not followed
                bad.append("block entry %d wait state(s) before  %s" % (ws, t))
                break
            j -= 1
            if j < 0:
                break
            p = ins[j][0]
            if p.startswith("s_nop"):
                ws += int(p.split()[1]) + 1
                continue
            w = _writes(p)
            if ws < 2 and (w & src):
                bad.append("%s  ->  %s" % (p, t))
            if "exec" in w:
                bad.append("%s  (EXEC)  ->  %s" % (p, t))
            ws += 1
    return total, bad


def scan_object(obj):
    """Disassemble the gfx950 code object bundled in a hipcc object file and scan it."""
    subprocess.check_call([OBJDUMP, "--offloading", obj], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    d = os.path.dirname(obj)
    parts = [os.path.join(d, f) for f in os.listdir(d) if f.startswith(os.path.basename(obj) + ".")]
    total, bad = 0, []
    try:
        for f in parts:
            if "amdgcn" in f:
                dis = subprocess.run([OBJDUMP, "-d", f], capture_output=True, text=True, check=True).stdout
                n, b = scan(dis)
                total += n
                bad += b
    finally:
        for f in parts:
            os.remove(f)
    return total, bad


def check_objects(objs):
    """Scan every object; returns (DPP instructions seen, hazards)."""
    total, bad = 0, []
    for o in objs:
        n, b = scan_object(o)
        total += n
        bad += ["%s: %s" % (os.path.basename(o), x) for x in b]
    return total, bad


def main(objs=None):
    if not objs:
        from .build import CSRC, SOURCES
        objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    total, bad = check_objects(objs)
    print("%d DPP instructions, %d hazards" % (total, len(bad)))
    for b in bad:
        print("HAZARD:", b)
    return total, bad


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1:])[1] else 0)
