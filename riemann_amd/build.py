"""Builds riemann_amd/csrc/libriemann_hip.so with hipcc for gfx950 (in-tree).

    python -m riemann_amd.build            # build if sources are newer
    python -m riemann_amd.build --force

hipcc cross-compiles without a GPU.  -ffp-contract=off: the random stream is
specified operation by operation (explicit fma), see csrc/rmh_device.hpp.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libriemann_hip.so")
SOURCES = ["rmh_api.hip", "rmh_run_api.hip", "rmh_kernels_mfma.hip", "rmh_kernels_logreg.hip", "rmh_kernels_mix.hip",
           "rmh_kernels_cp.hip", "rmh_kernels_big.hip", "rmh_comm.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


def _newest_source():
    names = [f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))]
    names.append(os.path.join("..", "..", "include", "riemann_hip.h"))
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in names)


def build(force=False, verbose=False):
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= _newest_source():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []

    hdr = [f for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    newest_hdr = max([os.path.getmtime(os.path.join(CSRC, f)) for f in hdr] +
                     [os.path.getmtime(os.path.join(CSRC, "..", "..", "include", "riemann_hip.h"))])

    def compile_one(src):
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if (not force and os.path.exists(obj) and
                os.path.getmtime(obj) >= max(newest_hdr, os.path.getmtime(os.path.join(CSRC, src)))):
            return obj
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # inline-asm DPP instructions are outside the compiler's hazard recogniser: scan the
    # disassembly of the objects that are about to be linked and refuse a hazardous build
    # (riemann_amd/_dpp_hazards.py; needs llvm-objdump -- RMH_SKIP_DPP_SCAN=1 builds without the scan
    # where the tool is absent, at the builder's own risk)
    from . import _dpp_hazards as chk
    if os.environ.get("RMH_SKIP_DPP_SCAN") == "1":
        if verbose:
            print("DPP hazard scan skipped (RMH_SKIP_DPP_SCAN=1)")
    elif not os.path.exists(chk.OBJDUMP):
        raise RuntimeError("riemann_amd.build: %s not found -- the DPP hazard scan of the built kernels needs "
                           "llvm-objdump (set LLVM_OBJDUMP, or RMH_SKIP_DPP_SCAN=1 to build without the scan)"
                           % chk.OBJDUMP)
    else:
        total, bad = chk.check_objects(objs)
        if bad:
            raise RuntimeError("DPP data hazards in the built kernels (riemann_amd/_dpp_hazards.py):\n  " +
                               "\n  ".join(bad[:20]))
        if verbose:
            print("DPP hazard scan: %d DPP instructions, none hazardous" % total)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl"])
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
