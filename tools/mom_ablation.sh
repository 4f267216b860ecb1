#!/bin/bash
# Where the in-kernel moment accumulation of the AdaptCov walk (MomFused, rmh_run.hpp) spends
# its time: rebuilds rmh_kernels_mfma.o with pieces compiled out (RMH_MOM_ABL) and times
# tools/adaptcov_timing.py's squares schedule.  Timing only -- the ablated builds produce wrong
# moments.  Run on the GPU box; restores the real build at the end.
#   1: no moment code at all   2: barriers + panel writes only   3: + operand reads
#   6: ONE barrier per iteration and nothing else (lockstep cost)   7: the nine barriers and nothing else
#   9: full, and workgroup 3 prints clock64 differences between its barriers at iteration 7   0: full
set -e
cd "$(dirname "$0")/.."
C=riemann_amd/csrc
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-function"
for abl in ${@:-1 6 7 2 3 0}; do
  /opt/rocm/bin/hipcc $FLAGS -DRMH_MOM_ABL=$abl -c $C/rmh_kernels_mfma.hip -o $C/rmh_kernels_mfma.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/libriemann_hip.so $C/rmh_api.o $C/rmh_run_api.o \
      $C/rmh_kernels_mfma.o $C/rmh_kernels_logreg.o $C/rmh_kernels_mix.o $C/rmh_kernels_cp.o $C/rmh_kernels_big.o $C/rmh_comm.o -ldl
  echo "RMH_MOM_ABL=$abl"
  python tools/adaptcov_timing.py 2>&1 | grep -v plain
done
