#!/usr/bin/env bash
# ms per hybrid step through the FORTRAN host (the drop-in modules driven by program main's loop, fortran/test_main_loop.f90): the
# checked loop first (2 steps), then SML_TEST_TIMED_STEPS bare iterations between two system_clock readings -- full-size reservoirs
# (SML_RES_M = 6000 as shipped), all 1152 regions on one rank, no slab.  Run on the GPU box from the repo root; output -> profiles/.
set -eo pipefail
TAG="${1:-r4}"
mkdir -p gpurun_out
export LD_LIBRARY_PATH="$PWD/speedy-ml_amd/csrc:/opt/rocm/lib:/opt/rocm/lib/llvm/lib:${LD_LIBRARY_PATH:-}"
SML_TEST_SLAB=0 SML_TEST_STEPS=2 SML_TEST_PREDICTIONS=1 SML_TEST_ERA_HOURS=800 SML_TEST_TIMED_STEPS=120 \
    ./speedy-ml_amd/fortran/test_main_loop > gpurun_out/${TAG}_fortran_main_loop.log 2>&1
grep -E "timed main loop|parity" gpurun_out/${TAG}_fortran_main_loop.log
# the same with weights as the reference's NetCDF weight files deliver them (float-valued: the banks read their compact copies)
SML_TEST_F32_WEIGHTS=1 SML_TEST_SLAB=0 SML_TEST_STEPS=2 SML_TEST_PREDICTIONS=1 SML_TEST_ERA_HOURS=800 SML_TEST_TIMED_STEPS=120 \
    ./speedy-ml_amd/fortran/test_main_loop > gpurun_out/${TAG}_fortran_main_loop_f32.log 2>&1
echo "with float-valued weights (SML_TEST_F32_WEIGHTS=1):"
grep -E "timed main loop|parity" gpurun_out/${TAG}_fortran_main_loop_f32.log
