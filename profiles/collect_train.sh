#!/usr/bin/env bash
# Round evidence for the training kernels (BASELINE config 4) on the GPU box, run from the repo root through gpurun:
#   1. rocprofv3 --kernel-trace --stats of profiles/bench_train.py (Gram updates, single / 8 / 16 ridge solves)
#        -> gpurun_out/trainprof_<tag>/ (kernel_stats.csv is copied to profiles/<tag>_train_kernel_stats.csv afterwards)
#   2. rocprofv3 --kernel-trace of three single ridge solves with each solver; profiles/micro/trace_summary.py prints the per-kernel
#      totals and the span of the LAST solve -> gpurun_out/<tag>_lu_timeline.txt
set -eo pipefail
TAG="${1:-r3}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trainprof_${TAG} -- python3 profiles/bench_train.py \
    > gpurun_out/trainprof_${TAG}.json 2> gpurun_out/trainprof_${TAG}.err
: > gpurun_out/${TAG}_lu_timeline.txt
for solver in chol lu; do
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fittrace_${solver}_${TAG} -- python3 profiles/micro/fit_solvers.py ${solver} 3 \
        > gpurun_out/fittrace_${solver}_${TAG}.log 2> gpurun_out/fittrace_${solver}_${TAG}.err
    f=$(ls -t gpurun_out/fittrace_${solver}_${TAG}/*/*kernel_trace.csv | head -1)
    {
        echo "== single ridge solve, 5892 x 5892 + 136 right-hand sides, solver = ${solver} (rocprofv3 --kernel-trace of profiles/micro/fit_solvers.py ${solver} 3; the last solve) =="
        cat gpurun_out/fittrace_${solver}_${TAG}.log
        python3 profiles/micro/trace_summary.py "$f" 24
        echo
    } >> gpurun_out/${TAG}_lu_timeline.txt
done
echo "collected train ${TAG}"
