#!/usr/bin/env bash
# Collects the round's rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   1. --kernel-trace --stats of the default bench command            -> gpurun_out/prof_<tag>/
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (sweep-only bench, 3 steps) -> gpurun_out/pmc_*_<tag>/
# Counters are never combined with tracing domains (the pool refuses that combination).
set -eo pipefail
TAG="${1:-r1}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-training --no-float32-block \
    > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_${TAG}.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_${TAG} -- python3 bench.py --mode sweep --steps 3 --warmup 1 --no-cpu-baseline \
    > /dev/null 2> gpurun_out/pmc_fetch_${TAG}.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_${TAG} -- python3 bench.py --mode sweep --steps 3 --warmup 1 --no-cpu-baseline \
    > /dev/null 2> gpurun_out/pmc_write_${TAG}.err
echo "collected ${TAG}"
