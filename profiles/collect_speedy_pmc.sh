#!/usr/bin/env bash
# Counter evidence for the SPEEDY window's four kernels (k_grid, k_spec, k_gridtend_physics, k_spectral): separate rocprofv3 --pmc
# passes of the default hybrid step (program directly after "--", never combined with a tracing domain), one per counter group
# that fits the gfx950 slots (SQ 8, TCC 4, GRBM 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  Run from the repo root through
# gpurun; profiles/summarize_speedy_pmc.py turns the CSVs under gpurun_out/ into profiles/<tag>_speedy_pmc.json.
set -eo pipefail
TAG="${1:-r3}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_available.txt 2>&1 || true
have() {   # keep only the counters this rocprofv3 knows
    local out=""
    for c in "$@"; do
        if grep -qw "$c" gpurun_out/counters_available.txt; then out="$out $c"; else echo "counter $c not offered on this box" >&2; fi
    done
    echo $out
}
run_pass() {
    local name="$1"; shift
    local ctrs
    ctrs=$(have "$@")
    [ -z "$ctrs" ] && return 0
    echo "pass $name: $ctrs"
    rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/spmc_${name}_${TAG} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-training \
        > gpurun_out/spmc_${name}_${TAG}.json 2> gpurun_out/spmc_${name}_${TAG}.err
}
run_pass sq1 GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run_pass sq2 GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
echo "collected speedy pmc ${TAG}"
