#!/usr/bin/env bash
# What one rank of an N-GPU job computes per step (its share of the regions + the replicated SPEEDY window), sequential against
# software-pipelined schedule (SML_PIPELINE=1: next step's reservoir advance + state block of the readout under the SPEEDY window),
# emulated on ONE GPU with bench.py --regions R (1152 / N regions resident; the peers' outvecs are absent, so the grid is not physical
# and the range guard is ignored -- load emulation only).  Python host (the pipelined schedule exists there only).
# Run on the GPU box from the repo root; writes profiles-ready JSON to gpurun_out/<tag>_per_rank_pipeline.json
set -eo pipefail
TAG="${1:-r4}"
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_per_rank_pipeline.json
echo "[" > $OUT
first=1
for R in 1152 576 288 144; do
  for P in 0 1; do
    line=$(SML_PIPELINE=$P python bench.py --host python --regions $R --steps 120 --warmup 10 --no-cpu-baseline --no-training 2>/dev/null | \
           python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'regions': $R, 'ranks_emulated': 1152 // $R, 'pipeline': $P, 'ms_per_step': d['ms_per_step'], 'per_rank': d['per_rank'][0]}))")
    [ $first = 1 ] || echo "," >> $OUT
    first=0
    echo "$line" >> $OUT
    echo "$line"
  done
done
echo "]" >> $OUT
