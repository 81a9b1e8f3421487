#!/usr/bin/env bash
# What one rank of an N-GPU job computes per step through the NATIVE engine (its share of the regions + the replicated SPEEDY window),
# emulated on ONE GPU with bench.py --regions R (1152 / N regions resident; the peers' outvecs are absent, so the grid is not physical and
# the range guard is ignored -- load emulation only; the all-gather's latency is not in it).  Run on the GPU box from the repo root.
set -eo pipefail
TAG="${1:-r4}"
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_per_rank_native.json
echo "[" > $OUT
first=1
for R in 1152 576 288 144; do
  line=$(python bench.py --regions $R --steps 120 --warmup 10 --no-cpu-baseline --no-training 2>/dev/null | \
         python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'regions': $R, 'ranks_emulated': 1152 // $R, 'ms_per_step': d['ms_per_step'], 'steps_per_s': 1e3 / d['ms_per_step'], 'per_rank': d['per_rank'][0]}))")
  [ $first = 1 ] || echo "," >> $OUT
  first=0
  echo "$line" >> $OUT
  echo "$line"
done
echo "]" >> $OUT
