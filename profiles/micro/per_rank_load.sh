#!/usr/bin/env bash
# The hybrid step with 1152 / N resident reservoirs on ONE GPU (SPEEDY leg replicated, no all-gather): what a rank of an N-GPU run
# carries, N = 1, 2, 4, 8.
for r in 1152 576 288 144; do
  timeout -k 10 200 python bench.py --regions $r --steps 60 --warmup 5 --no-cpu-baseline --no-training 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['per_rank'][0]
print('regions $r ms_per_step %.3f steps/s %.1f readout %.3f update %.3f speedy %.3f' % (d['ms_per_step'], 1e3/d['ms_per_step'], p['readout_ms'], p['update_ms'], p['speedy_ms']))"
done
