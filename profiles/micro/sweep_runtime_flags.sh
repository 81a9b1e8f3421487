#!/usr/bin/env bash
# HIP runtime flags that move where kernel arguments live / how launches are queued, on the hybrid step (its SPEEDY window is 104
# dependent launches of 8-16 us): step time and the window's share.
run() {
  echo "== $*"
  env "$@" timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-training 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['per_rank'][0]
print('  ms_per_step %.4f  speedy_ms %.4f  readout_ms %.4f  update_ms %.4f' % (d['ms_per_step'], p['speedy_ms'], p['readout_ms'], p['update_ms']))"
}
run SML_NOFLAG=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run ROC_USE_FGS_KERNARG=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run HIP_FORCE_DEV_KERNARG=1 ROC_SKIP_KERNEL_ARG_COPY=1
