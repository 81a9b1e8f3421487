#!/usr/bin/env bash
# Training pass (20 batches of 98 columns): workgroups per reservoir in the recurrence's update launches (SML_UPD_CFG = parts; -1 = the
# library's choice) for 32 and 64 resident reservoirs.  Measured (ms per pass): 32 residents 63.2 at 4 parts, 59.7 / 60.1 / 59.2 / 59.0 /
# 60.0 at 6 / 8 / 12 / 16 / 24; 64 residents 112.2 / 112.0 / 112.2 / 109.3 / 115.1 / 118.0.  (The same script once carried a form
# with the Gram flushes on a CU-masked second stream beside the recurrence of the next batches: 69-90 ms at 32 residents and 134-151
# at 64 with 16-64 reserved CUs -- the recurrence beside the matrix-core product is slower than the two in turn; not kept.)
set -e
for nres in 32 64; do
  for cfg in -1 4 6 8 12 16 24; do
    echo "NRES=$nres SML_UPD_CFG=$cfg"; NRES=$nres SML_UPD_CFG=$cfg timeout -k 10 300 python profiles/micro/train_pass_time.py 2>/dev/null | tail -1
  done
done
