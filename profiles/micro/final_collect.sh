#!/usr/bin/env bash
# End-of-round refresh of the committed evidence (run from the repo root through gpurun): the default bench line, the hybrid step's
# kernel stats + FETCH/WRITE passes (collect.sh), the SPEEDY-window counter passes, the training kernels' stats and timelines.
TAG="${1:-r3}"
timeout -k 10 600 python bench.py > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_${TAG}_final.err
python -c "
import json; d=json.loads(open('gpurun_out/bench_${TAG}_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms']); print(d['per_rank'][0])" && timeout -k 10 500 bash profiles/collect.sh ${TAG} && timeout -k 10 300 bash profiles/collect_speedy_pmc.sh ${TAG} && timeout -k 10 300 bash profiles/collect_train.sh ${TAG}
echo done
