# readout kernel variants (rows per workgroup x threads) at 1152 / 576 / 288 / 144 resident reservoirs; SML_RO_VARIANT forces one
cd $GRAFT_REPO_ROOT
for R in ${REGIONS:-1152 144}; do
  for V in ${VARIANTS:--1 7 12 11 4 13 1 9 2}; do
    SML_RO_VARIANT=$V python bench.py --mode sweep --no-cpu-baseline --steps 20 --regions $R 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('regions $R variant $V readout %.3f ms %.0f GB/s' % (r['avg_launch_ms'], r['achieved']))"
  done
done
