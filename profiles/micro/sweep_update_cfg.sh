# k_update work split (parts per reservoir) at small resident counts; SML_UPD_CFG forces a configuration (3..8 = parts, 2 = 256 threads x 4 parts)
cd $GRAFT_REPO_ROOT
for R in ${REGIONS:-144 288}; do
  for V in ${CFGS:--1 3 4 6 8 2}; do
    SML_UPD_CFG=$V python bench.py --mode sweep --no-cpu-baseline --steps 30 --regions $R 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['secondary']
print('regions $R cfg $V update %.4f ms %.0f GB/s' % (r['avg_launch_ms'], r['achieved']))"
  done
done
