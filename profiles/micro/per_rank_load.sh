# what one rank of an N-GPU run does per step: the hybrid step with 1152/N resident reservoirs (the SPEEDY leg is replicated)
cd $GRAFT_REPO_ROOT
for R in 1152 576 288 144; do
  python bench.py --no-cpu-baseline --steps 30 --regions $R 2>gpurun_out/perrank.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print($R, 'ms/step %.3f' % d['ms_per_step'], 'readout %.3f ms %.0f GB/s' % (r['avg_launch_ms'], r['achieved']), 'update %.3f ms %.0f GB/s' % (r['secondary']['avg_launch_ms'], r['secondary']['achieved']))"
done
