set -e
cd $GRAFT_REPO_ROOT
for L in 3 6 2 4; do
  touch speedy-ml_amd/csrc/spectral.hip
  make -C speedy-ml_amd/csrc EXTRA=-DSML_LATG=$L > /dev/null 2>&1
  echo "LATG=$L" >> gpurun_out/latg.log
  python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/latg.log
  python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/latg.log
done
cat gpurun_out/latg.log
