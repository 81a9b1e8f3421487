# latitude pairs per workgroup of the inverse transform (k_grid): 24 / LATG workgroups per field; run on the GPU box from the repo root
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for L in ${LATGS:-3 6 2 4}; do
  touch speedy-ml_amd/csrc/spectral.hip
  make -C speedy-ml_amd/csrc EXTRA=-DSML_LATG=$L > /dev/null 2>&1
  echo "LATG=$L" >> gpurun_out/latg.log
  python -m pytest tests/test_spectral_gpu.py -m gpu -q -x 2>&1 | tail -1 >> gpurun_out/latg.log
  for r in 1 2; do python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/latg.log; done
done
cat gpurun_out/latg.log
