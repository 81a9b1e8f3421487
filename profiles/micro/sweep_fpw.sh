# sweep of the fields per workgroup of the inverse transform (k_grid); run on the GPU box from the repo root
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for F in 2 1; do
  touch speedy-ml_amd/csrc/spectral.hip
  make -C speedy-ml_amd/csrc EXTRA=-DSML_FPW=$F > /dev/null 2>&1
  echo "FPW=$F" >> gpurun_out/fpw.log
  python -m pytest tests/test_spectral_gpu.py tests/test_dynamics_gpu.py -m gpu -q -x 2>&1 | tail -1 >> gpurun_out/fpw.log
  for r in 1 2; do python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/fpw.log; done
done
cat gpurun_out/fpw.log
