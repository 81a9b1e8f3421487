# phase stamps of a few k_grid workgroups inside a time step's inverse launch; run on the GPU box from the repo root
set -e
cd $GRAFT_REPO_ROOT
for W in 100 333; do
  touch speedy-ml_amd/csrc/spectral.hip
  make -C speedy-ml_amd/csrc EXTRA=-DSML_GRID_STAMPS=$W > /dev/null 2>&1
  echo "workgroup $W"; python profiles/micro/grid_phase_stamps.py 2>/dev/null | tail -2
done
