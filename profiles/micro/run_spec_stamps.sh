# phase stamps of two k_spec workgroups inside a time step's forward launch; run on the GPU box from the repo root
set -e
cd $GRAFT_REPO_ROOT
for W in 100 250; do
  touch speedy-ml_amd/csrc/spectral.hip
  make -C speedy-ml_amd/csrc EXTRA=-DSML_GRID_STAMPS=$W > /dev/null 2>&1
  echo "workgroup $W"; python profiles/micro/spec_phase_stamps.py 2>/dev/null | tail -2
done
