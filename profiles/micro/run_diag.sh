set -e
cd $GRAFT_REPO_ROOT
S=$GRAFT_REPO_ROOT/speedy-ml_amd/csrc/libspeedyml_hip_stamps.so
echo "== window span"; python profiles/micro/window_span.py gpurun_out/window_span.json > gpurun_out/window_span.txt 2>&1; grep -v Warn gpurun_out/window_span.txt | head -8
echo "== k_grid stamps"; SML_LIB_PATH=$S python profiles/micro/grid_phase_stamps.py 2>/dev/null | tail -2
echo "== k_spec stamps"; SML_LIB_PATH=$S python profiles/micro/spec_phase_stamps.py 2>/dev/null | tail -2
echo "== physics stamps"; SML_LIB_PATH=$S python profiles/micro/physics_wave_stamps.py 2>/dev/null | tail -40
echo "== k_spectral stamps"; SML_LIB_PATH=$S python profiles/micro/spectral_step_stamps.py 2>/dev/null | tail -12
