"""Two identical closed-loop runs of the hybrid model (full SPEEDY window with physics, small reservoirs) must end in the same bits;
prints the ranges along the way.  A race between the two wavefronts of k_gridtend_physics, or anything else order-dependent, shows
up here."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package; load_package()
from speedy_ml_amd import domain, hybrid, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sea = synth.land_mask(); classes = hybrid.region_classes(sea)
finals = []
for run in range(2):
    m = hybrid.HybridRank(list(range(1152)), classes, sea_mask=sea, mode="hybrid", n_override=None if os.environ.get("SML_FULL_SIZE") else 1)
    st = torch.cuda.current_stream()
    eng = hybrid.NativeEngine(m) if os.environ.get("SML_SOAK_ENGINE") else None      # the native engine (sml_hybrid_step: fused hand-off) instead of the Python host
    t0 = time.time()
    for k in range(steps):
        (eng or m).step(st)
    torch.cuda.synchronize()
    if eng:
        g, f = eng.state()
        G, Fs, safe = torch.from_numpy(g), torch.from_numpy(f), int(eng.safe())
        state = m.feedback.clone()                                       # (the bank's next inputs: what the engine's last gather left)
        eng.close()
    else:
        G, Fs, safe, state = m.G.clone(), m.F.clone(), int(m.safe.item()), m.state.clone()
    F = Fs[:domain.G2_OFF].reshape(8, 48, 96, 4)
    print("run", run, "engine" if eng else "python host", "steps", steps, "%.1f s" % (time.time() - t0), "safe", safe, "finite", bool(torch.isfinite(Fs[:domain.GP_OFF]).all()),
          "T %.1f..%.1f" % (float(F[..., 0].min()), float(F[..., 0].max())), flush=True)
    finals.append((G, Fs, state))
    del m
same = all(torch.equal(a, b) for a, b in zip(finals[0], finals[1]))
print("bitwise identical:", same)
sys.exit(0 if same else 1)
