"""Development (GPU box): per-iteration times of the 20-iteration bench run from srk_ba_iteration_log: singles, pairs, pair + third."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import surikatoko_amd as sa
spec = sa.CONFIGS["C3_1kcam_100kpt"]; sc = sa.generate_scene(spec)
ba = sa.BundleAdjustmentKanatani(0); ba.set_profile(0)
assert ba.upload(spec.f0, sc)
ba.optimize(None, max_iterations=5); ba.reset()
for rep in range(3):
    ba.reset(); ba.optimize(None, max_iterations=20)
    log = ba.iteration_log()
    dt = np.diff(np.concatenate([[0.0], log["ms"]]))
    by = {}
    for a, t in zip(log["attempts"], dt):
        by.setdefault(int(a), []).append(t)
    print("run", rep, "attempts", int(log["attempts"].sum()), {a: (len(v), round(float(np.median(v)), 3)) for a, v in sorted(by.items())})
