"""Development (GPU box): solve phase of the bench scene and a deterministic dense system with the library SRK_BA_LIBRARY
names (or the product library).  usage: python tools/r4_ab_solve.py <tag> [config]; writes gpurun_out/ab/<tag>_x.npy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import surikatoko_amd as sa
from surikatoko_amd import ba as B

tag = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "C3_1kcam_100kpt"
os.makedirs("gpurun_out/ab", exist_ok=True)
h = sa.BundleAdjustmentKanatani(0)
h.set_speculation(False)
rng = np.random.RandomState(7)
xs = []
for n in (300, 1280):
    A = rng.randn(n, n)
    A = A @ A.T + n * np.eye(n) + np.diag(np.arange(n) * 0.37)
    b = rng.randn(n)
    for fused in (1, 0):
        h.set_solver_fusion(fused)
        ok, x, _ = h.dense_spd_solve(A, b)
        assert ok
        xs.append(x)
        print(tag, "dense", n, "fused", fused, "err vs numpy", np.abs(x - np.linalg.solve(A, b)).max())
h.set_solver_fusion(1)
np.save("gpurun_out/ab/%s_x.npy" % tag, np.concatenate(xs))
spec = sa.CONFIGS[cfg]
sc = sa.generate_scene(spec)
assert h.upload(spec.f0, sc)
h.phase_derivatives()
h.phase_schur(1e-3)
for fused in (1, 0, 1):
    h.set_solver_fusion(fused)
    for _ in range(3):
        assert h.phase_solve()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        assert h.phase_solve()
    torch.cuda.synchronize()
    print(tag, cfg, "fused", fused, "solve phase %.4f ms" % ((time.perf_counter() - t0) / n * 1e3), "timeouts", h.solver_sync_timeouts())
h.close()
