import numpy as np, surikatoko_amd as sa
from surikatoko_amd import ba as B
spec = sa.SceneSpec(n_frames=330, grid_nx=60, grid_ny=40, vis_window=8)
sc = sa.generate_scene(spec)
g = sa.BundleAdjustmentKanatani(0)
g.set_rcs_mode(2); g.upload(spec.f0, sc); print("chunks", g.rcs_chunks(), "fill", g.rcs_fill())
g.phase_derivatives(); g.phase_schur(1e-3)
S = g.buffer(B.BUF_RCS).reshape(3300,3300); rhs = g.buffer(B.BUF_RCS_RHS)
ok = g.phase_solve(); print("ok", ok)
dc = g.buffer(B.BUF_CORRECTIONS)[3*sc.N:]
x = np.linalg.solve(S, rhs)
print("nan count", np.isnan(dc).sum(), "first bad", np.where(~np.isfinite(dc))[0][:5])
err = np.abs(dc - x); print("max err", np.nanmax(err), "at", np.nanargmax(err), "scale", np.abs(x).max())
for lo in range(0, 3300, 256): print(lo, float(np.nanmax(err[lo:lo+256])))
