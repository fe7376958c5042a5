import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
import surikatoko_amd as sa, torch
spec = sa.SceneSpec(n_frames=29, grid_nx=40, grid_ny=30, vis_window=4, f0=1.0)
sc0 = sa.generate_scene(spec)
sc = sa.Scene(sc0.points, sc0.cam_R, sc0.cam_T, sc0.K[0:1], 1, sc0.row_ptr, sc0.obs_frame, sc0.obs_uv)
print(sc.N, sc.M, sc.O)
ba = sa.BundleAdjustmentKanatani(0)
crit = sa.BundleAdjustmentKanataniTermCriteria(); crit.AllowedReprojErrRelativeChange(1e-3)
for rep in range(4):
    t0=time.perf_counter(); ba.upload(1.0, sc); torch.cuda.synchronize(); t1=time.perf_counter()
    ba.optimize(crit, max_iterations=50); torch.cuda.synchronize(); t2=time.perf_counter()
    out = sc.copy(); ba.download(out); t3=time.perf_counter()
    print("upload %.3f ms  optimize %.3f ms (%d it, %d att)  download %.3f ms" % (1e3*(t1-t0), 1e3*(t2-t1), ba.report.iterations, ba.report.attempts, 1e3*(t3-t2)))
s2 = sc.copy()
t0=time.perf_counter(); ba.ComputeInplace(1.0, s2, crit, 50); print("compute_inplace %.3f ms" % (1e3*(time.perf_counter()-t0)))
