"""Phase times on the all-visible scenes the reference's demos hand to the path (every track in every frame): the Schur
sum of tracks over more than 24 frames (k_schur_long; SRK_SCHUR_NO_LONG=1 = the per-landmark kernel it replaced)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import surikatoko_amd as sa, torch
for name, spec in (("circle_grid 36 x 81", sa.SceneSpec(36, 9, 9, vis_window=0)),
                   ("mvf flagfile 60 x 3321", sa.SceneSpec(60, 81, 41, vis_window=0)),
                   ("window 40, 200 x 20000", sa.SceneSpec(200, 200, 100, vis_window=40)),
                   ("all visible, 200 x 5000 (dense variant of config 2, a quarter of its points)", sa.SceneSpec(200, 100, 50, vis_window=0))):
    sc = sa.generate_scene(spec)
    ba = sa.BundleAdjustmentKanatani(0)
    ba.set_profile(1)
    assert ba.upload(spec.f0, sc)
    ba.optimize(None, max_iterations=2); ba.reset()
    torch.cuda.synchronize(); t = time.perf_counter()
    ba.optimize(None, max_iterations=5); torch.cuda.synchronize(); dt = time.perf_counter() - t
    r = ba.report
    print(f"{name}: N={sc.N} O={sc.O} it={r.iterations} attempts={r.attempts} {dt * 1e3 / max(r.attempts, 1):.3f} ms/attempt; per attempt: "
          f"schur {r.ms_schur / max(r.attempts, 1):.3f} solve {r.ms_solve / max(r.attempts, 1):.3f} backsub {r.ms_backsub / max(r.attempts, 1):.3f} "
          f"jac/iter {r.ms_jacobian / max(r.iterations, 1):.3f}; err {r.err_initial:.4e} -> {r.err_final:.4e}")
    ba.close()
