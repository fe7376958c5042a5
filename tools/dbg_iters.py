import sys, time
sys.path.insert(0, "/root/repo")
import surikatoko_amd as sa, torch
spec = sa.CONFIGS["C3_1kcam_100kpt"]; sc = sa.generate_scene(spec)
ba = sa.BundleAdjustmentKanatani(0); ba.set_profile(0)
assert ba.upload(spec.f0, sc)
ba.optimize(None, max_iterations=2); ba.reset()
for K in (1, 5, 10, 20):
    ba.reset(); torch.cuda.synchronize(); t = time.perf_counter()
    ba.optimize(None, max_iterations=K); torch.cuda.synchronize(); dt = time.perf_counter() - t
    r = ba.report
    print(K, "iterations", r.iterations, "attempts", r.attempts, "ms/iter %.3f" % (dt * 1e3 / max(r.iterations,1)), "err", r.err_initial, "->", r.err_final, "status", r.status)
