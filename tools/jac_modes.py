"""GPU box: derivative phase time by kernel selection (srk_ba_set_jacobian_mode) on C2, C2 with dropped observations, C1, C3."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import surikatoko_amd as sa, torch
def scene(name, drop):
    sc = sa.config_scene(name)
    return sa.drop_observations(sc, drop, seed=0) if drop else sc
for name, drop in (("C2_200cam_20kpt", 0), ("C2_200cam_20kpt", 0.1), ("C1_dino_standin", 0), ("C3_1kcam_100kpt", 0), ("C3_1kcam_100kpt", 0.1)):
    sc = scene(name, drop); f0 = sa.CONFIGS[name].f0
    out = []
    for mode in (-1, 0, 1, 2):
        ba = sa.BundleAdjustmentKanatani(0); ba.set_jacobian_mode(mode)
        assert ba.upload(f0, sc)
        ba.phase_error()
        for _ in range(5): ba.phase_derivatives()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): ba.phase_derivatives()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
        out.append("mode %2d kernel %d %.1f us" % (mode, ba.jacobian_kernel(), dt * 1e6))
        ba.close()
    print(name, "drop", drop, "|", " | ".join(out), flush=True)
