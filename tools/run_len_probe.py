"""GPU box, development library (SRK_BA_LIBRARY=surikatoko_amd/libsrk_ba_dev.so): Schur phase and LM iteration time by the
number of equal parts the Schur kernels' landmark runs are cut into (SRK_SCHUR_RUN_SPLIT; unset = the product's rule) on small and mid-size scenes."""
import os, subprocess, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
SCENES = {  # name -> (config name | SceneSpec arguments, observations dropped)
    "C1_dino_standin": ("C1_dino_standin", 0), "C1_drop10": ("C1_dino_standin", 0.1),
    "mvf_30x1206_w12": ((30, 67, 18, 12), 0), "100x5000_w20": ((100, 100, 50, 20), 0), "C2_200cam_20kpt": ("C2_200cam_20kpt", 0),
    "C2_drop10": ("C2_200cam_20kpt", 0.1),
}
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import surikatoko_amd as sa, torch
    name = sys.argv[2]
    cfg, drop = SCENES[name]
    if isinstance(cfg, str):
        sc, f0 = sa.config_scene(cfg), sa.CONFIGS[cfg].f0
    else:
        spec = sa.SceneSpec(cfg[0], cfg[1], cfg[2], vis_window=cfg[3]); sc, f0 = sa.generate_scene(spec), spec.f0
    if drop: sc = sa.drop_observations(sc, drop, seed=0)
    ba = sa.BundleAdjustmentKanatani(0)
    assert ba.upload(f0, sc)
    ba.phase_error(); ba.phase_derivatives()
    for _ in range(5): ba.phase_schur(1e-3)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): ba.phase_schur(1e-3)
    torch.cuda.synchronize(); schur = (time.perf_counter() - t) / 50
    crit = sa.BundleAdjustmentKanataniTermCriteria(); crit.AllowedReprojErrRelativeChange(1e-30)
    best = 1e9
    for _ in range(3):
        ba.reset(); torch.cuda.synchronize(); t = time.perf_counter()
        ba.optimize(crit, 10); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    r = ba.report
    print(f"{name:18s} split {os.environ.get('SRK_SCHUR_RUN_SPLIT', 'rule'):>4s}: Schur phase {schur * 1e6:7.1f} us, {r.iterations} iterations / {r.attempts} attempts "
          f"in {best * 1e3:7.3f} ms = {best * 1e3 / max(1, r.attempts):.3f} ms an attempt, err {r.err_final:.9e}", flush=True)
    ba.close()
    sys.exit(0)
for name in SCENES:
    for cap in ("1", "2", "3", "4", "6", "8", None):
        env = dict(os.environ)
        env.pop("SRK_SCHUR_RUN_SPLIT", None)
        if cap: env["SRK_SCHUR_RUN_SPLIT"] = cap
        subprocess.run([sys.executable, os.path.abspath(__file__), "worker", name], env=env, check=False, timeout=300)
