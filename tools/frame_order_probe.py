"""GPU box: what the internal frame order buys on scenes whose frame numbers are not in time order.
(1) the headline scene C3 with its frame numbers shuffled (an unordered image set), (2) a 1000-frame sequence that closes a
loop (20-frame tracks, 20 000 landmarks).  For each: srk_ba_set_frame_reordering 0 (the caller's numbering) against the
default; 5 + 10 LM iterations, ms / iteration, chunks of the solver plan, skyline fill, the derivative kernel chosen."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import surikatoko_amd as sa, torch

def run(name, sc, f0):
    out = {}
    for mode in (0, -1):
        ba = sa.BundleAdjustmentKanatani(0); ba.set_profile(0); ba.set_frame_reordering(mode)
        t = time.perf_counter(); assert ba.upload(f0, sc); t_up = time.perf_counter() - t
        ba.optimize(None, max_iterations=5)
        torch.cuda.synchronize(); t = time.perf_counter()
        ba.optimize(None, max_iterations=10); torch.cuda.synchronize(); dt = time.perf_counter() - t
        r = ba.report
        out["callers_order" if mode == 0 else "renumbered"] = dict(
            ms_per_iteration=round(dt * 1e3 / r.iterations, 3), iterations=r.iterations, attempts=r.attempts, err_final=r.err_final,
            rcs_chunks=ba.rcs_chunks(), rcs_fill=round(ba.rcs_fill(), 4), jacobian_kernel=ba.jacobian_kernel(),
            renumbered=ba.frame_order() is not None, upload_s=round(t_up, 3))
        ba.close()
    print(json.dumps({name: out}), flush=True)

spec = sa.CONFIGS["C3_1kcam_100kpt"]
c3 = sa.config_scene("C3_1kcam_100kpt")
run("C3_frames_shuffled", sa.renumber_frames(c3, np.random.RandomState(0).permutation(c3.M)), spec.f0)
run("C3_time_order", c3, spec.f0)
run("loop_1000_frames", sa.loop_scene(sa.SceneSpec(n_frames=1000, grid_nx=200, grid_ny=100, vis_window=0), window=20), 600.0)
