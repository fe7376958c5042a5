"""GPU box: host + device time of srk_ba_upload_scene by stage (SRK_DEBUG trace), first and second upload of a handle."""
import os, sys, time
os.environ["SRK_DEBUG"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import surikatoko_amd as sa
name = sys.argv[1] if len(sys.argv) > 1 else "C3_1kcam_100kpt"
if name == "all_visible_60":
    spec = sa.SceneSpec(n_frames=60, grid_nx=81, grid_ny=41, vis_window=0); sc = sa.generate_scene(spec); f0 = spec.f0
else:
    sc = sa.config_scene(name); f0 = sa.CONFIGS[name].f0
ba = sa.BundleAdjustmentKanatani(0)
for k in range(3):
    t = time.perf_counter(); assert ba.upload(f0, sc); print("upload", k, "%.1f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr, flush=True)
