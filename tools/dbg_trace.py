"""SRK_DEBUG trace of a 10-iteration run on C3 (host time of every judged attempt since the iteration's derivatives)."""
import os, sys
os.environ["SRK_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import surikatoko_amd as sa
spec = sa.CONFIGS["C3_1kcam_100kpt"]; sc = sa.generate_scene(spec)
ba = sa.BundleAdjustmentKanatani(0); ba.set_profile(0)
assert ba.upload(spec.f0, sc)
ba.optimize(None, max_iterations=2); ba.reset()
print("---- timed run", file=sys.stderr)
ba.optimize(None, max_iterations=int(os.environ.get("SRK_TRACE_K", "10")))
