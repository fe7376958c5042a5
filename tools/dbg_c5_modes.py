"""development: C5 (4000 cams / 1M pts / 20M obs) -- nested dissection vs single skyline chain, one LM iteration"""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import surikatoko_amd as sa
from surikatoko_amd import ba as B
t = time.time()
spec = sa.CONFIGS["C5_4kcam_1Mpt"]
sc = sa.generate_scene(spec)
print("scene", time.time() - t, flush=True)
h = sa.BundleAdjustmentKanatani(0)
res = {}
for mode in (2, 1):
    h.set_rcs_mode(mode)
    t = time.time()
    assert h.upload(spec.f0, sc)
    print("upload", mode, time.time() - t, "chunks", h.rcs_chunks(), flush=True)
    h.optimize(None, max_iterations=2)
    r = h.report
    res[mode] = (r.iterations, r.attempts, r.err_initial, r.err_final, h.buffer(B.BUF_CORRECTIONS).copy())
    print(mode, res[mode][:4], "ms solve", r.ms_solve / max(r.attempts, 1), flush=True)
d = np.abs(res[2][4] - res[1][4]).max() / np.abs(res[1][4]).max()
print("corrections rel diff", d, "err_final rel diff", abs(res[2][3] - res[1][3]) / res[1][3])
