"""Stress of the in-launch hand-offs of k_step256: the same reduced camera system (C3, nested chunks) solved over and over
with the fused outer step, every solution compared bit for bit with the FIRST fused solution (the solver has a fixed summation
order: a stale read of a handed-off tile would differ) and to rounding with the unfused sequence's -- alone on the chip, and
beside a stream of HBM-heavy torch kernels on another stream (uneven load; consumer caches warm from the previous solve)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import surikatoko_amd as sa
from surikatoko_amd import ba as B
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 300
spec = sa.CONFIGS["C3_1kcam_100kpt"]; sc = sa.generate_scene(spec)
ba = sa.BundleAdjustmentKanatani(0); ba.set_speculation(False)
assert ba.upload(spec.f0, sc)
ba.phase_derivatives(); ba.phase_schur(1e-3)
ba.set_solver_fusion(0); assert ba.phase_solve(); unfused = ba.buffer(B.BUF_CORRECTIONS)[3 * sc.N:].copy()
ba.set_solver_fusion(1); assert ba.phase_solve(); ref = ba.buffer(B.BUF_CORRECTIONS)[3 * sc.N:].copy()
assert np.abs(ref - unfused).max() < 1e-9 * np.abs(unfused).max()
side = torch.cuda.Stream()
x = torch.empty(64 * 1024 * 1024, device="cuda", dtype=torch.float64); y = torch.empty_like(x)
for load in (False, True):
    bad = 0; t0 = time.perf_counter()
    for it in range(n_rep):
        if load:
            with torch.cuda.stream(side):
                for _ in range(2): y.copy_(x); x.add_(1.0)
        assert ba.phase_solve()
        got = ba.buffer(B.BUF_CORRECTIONS)[3 * sc.N:]
        if not np.array_equal(got, ref): bad += 1
    torch.cuda.synchronize()
    print(f"side load {load}: {n_rep} fused solves, {bad} differ from the first fused solution, hand-off timeouts {ba.solver_sync_timeouts()}, "
          f"{1e3 * (time.perf_counter() - t0) / n_rep:.2f} ms per solve + download")
    assert bad == 0 and ba.solver_sync_timeouts() == 0
print("ok")
