"""One-call latency of config 1 (dino stand-in, the dino flagfile's threshold), fused and unfused solve, with and without
speculative attempts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import surikatoko_amd as sa, torch
crit = sa.BundleAdjustmentKanataniTermCriteria(); crit.AllowedReprojErrRelativeChange(4.56e-8)
c1 = sa.config_scene("C1_dino_standin")
for fused in (1, 0):
    for spec in (1, 0):
        ba = sa.BundleAdjustmentKanatani(0); ba.set_solver_fusion(fused); ba.set_speculation(spec)
        for rep in range(3):
            sg = c1.copy(); torch.cuda.synchronize(); t0 = time.perf_counter()
            ok = ba.ComputeInplace(600.0, sg, crit, 0); dt = 1e3 * (time.perf_counter() - t0)
            r = ba.report
            print(f"fused {fused} speculation {spec} call {rep}: {dt:.2f} ms, LM loop {r.ms_total:.2f} ms, {r.iterations} it / {r.attempts} attempts, timeouts {ba.solver_sync_timeouts()}")
        ba.close()
