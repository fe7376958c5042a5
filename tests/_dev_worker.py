"""Worker of the tests that need the DEVELOPMENT build of the library (surikatoko_amd/libsrk_ba_dev.so, -DSRK_DEV: the
fault-injection hooks srk_dbg_step_fault / srk_dbg_dp_corrupt).  tests/test_gpu_parity.py starts it as a subprocess with
SRK_BA_LIBRARY pointing at that build; exit code 0 = the assertions held."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import surikatoko_amd as sa  # noqa: E402
from surikatoko_amd import ba as B  # noqa: E402
from conftest import rel_err  # noqa: E402


def approx(a, b, rel):
    return abs(a - b) <= rel * abs(b)


def lost_hand_off_lm_loop():
    gpu = sa.BundleAdjustmentKanatani(0)
    spec = sa.SceneSpec(n_frames=400, grid_nx=60, grid_ny=40, vis_window=10, noise_uv_pix=0.2)
    sc = sa.generate_scene(spec)
    try:
        gpu.set_solver_fusion(1)
        s1 = sc.copy()
        gpu.ComputeInplace(spec.f0, s1, None, 3)
        ref = (gpu.report.iterations, gpu.report.attempts, gpu.report.err_final)
        before = gpu.solver_sync_timeouts()
        sa.lib().srk_dbg_step_fault(1)
        s2 = sc.copy()
        gpu.ComputeInplace(spec.f0, s2, None, 3)
        assert gpu.solver_sync_timeouts() == before + 1
        assert not gpu.solver_fusion()    # unfused for the rest of that call ...
        assert (gpu.report.iterations, gpu.report.attempts) == ref[:2]
        assert approx(gpu.report.err_final, ref[2], 1e-9)
        assert np.abs(s2.points - s1.points).max() < 1e-9
        assert np.abs(s2.cam_T - s1.cam_T).max() < 1e-9
        # ... and fused again from the next call on (a timeout is a scheduling event, not a property of the handle); after
        # three timeouts the unfused sequence stays until the caller asks for fusion again
        s3 = sc.copy()
        gpu.ComputeInplace(spec.f0, s3, None, 3)
        assert gpu.solver_fusion() and gpu.solver_sync_timeouts() == before + 1
        assert np.abs(s3.points - s1.points).max() < 1e-9
        for k in range(2, 5):
            sa.lib().srk_dbg_step_fault(1)
            gpu.ComputeInplace(spec.f0, sc.copy(), None, 2)
            assert gpu.solver_sync_timeouts() == before + min(k, 4)
        gpu.ComputeInplace(spec.f0, sc.copy(), None, 2)
        assert not gpu.solver_fusion() and gpu.solver_sync_timeouts() == before + 4
        gpu.set_solver_fusion(1)
        assert gpu.solver_fusion()
    finally:
        sa.lib().srk_dbg_step_fault(0)
        gpu.close()


def lost_hand_off_staged():
    gpu = sa.BundleAdjustmentKanatani(0)
    spec = sa.SceneSpec(n_frames=400, grid_nx=60, grid_ny=40, vis_window=10, noise_uv_pix=0.2)
    sc = sa.generate_scene(spec)
    try:
        gpu.set_solver_fusion(1)
        assert gpu.upload(spec.f0, sc) and gpu.rcs_chunks() >= 2
        gpu.phase_error()
        gpu.phase_derivatives()
        gpu.phase_schur(1e-3)
        assert gpu.phase_solve()
        ref = gpu.buffer(B.BUF_CORRECTIONS)[3 * sc.N:].copy()
        # (the nested solve works on copies: the SAME system again -- a second Schur sum would differ in the last bits, its
        # fp64 atomics arrive in another order)
        before = gpu.solver_sync_timeouts()
        sa.lib().srk_dbg_step_fault(1)
        assert gpu.phase_solve()                      # timed out inside, repeated unfused, succeeded
        assert gpu.solver_sync_timeouts() == before + 1
        assert rel_err(gpu.buffer(B.BUF_CORRECTIONS)[3 * sc.N:], ref) < 1e-9   # fused and unfused sequences agree to rounding
        gpu.set_solver_fusion(1)
        rng = np.random.RandomState(5)
        A = rng.randn(300, 300)
        A = A @ A.T + 300 * np.eye(300)
        b = rng.randn(300)
        sa.lib().srk_dbg_step_fault(1)
        ok, x, _ = gpu.dense_spd_solve(A, b)
        assert ok and np.abs(x - np.linalg.solve(A, b)).max() < 1e-10 * max(1.0, np.abs(x).max())
        assert gpu.solver_sync_timeouts() == before + 2
    finally:
        sa.lib().srk_dbg_step_fault(0)
        gpu.close()


def dp_selfcheck(stage):
    from oracle import oracle as orc
    spec = sa.SceneSpec(n_frames=30, grid_nx=23, grid_ny=17, vis_window=7, noise_uv_pix=0.3)
    sc = sa.generate_scene(spec)
    so = orc.Scene(sc.points, sc.cam_R, sc.cam_T, sc.K, sc.shared_k, sc.row_ptr, sc.obs_frame, sc.obs_uv)
    rc_o, rep_o = orc.compute_inplace(spec.f0, so, 1e-7, 1e6, 40)
    crit = sa.BundleAdjustmentKanataniTermCriteria()
    crit.AllowedReprojErrRelativeChange(1e-7)
    crit.MaxHessianFactor(1e6)
    ba = sa.BundleAdjustmentKanatani(0)
    try:
        ba.set_multi_schedule("dp_force")
        ba.rccl_init(ba.rccl_unique_id(), 0, 1)
        sa.lib().srk_dbg_dp_corrupt(1 if stage == "reduce" else 2)
        sg = sc.copy()
        ok = ba.ComputeInplace(spec.f0, sg, crit, 40)
        rep = ba.report
        assert ba.multi_schedule() == "allreduce (dp self-check failed)", ba.multi_schedule()
        assert "self-check" in ba.last_error()
        assert ok == (rc_o == 0) and (rep.iterations, rep.attempts) == (rep_o.iterations, rep_o.attempts)
        assert approx(rep.err_final, rep_o.err_final, 1e-6)
        assert np.abs(sg.points - so.points).max() < 1e-6 and np.abs(sg.cam_T - so.cam_T).max() < 1e-6
        # asking for the schedule again clears the verdict; the next first round checks itself again and passes
        ba.set_multi_schedule("dp_force")
        sg2 = sc.copy()
        ba.ComputeInplace(spec.f0, sg2, crit, 40)
        assert ba.multi_schedule() == "dp (self-check passed)"
        assert (ba.report.iterations, ba.report.attempts) == (rep_o.iterations, rep_o.attempts)
    finally:
        sa.lib().srk_dbg_dp_corrupt(0)
        ba.close()


if __name__ == "__main__":
    which = sys.argv[1]
    assert hasattr(sa.lib(), "srk_dbg_step_fault"), "not the development build: " + sa._lib.library_path()
    if which == "lost_hand_off_lm_loop":
        lost_hand_off_lm_loop()
    elif which == "lost_hand_off_staged":
        lost_hand_off_staged()
    elif which.startswith("dp_selfcheck_"):
        dp_selfcheck(which[len("dp_selfcheck_"):])
    else:
        raise SystemExit("unknown case " + which)
    print("ok", which)
