"""the reference's finite-difference derivative checkers (bundle-adj-kanatani.cpp:895-1138, patches
:336-394, AddDeltaToFrameInplace :94-120), restated in the oracle as a test utility, against the closed-form
derivatives (:1140-1549) -- of the oracle here, of the HIP path in the `gpu` test at the bottom.

The reference compares with rough_rtol = 0.2 and only logs.  What the comparison can and cannot show:
* first derivatives (formula 8) are exact, so finite differences agree to their own truncation error;
* second derivatives (formula 9) are the Gauss-Newton products of first derivatives: they equal the true second
  derivatives only where the residuals vanish, i.e. on a noise-free scene;
* the frame derivatives embed f0 as if K(2,2) = f0 (:1476-1513, the paper's convention).  With the demos' convention
  (K pre-divided by f0, K(2,2) = 1, f0 = 600) the closed forms are NOT the derivatives of the error any more -- the
  reference computes them like that and so does every implementation here; the checker makes it visible.
"""
import numpy as np
import pytest

import surikatoko_amd as sa
from conftest import rel_err


def _oscene(orc, sc, K=None):
    return orc.Scene(sc.points, sc.cam_R, sc.cam_T, sc.K if K is None else K, 0, sc.row_ptr, sc.obs_frame, sc.obs_uv)


def _frame_grad(g, N, j):
    return g[3 * N + 10 * j:3 * N + 10 * j + 10]


def test_first_derivatives_match_finite_differences_f0_1(orc):
    spec = sa.SceneSpec(n_frames=6, grid_nx=5, grid_ny=4, vis_window=4, f0=1.0)   # noisy: non-zero residuals
    so = _oscene(orc, sa.generate_scene(spec))
    g, V, U, W = orc.derivatives(1.0, so)
    for pi in (0, 3, 11, 19):
        d1, _ = orc.fd_point(1.0, so, pi, 1e-5)
        assert rel_err(d1, g[3 * pi:3 * pi + 3]) < 1e-7
    for fj in range(so.M):
        d1, _ = orc.fd_frame(1.0, so, fj, 1e-6)
        assert rel_err(d1, _frame_grad(g, so.N, fj)) < 1e-6


def test_second_derivatives_match_finite_differences_on_a_noise_free_scene(orc):
    spec = sa.SceneSpec(n_frames=6, grid_nx=5, grid_ny=4, vis_window=4, f0=1.0, noise_x3d_hi=0.0, noise_r_hi=0.0)
    so = _oscene(orc, sa.generate_scene(spec))
    e, _ = orc.reproj_error(1.0, so)
    assert e < 1e-20
    g, V, U, W = orc.derivatives(1.0, so)
    for pi in (0, 7, 19):
        _, d2 = orc.fd_point(1.0, so, pi, 1e-4)
        assert rel_err(d2, V[pi]) < 1e-7
    for fj in (0, 2, 5):
        _, d2 = orc.fd_frame(1.0, so, fj, 1e-4)
        assert rel_err(d2, U[fj]) < 1e-6
    # a mixed block: landmark 7 in each of its frames; in a frame that does not see it the derivative is zero
    o0, o1 = so.row_ptr[7], so.row_ptr[8]
    for o in range(o0, o1):
        d2 = orc.fd_point_frame(1.0, so, 7, int(so.obs_frame[o]), 1e-4)
        assert rel_err(d2, W[o]) < 1e-6
    unseen = [j for j in range(so.M) if j not in set(so.obs_frame[o0:o1])]
    assert unseen
    assert np.abs(orc.fd_point_frame(1.0, so, 7, unseen[0], 1e-4)).max() < 1e-6 * np.abs(W[o0]).max()


def test_gauss_newton_blocks_differ_from_true_second_derivatives_with_residuals(orc):
    """Formula 9 drops the residual x second-derivative term: with noise the checker reports a difference (which is why
    the reference only logs it)."""
    spec = sa.SceneSpec(n_frames=6, grid_nx=5, grid_ny=4, vis_window=4, f0=1.0, noise_x3d_hi=0.02, noise_r_hi=0.02)
    so = _oscene(orc, sa.generate_scene(spec))
    g, V, U, W = orc.derivatives(1.0, so)
    _, d2 = orc.fd_frame(1.0, so, 2, 1e-4)
    assert 1e-4 < rel_err(d2, U[2]) < 0.2       # different, yet within the reference's rough_rtol


def test_frame_derivatives_assume_k22_equals_f0(orc):
    """Demo convention (K / f0 with K(2,2) = 1, f0 = 600): the closed-form frame derivatives are not the derivatives of
    the error (u0, v0 entries are off by exactly 1 / f0); with K scaled so that K(2,2) = f0 (same projections, the
    paper's and the prototype's convention) they are.  Landmark derivatives do not involve f0 and agree either way."""
    f0 = 600.0
    spec = sa.SceneSpec(n_frames=6, grid_nx=5, grid_ny=4, vis_window=4, f0=f0)
    sc = sa.generate_scene(spec)
    assert sc.K[0, 8] == 1.0
    demo = _oscene(orc, sc)
    paper = _oscene(orc, sc, K=sc.K * f0)
    e_demo, _ = orc.reproj_error(f0, demo)
    e_paper, _ = orc.reproj_error(f0, paper)
    assert e_paper == pytest.approx(e_demo, rel=1e-12)
    g_demo = orc.derivatives(f0, demo)[0]
    g_paper = orc.derivatives(f0, paper)[0]
    d1p, _ = orc.fd_point(f0, demo, 3, 1e-5)
    assert rel_err(d1p, g_demo[9:12]) < 1e-7
    fd_demo, _ = orc.fd_frame(f0, demo, 2, 1e-6)
    fd_paper, _ = orc.fd_frame(f0, paper, 2, 1e-4)
    assert rel_err(fd_paper, _frame_grad(g_paper, demo.N, 2)) < 1e-5
    cl = _frame_grad(g_demo, demo.N, 2)
    assert rel_err(fd_demo, cl) > 0.5                                        # not the true gradient ...
    assert cl[2] / fd_demo[2] == pytest.approx(1 / f0, rel=1e-4)             # ... u0, v0 off by 1 / f0
    assert cl[3] / fd_demo[3] == pytest.approx(1 / f0, rel=1e-4)


@pytest.mark.gpu
def test_hip_derivatives_against_the_finite_difference_checker(orc):
    """The HIP path's gradient and blocks against the finite-difference checker directly (f0 = 1; noise-free scene for
    the second derivatives), both derivative kernels."""
    from surikatoko_amd import ba as B
    gpu = sa.BundleAdjustmentKanatani(0)
    try:
        for mode in (0, 1):
            gpu.set_jacobian_mode(mode)
            for noise in (0.005, 0.0):
                spec = sa.SceneSpec(n_frames=6, grid_nx=5, grid_ny=4, vis_window=4, f0=1.0, noise_x3d_hi=noise, noise_r_hi=noise)
                sc = sa.generate_scene(spec)
                so = _oscene(orc, sc)
                ok, _ = orc.normalize(so)
                assert ok
                scn = sa.Scene(so.points, so.cam_R, so.cam_T, so.K, 0, so.row_ptr, so.obs_frame, so.obs_uv)
                assert gpu.upload(1.0, scn, already_normalized=True)
                assert gpu.jacobian_kernel() == (2 if mode == 1 else 1)
                gpu.phase_derivatives()
                g = gpu.buffer(B.BUF_GRAD)
                V = gpu.buffer(B.BUF_POINT_BLOCKS).reshape(-1, 3, 3)
                U = gpu.buffer(B.BUF_FRAME_BLOCKS).reshape(-1, 10, 10)
                W = gpu.buffer(B.BUF_POINT_FRAME).reshape(-1, 3, 10)
                if noise > 0:
                    for pi in (0, 11):
                        assert rel_err(orc.fd_point(1.0, so, pi, 1e-5)[0], g[3 * pi:3 * pi + 3]) < 1e-6
                    for fj in (1, 4):
                        assert rel_err(orc.fd_frame(1.0, so, fj, 1e-6)[0], _frame_grad(g, so.N, fj)) < 1e-5
                else:
                    assert rel_err(orc.fd_point(1.0, so, 7, 1e-4)[1], V[7]) < 1e-6
                    assert rel_err(orc.fd_frame(1.0, so, 3, 1e-4)[1], U[3]) < 1e-5
                    o = int(so.row_ptr[7]) + 1
                    assert rel_err(orc.fd_point_frame(1.0, so, 7, int(so.obs_frame[o]), 1e-4), W[o]) < 1e-5
    finally:
        gpu.close()
