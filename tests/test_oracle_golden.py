"""CPU oracle vs golden vectors captured from the reference's Python prototype
(oracle/gen_golden.py; py_proto/suriko/bundle_adjustment_kanatani_impl.py)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err


def _scene(orc, g):
    return orc.Scene(g["in_points"], g["in_cam_R"], g["in_cam_T"], g["in_K"], 0, g["in_row_ptr"], g["in_obs_frame"],
                     g["in_obs_uv"])


@pytest.mark.parametrize("case", ["pyproto_case_a", "pyproto_case_b"])
def test_normalize_and_revert(orc, case):
    g = load_golden(case)
    sc = _scene(orc, g)
    ok, nrm = orc.normalize(sc)
    assert ok
    assert nrm.world_scale == pytest.approx(float(g["world_scale"]), rel=1e-13)
    assert np.abs(sc.points - g["norm_points"]).max() < 1e-13
    assert np.abs(sc.cam_R - g["norm_cam_R"]).max() < 1e-14
    assert np.abs(sc.cam_T - g["norm_cam_T"]).max() < 1e-13
    assert orc.check_normalized(sc)
    orc.revert(sc, nrm)
    assert np.abs(sc.points - g["reverted_points"]).max() < 1e-13
    assert np.abs(sc.cam_R - g["reverted_cam_R"]).max() < 1e-14
    assert np.abs(sc.cam_T - g["reverted_cam_T"]).max() < 1e-13
    # and revert restores the input (test_bundle_adjustment_kanatani.py:43-77, atol 1e-5)
    assert np.abs(sc.points - g["in_points"]).max() < 1e-12


def test_reproj_error_f0_1(orc):
    g = load_golden("pyproto_case_a")
    sc = _scene(orc, g)
    orc.normalize(sc)
    e, seen = orc.reproj_error(float(g["f0"]), sc)
    assert seen == sc.O
    assert e == pytest.approx(float(g["reproj_error"]), rel=1e-12)


@pytest.mark.parametrize("case", ["pyproto_case_a", "pyproto_case_b"])
def test_derivative_blocks(orc, case):
    g = load_golden(case)
    sc = _scene(orc, g)
    orc.normalize(sc)
    gradE, V, U, W = orc.derivatives(float(g["f0"]), sc)
    N, M = sc.N, sc.M
    assert rel_err(gradE, g["gradE"]) < 1e-11          # sums with cancellation
    assert rel_err(V.reshape(3 * N, 3), g["deriv_second_point"]) < 1e-13
    assert rel_err(U.reshape(10 * M, 10), g["deriv_second_frame"]) < 1e-13
    Wd = np.zeros((3 * N, 10 * M))
    for i in range(N):
        for o in range(sc.row_ptr[i], sc.row_ptr[i + 1]):
            j = sc.obs_frame[o]
            Wd[3 * i:3 * i + 3, 10 * j:10 * j + 10] = W[o]
    assert rel_err(Wd, g["deriv_second_pointframe"]) < 1e-13


@pytest.mark.parametrize("case", ["pyproto_case_a", "pyproto_case_b"])
@pytest.mark.parametrize("tag,c", [("c1e-4", 1e-4), ("c1e-1", 1e-1), ("c1e2", 1e2)])
def test_two_phase_corrections(orc, case, tag, c):
    """Householder QR (C++) vs LA.solve (prototype): tolerance 1e-7 relative (1e-8 for damping >= 1e-4
    on well-conditioned scenes; case b at c=1e-4 is the worst conditioned)."""
    g = load_golden(case)
    sc = _scene(orc, g)
    orc.normalize(sc)
    # use the prototype's own blocks as input so only the solve is compared
    N, M = sc.N, sc.M
    V = g["deriv_second_point"].reshape(N, 3, 3)
    U = g["deriv_second_frame"].reshape(M, 10, 10)
    W = np.zeros((sc.O, 3, 10))
    for i in range(N):
        for o in range(sc.row_ptr[i], sc.row_ptr[i + 1]):
            j = sc.obs_frame[o]
            W[o] = g["deriv_second_pointframe"][3 * i:3 * i + 3, 10 * j:10 * j + 10]
    ok, corr, S, rhs = orc.two_phase(sc, g["gradE"], V, U, W, c, want_system=True)
    assert ok
    assert rel_err(corr, g["corrections_" + tag]) < 1e-7
    # gauge gaps are exact zeros (bundle-adj-kanatani.cpp:1618-1654)
    fr = corr[3 * N:]
    assert np.all(fr[4:10] == 0) and fr[15] == 0
    # dense-literal storage gives bitwise the same numbers as block-sparse
    ok2, corr2 = orc.two_phase(sc, g["gradE"], V, U, W, c, dense_literal=True)
    assert ok2 and np.array_equal(corr, corr2)
    # two-phase vs full-Hessian solve (reference's compare_with_naive, :788-797)
    ok3, corr3 = orc.naive_solve(sc, g["gradE"], V, U, W, c)
    assert ok3 and rel_err(corr3, corr) < 1e-7
    # the reduced camera system is symmetric
    assert np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max()


def test_rodrigues_golden(orc):
    g = load_golden("pyproto_rodrigues")
    for w, R, inv in zip(g["w"], g["R"], g["se3inv"]):
        ok, Ro = orc.rot_from_axis_angle(w)
        assert ok
        assert np.abs(Ro - R).max() < 1e-15
