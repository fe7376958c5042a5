"""The C-ABI library loads and exports every symbol include/srk_ba.h declares; host-only entry points
(normalisation, scene generator) agree with the oracle and the reference's known answers.  No GPU needed."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import surikatoko_amd as sa
from surikatoko_amd import _lib
from conftest import ROOT, load_golden


def test_library_loads_and_exports_header_symbols():
    L = sa.lib()
    header = open(os.path.join(ROOT, "include", "srk_ba.h")).read()
    declared = set(re.findall(r"\b(srk_[a-z0-9_]+)\s*\(", header))
    declared -= {"srk_allreduce_fn"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/srk_ba.h but not exported"
    assert set(_lib.EXPORTS) == declared


def test_status_strings_are_the_reference_strings():
    # bundle-adj-kanatani.cpp:751,866,868,882
    assert sa.status_string(1) == "abs err threshold"
    assert sa.status_string(2) == "small relative err change"
    assert sa.status_string(3) == "hessian overflow"
    assert sa.status_string(4) == "err converged to limit value"
    assert sa.status_string(0) == ""


def test_no_gpu_fails_loudly():
    if sa.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        sa.BundleAdjustmentKanatani()


def _scene_from_golden(g):
    return sa.Scene(g["in_points"], g["in_cam_R"], g["in_cam_T"], g["in_K"], 0, g["in_row_ptr"], g["in_obs_frame"],
                    g["in_obs_uv"])


@pytest.mark.parametrize("case", ["pyproto_case_a", "pyproto_case_b"])
def test_host_normalisation_vs_python_prototype(case):
    g = load_golden(case)
    sc = _scene_from_golden(g)
    ok, nrm = sa.normalize_scene_inplace(sc)
    assert ok
    assert nrm.world_scale == pytest.approx(float(g["world_scale"]), rel=1e-13)
    assert np.abs(sc.points - g["norm_points"]).max() < 1e-13
    assert np.abs(sc.cam_R - g["norm_cam_R"]).max() < 1e-14
    assert np.abs(sc.cam_T - g["norm_cam_T"]).max() < 1e-13
    assert sa.check_world_is_normalized(sc)
    from surikatoko_amd.ba import revert_normalization
    revert_normalization(sc, nrm)
    assert np.abs(sc.points - g["in_points"]).max() < 1e-12
    assert np.abs(sc.cam_R - g["in_cam_R"]).max() < 1e-13
    assert np.abs(sc.cam_T - g["in_cam_T"]).max() < 1e-12


def test_normalization_simple_known_answers():
    """cpp_impl/suriko-test/test-bundle-adj-kanatani.cpp:22-128 through the product's host code."""
    atol = 1e-2
    L = sa.lib()
    angles = np.array([3 * math.pi / 2 + math.pi / 6, 3 * math.pi / 2])
    R = np.zeros((2, 9))
    T = np.zeros((2, 3))
    center = np.zeros(3)
    L.srk_circle_camera_shots(center.ctypes.data_as(C.c_void_p), C.c_double(1.0), C.c_double(0.0), C.c_int32(2),
                              angles.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p),
                              T.ctypes.data_as(C.c_void_p))
    pts = np.array([[-1, 0, 0], [-0.5, 0.866, 0], [0, 1, 0], [1, 0, 0], [0, -1, 0]], dtype=np.float64)
    sc = sa.Scene(pts, R, T, np.eye(3).reshape(1, 9), 1, np.arange(6), np.zeros(5, dtype=np.int32), np.zeros((5, 2)))
    before = sc.copy()
    ok, nrm = sa.normalize_scene_inplace(sc, 1.0, 0)
    assert ok
    assert np.linalg.norm(sc.cam_T[0]) < atol
    assert np.abs(sc.cam_R[0].reshape(3, 3) - np.eye(3)).max() < atol
    R1 = sc.cam_R[1].reshape(3, 3)
    assert abs(abs((-R1.T @ sc.cam_T[1])[0]) - 1.0) < 0.01
    s = nrm.world_scale
    cam0 = np.array([[-0.866, 0, 1.5], [0, 0, 2], [0.5, 0, 1.866], [0.866, 0, 0.5], [-0.5, 0, 0.133975]]) * s
    assert np.linalg.norm(cam0 - sc.points, axis=1).max() < atol
    cam1 = np.array([[-1, 0, 1], [-0.5, 0, 1.866], [0, 0, 2], [1, 0, 1], [0, 0, 0]], dtype=np.float64) * s
    got = (R1 @ sc.points.T).T + sc.cam_T[1]
    assert np.linalg.norm(cam1 - got, axis=1).max() < atol
    from surikatoko_amd.ba import revert_normalization
    revert_normalization(sc, nrm)
    assert np.abs(sc.points - before.points).max() < atol
    assert np.abs(sc.cam_R - before.cam_R).max() < atol
    assert np.abs(sc.cam_T - before.cam_T).max() < atol


def test_circle_camera_shots_vs_oracle(orc):
    L = sa.lib()
    angles = np.linspace(-1.0, 2.0, 7)
    R = np.zeros((7, 9))
    T = np.zeros((7, 3))
    center = np.array([1.0, 0.5, 0.0])
    L.srk_circle_camera_shots(center.ctypes.data_as(C.c_void_p), C.c_double(7.5), C.c_double(5.0), C.c_int32(7),
                              angles.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p),
                              T.ctypes.data_as(C.c_void_p))
    Ro, To = orc.circle_camera_shots(center, 7.5, 5.0, angles)
    assert np.abs(R - Ro).max() < 1e-15 and np.abs(T - To).max() < 1e-14


def test_scene_generator_shapes_and_gauge():
    spec = sa.SceneSpec(n_frames=12, grid_nx=7, grid_ny=5, vis_window=4, noise_uv_pix=0.0)
    sc, pts_gt, Rg, Tg = sa.generate_scene(spec, with_gt=True)
    assert sc.N == 35 and sc.M == 12 and sc.O == 35 * 4
    assert np.all(np.diff(sc.row_ptr) == 4)
    for i in range(sc.N):
        f = sc.obs_frame[sc.row_ptr[i]:sc.row_ptr[i + 1]]
        assert np.all(np.diff(f) == 1) and f[0] == (i * 2654435761 % 2**32) % (12 - 4 + 1)
    # noise recipe: uniform [hi/2, hi] per axis (demo-bundle-adj-circle-grid.cpp:116-127)
    dlt = sc.points - pts_gt
    assert dlt.min() >= 0.0025 and dlt.max() <= 0.005
    # pixels are exact projections of the ground truth (demo :196-207)
    j = int(sc.obs_frame[0])
    xc = Rg[j].reshape(3, 3) @ pts_gt[0] + Tg[j]
    K = sc.K[j].reshape(3, 3)
    uv = (K @ (xc / xc[2]))[:2] * 600.0
    assert np.abs(uv - sc.obs_uv[0]).max() < 1e-9
    # determinism (mt19937 seed 1234)
    sc2 = sa.generate_scene(spec)
    assert np.array_equal(sc.points, sc2.points) and np.array_equal(sc.cam_R, sc2.cam_R)
    # the two-ring rig keeps the gauge scale O(1)
    ok, nrm = sa.normalize_scene_inplace(sc.copy())
    assert ok and 0.1 < nrm.world_scale < 10


def test_shard_bounds_cover_and_balance():
    from surikatoko_amd.ba import shard_bounds
    rng = np.random.RandomState(0)
    counts = rng.randint(1, 30, size=1000)
    row_ptr = np.concatenate([[0], np.cumsum(counts)])
    for world in (1, 2, 3, 8):
        cuts = [shard_bounds(row_ptr, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == 1000
        for a, b in zip(cuts[:-1], cuts[1:]):
            assert a[1] == b[0]
        obs = [row_ptr[hi] - row_ptr[lo] for lo, hi in cuts]
        assert max(obs) - min(obs) <= 2 * counts.max()
