"""Multi-view-factorization steps: the oracle restatement against exact known answers (noise-free
two- and multi-view geometry; the reference holds no fixture for these functions, so beyond that parity is unpinned),
the host-side SO(3) projection, and -- on the GPU -- the product path against the oracle."""
import numpy as np
import pytest

import surikatoko_amd as sa
from surikatoko_amd import mvf


def _rot(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _two_views(n, seed, noise=0.0):
    rng = np.random.RandomState(seed)
    R = _rot([0.1, -0.2, 0.15])
    T = np.array([0.3, -0.1, 0.2])
    X = rng.rand(n, 3) * 2 + np.array([-1.0, -1.0, 3.0])  # anchor-frame coordinates
    x1 = X / X[:, 2:3]
    X2 = X @ R.T + T
    x2 = X2 / X2[:, 2:3] + noise * rng.randn(n, 3) * np.array([1, 1, 0])
    return R, T, x1, x2, X[:, 2].copy()


def _tracks(n_tracks, n_frames, seed):
    """random world points seen from a ring of cameras; every track starts at a random frame and skips some frames."""
    rng = np.random.RandomState(seed)
    cam_R, cam_T = [], []
    for j in range(n_frames):
        R = _rot([0.02 * j + 1e-3, -0.03 * j, 0.01 * j])
        cam_R.append(R.reshape(9))
        cam_T.append(np.array([0.1 * j, -0.05 * j, 0.02 * j]))
    cam_R, cam_T = np.array(cam_R), np.array(cam_T)
    X = rng.rand(n_tracks, 3) * 2 + np.array([-1.0, -1.0, 4.0])
    row_ptr, frame, xm, depth = [0], [], [], []
    for i in range(n_tracks):
        start = rng.randint(0, n_frames - 2)
        fr = [start] + [j for j in range(start + 1, n_frames) if rng.rand() < 0.7]
        if len(fr) < 2:
            fr.append(n_frames - 1)
        for j in fr:
            xc = cam_R[j].reshape(3, 3) @ X[i] + cam_T[j]
            frame.append(j)
            xm.append(xc / xc[2])
        depth.append((cam_R[start].reshape(3, 3) @ X[i] + cam_T[start])[2])
        row_ptr.append(len(frame))
    return np.array(row_ptr), np.array(frame, dtype=np.int32), np.array(xm), cam_R, cam_T, np.array(depth)


# ------------------------------------------------------------------ CPU: oracle known answers, host projection

def test_oracle_relative_motion_recovers_exact_geometry(orc):
    R, T, x1, x2, d1 = _two_views(12, 0)
    ok, Re, Te = orc.mvf_relative_motion(x1, x2, d1)
    assert ok and np.abs(Re - R).max() < 1e-12 and np.abs(Te - T).max() < 1e-12


def test_oracle_point_depth_recovers_exact_depth(orc):
    rp, fr, xm, cam_R, cam_T, depth = _tracks(20, 7, 1)
    for i in range(20):
        lo, hi = rp[i], rp[i + 1]
        assert orc.mvf_point_depth(fr[lo:hi], xm[lo:hi], cam_R, cam_T) == pytest.approx(depth[i], rel=1e-11)


@pytest.mark.parametrize("scale", [1.7, -1.7, 0.3])
def test_so3_projection_host_and_oracle(orc, scale):
    """MASKS 8.41/8.42: a scaled (even reflected) rotation projects back onto the rotation, T is rescaled alike."""
    R, T = _rot([0.3, 0.2, -0.4]), np.array([0.5, -0.2, 0.1])
    for f in (orc.project_onto_so3, mvf.project_onto_so3):
        ok, Rp, Tp = f(scale * R, scale * T)
        assert ok and np.abs(Rp - R).max() < 1e-13 and np.abs(Tp - T).max() < 1e-13
    rng = np.random.RandomState(5)
    Rn = R + 0.05 * rng.randn(3, 3)
    ok1, R1, T1 = orc.project_onto_so3(Rn, T)
    ok2, R2, T2 = mvf.project_onto_so3(Rn, T)
    assert ok1 and ok2 and np.abs(R1 - R2).max() < 1e-12 and np.abs(T1 - T2).max() < 1e-12
    assert np.abs(R2 @ R2.T - np.eye(3)).max() < 1e-13 and np.linalg.det(R2) == pytest.approx(1.0, abs=1e-13)
    assert not mvf.project_onto_so3(np.zeros((3, 3)), T)[0] and not orc.project_onto_so3(np.zeros((3, 3)), T)[0]


# ------------------------------------------------------------------ GPU: product path vs oracle

@pytest.fixture(scope="module")
def gpu():
    h = sa.BundleAdjustmentKanatani(0)
    yield h
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,noise", [(6, 0.0), (12, 0.0), (700, 0.0), (300, 1e-3)])
def test_relative_motion_matches_oracle(orc, gpu, n, noise):
    R, T, x1, x2, d1 = _two_views(n, n, noise)
    ok_o, Ro, To = orc.mvf_relative_motion(x1, x2, d1)
    ok_g, Rg, Tg = mvf.relative_motion(gpu, x1, x2, d1)
    assert ok_g and ok_o
    # eigenvector of A^T A (product) vs singular vector of A (oracle): the gap to the next singular value sets the
    # attainable agreement; exact data have sigma_min = 0
    tol = 1e-9 if noise == 0 else 1e-7
    assert np.abs(Rg - Ro).max() < tol and np.abs(Tg - To).max() < tol
    if noise == 0:
        assert np.abs(Rg - R).max() < 1e-9 and np.abs(Tg - T).max() < 1e-9
    assert np.abs(Rg @ Rg.T - np.eye(3)).max() < 1e-12


@pytest.mark.gpu
def test_estimate_depths_matches_oracle(orc, gpu):
    rp, fr, xm, cam_R, cam_T, depth = _tracks(1000, 9, 2)
    got = mvf.estimate_depths(gpu, rp, fr, xm, cam_R, cam_T)
    want = np.array([orc.mvf_point_depth(fr[rp[i]:rp[i + 1]], xm[rp[i]:rp[i + 1]], cam_R, cam_T) for i in range(1000)])
    assert np.abs(got - want).max() < 1e-11 * np.abs(want).max()
    assert np.abs(got - depth).max() < 1e-9
    # a track seen once has no depth
    rp1 = np.array([0, 1, 3])
    got1 = mvf.estimate_depths(gpu, rp1, fr[:3], xm[:3], cam_R, cam_T)
    assert np.isnan(got1[0]) and np.isfinite(got1[1])


@pytest.mark.gpu
def test_mvf_argument_errors(gpu):
    R, T, x1, x2, d1 = _two_views(5, 0)
    with pytest.raises(ValueError):
        mvf.relative_motion(gpu, x1, x2, d1)  # fewer than 6 points: the null vector is not unique
    rp, fr, xm, cam_R, cam_T, _ = _tracks(5, 4, 3)
    bad = fr.copy()
    bad[0] = 99
    with pytest.raises(ValueError):
        mvf.estimate_depths(gpu, rp, bad, xm, cam_R, cam_T)
