#!/usr/bin/env python3
"""Generate golden vectors for the BA hot path by importing the reference's Python prototype.

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Writes tests/golden/pyproto_*.npz (inputs + the prototype's outputs).  Only data is
committed; no reference source or bytecode travels.  The prototype functions used:
  py_proto/suriko/bundle_adjustment_kanatani_impl.py
     NormalizeWorldInplace / WorldNormalizer                      :63-145
     BundleAdjustmentKanataniReprojError                          :360-425
     BundleAdjustmentKanatani.__ComputeDerivativesCloseForm       :760-1005
     BundleAdjustmentKanatani.__EstimateCorrectionsDecomposedInTwoPhases :1748-1877
  py_proto/suriko/obs_geom.py  RotMatFromAxisAngle :234, SE3Inv :75
The prototype reads f0 from K[2,2] (:777,829), so it equals the C++ formulas when the
caller passes K with K[2,2] == f0 (case A: f0 = 1, case B: f0 = 600).
"""
import os
import sys

import numpy as np

REF = "/root/reference/py_proto"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from suriko import bundle_adjustment_kanatani_impl as ba  # noqa: E402
from suriko import obs_geom  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class Track:
    """Stand-in for mvg.PointLife (mvg.py:3400-3413): the three fields the BA path reads."""

    def __init__(self, track_id, n_frames):
        self.track_id = track_id
        self.virtual_feat_id = track_id
        self.points_list_pixel = [None] * n_frames


def look_at_cams(M, rng):
    """M cameras on an arc looking roughly at the origin (plain numpy, input data only)."""
    Rs, Ts = [], []
    for k in range(M):
        ang = -1.0 + 2.2 * k / max(M - 1, 1)
        pos = np.array([6.0 * np.cos(ang), 6.0 * np.sin(ang), 4.0 - 1.5 * min(k, 1) - 0.1 * k])
        fwd = -pos / np.linalg.norm(pos)
        up = np.array([0.0, 0.0, 1.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.vstack([right, down, fwd])  # world -> cam
        w = rng.uniform(-0.02, 0.02, 3)
        ok, dR = obs_geom.RotMatFromAxisAngle(w)
        if ok:
            R = dR.dot(R)
        T = -R.dot(pos)
        Rs.append(R)
        Ts.append(T)
    return Rs, Ts


def build_case(seed, M, N, f0, k22):
    rng = np.random.RandomState(seed)
    pts_true = rng.uniform(-1.0, 1.0, (N, 3))
    pts_true[:, 2] = rng.uniform(0.0, 1.0, N)
    Rs, Ts = look_at_cams(M, rng)
    Ks = []
    for k in range(M):
        K = np.array([[880.0 / 600 * k22 * (1 + 0.01 * k), 0, 400.0 / 600 * k22],
                      [0, 660.0 / 600 * k22, 300.0 / 600 * k22 * (1 - 0.01 * k)],
                      [0, 0, k22]])
        Ks.append(K)
    tracks = []
    for i in range(N):
        tr = Track(i, M)
        start = rng.randint(0, M - 1)
        length = rng.randint(2, M + 1)
        for j in range(start, min(M, start + length)):
            if rng.rand() < 0.15 and j not in (start, min(M, start + length) - 1):
                continue  # gap inside a track
            xc = Rs[j].dot(pts_true[i]) + Ts[j]
            xi = Ks[j].dot(xc)
            uv = xi[0:2] / xi[2] * f0  # pixels (C++ convention: error uses uv / f0)
            uv = uv + rng.normal(0, 0.3 * f0 / 600.0, 2)
            tr.points_list_pixel[j] = uv
        tracks.append(tr)
    # every point needs >= 1 observation
    for tr in tracks:
        assert any(p is not None for p in tr.points_list_pixel)
    pts_noisy = [pts_true[i] + rng.uniform(0.0025, 0.005, 3) for i in range(N)]
    return tracks, pts_noisy, Rs, Ts, Ks


def flat_scene(tracks, pts, Rs, Ts, Ks):
    N, M = len(pts), len(Rs)
    row_ptr = [0]
    frames, uvs = [], []
    for tr in tracks:
        for j, p in enumerate(tr.points_list_pixel):
            if p is not None:
                frames.append(j)
                uvs.append(p)
        row_ptr.append(len(frames))
    return dict(points=np.array(pts, dtype=np.float64).reshape(N, 3),
                cam_R=np.array(Rs, dtype=np.float64).reshape(M, 9),
                cam_T=np.array(Ts, dtype=np.float64).reshape(M, 3),
                K=np.array(Ks, dtype=np.float64).reshape(M, 9),
                row_ptr=np.array(row_ptr, dtype=np.int64),
                obs_frame=np.array(frames, dtype=np.int32),
                obs_uv=np.array(uvs, dtype=np.float64).reshape(-1, 2))


def run_case(name, seed, M, N, f0, k22, with_error):
    tracks, pts, Rs, Ts, Ks = build_case(seed, M, N, f0, k22)
    out = {}
    for k, v in flat_scene(tracks, pts, Rs, Ts, Ks).items():
        out["in_" + k] = v
    out["f0"] = np.float64(f0)

    world_pnts = [p.copy() for p in pts]
    rts = [(R.copy(), T.copy()) for R, T in zip(Rs, Ts)]
    ids = list(range(N))
    comp = 1
    nrm = ba.NormalizeWorldInplace(world_pnts, rts, ids, 1.0, comp)
    out["world_scale"] = np.float64(nrm.world_scale)
    assert nrm.world_scale > 0, "pick a scene with T01[y] > 0 (C++ takes abs(), the prototype does not)"
    out["norm_points"] = np.array(world_pnts).reshape(N, 3)
    out["norm_cam_R"] = np.array([rt[0] for rt in rts]).reshape(M, 9)
    out["norm_cam_T"] = np.array([rt[1] for rt in rts]).reshape(M, 3)

    if with_error:  # consistent with the C++ only when f0 == 1 (the prototype does not divide uv by f0)
        out["reproj_error"] = np.float64(ba.BundleAdjustmentKanataniReprojError(
            tracks, world_pnts, rts, ids, cam_mat_pixel_from_meter_list=Ks))

    obj = ba.BundleAdjustmentKanatani(debug=0)
    obj.points_life = tracks
    obj.bundle_pnt_ids = ids
    obj.variable_intrinsics = True
    obj.same_focal_length_xy = False
    obj.cam_mat_pixel_from_meter = None
    obj.cam_mat_pixel_from_meter_list = Ks
    obj.world_pnts = world_pnts
    obj.framei_from_world_RT_list = rts
    obj.elem_type = np.float64
    obj.POINT_VARS = 3
    obj.FRAME_VARS = 10
    obj.INTRINSICS_VARS = 4
    obj.unity_comp_ind = comp
    obj._BundleAdjustmentKanatani__UpdateNormalizePattern()

    gradE = np.zeros(3 * N + 10 * M)
    gradE2 = np.zeros(3 * N + 10 * M)
    d2p = np.zeros((3 * N, 3))
    d2f = np.zeros((10 * M, 10))
    d2pf = np.zeros((3 * N, 10 * M))
    obj._BundleAdjustmentKanatani__ComputeDerivativesCloseForm(N, M, False, gradE, gradE2, d2p, d2f, d2pf)
    out["gradE"] = gradE
    out["deriv_second_point"] = d2p
    out["deriv_second_frame"] = d2f
    out["deriv_second_pointframe"] = d2pf

    n = 10 * M - 7
    for tag, c in (("c1e-4", 1e-4), ("c1e-1", 1e-1), ("c1e2", 1e2)):
        matG = np.zeros((n, n))
        left = np.zeros((n, n))
        right = np.zeros(n)
        corr = np.zeros(3 * N + 10 * M)
        obj._BundleAdjustmentKanatani__EstimateCorrectionsDecomposedInTwoPhases(
            N, M, c, gradE, d2p, d2f, d2pf, matG, left, right, corr, None)
        out["corrections_" + tag] = corr
        out["matG_" + tag] = matG
    # revert round trip
    nrm.RevertNormalization()
    out["reverted_points"] = np.array(world_pnts).reshape(N, 3)
    out["reverted_cam_R"] = np.array([rt[0] for rt in rts]).reshape(M, 9)
    out["reverted_cam_T"] = np.array([rt[1] for rt in rts]).reshape(M, 3)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("in_")})


def rodrigues_case():
    rng = np.random.RandomState(7)
    ws = [np.array([1.0, 1.0, 1.0]) * (2 * np.pi / 3) / np.sqrt(3.0), np.array([1.0, 1.0, 1.0]) * (np.pi / 4) / np.sqrt(3.0)]
    ws += [rng.uniform(-1, 1, 3) * s for s in (1e-6, 1e-3, 0.1, 1.0, 3.0)]
    Rs, invs = [], []
    for w in ws:
        ok, R = obs_geom.RotMatFromAxisAngle(w)
        assert ok
        Rs.append(R)
        Ri, Ti = obs_geom.SE3Inv((R, w))
        invs.append(np.hstack([Ri.reshape(9), Ti]))
    np.savez_compressed(os.path.join(OUT, "pyproto_rodrigues.npz"), w=np.array(ws), R=np.array(Rs), se3inv=np.array(invs))
    print("wrote rodrigues")




def dino_prep_case():
    """DecomposeProjMat / Triangulate3DPointByLeastSquares of py_proto/suriko/obs_geom.py:149-199, 430-456
    (the dino demo's pre-processing)."""
    rng = np.random.RandomState(5)
    Rs, Ts = look_at_cams(7, rng)
    Ps, scales, Ks, Rd, Td = [], [], [], [], []
    for k, (R, T) in enumerate(zip(Rs, Ts)):
        K = np.array([[880.0 * (1 + 0.02 * k), 3.0 * k, 400.0 - 5 * k], [0, 660.0, 300.0 + 4 * k], [0, 0, 1.0]])
        P = (1.5 - 0.7 * k) * K.dot(np.hstack([R, T.reshape(3, 1)]))  # both signs of the scale
        sc, Kd, (Rdir, tdir) = obs_geom.DecomposeProjMat(P)
        Ps.append(P)
        scales.append(sc)
        Ks.append(Kd)
        Rd.append(Rdir)
        Td.append(tdir)
    pts = rng.uniform(-1, 1, (12, 3))
    f0 = 600.0
    Pn = [np.array(P) / np.linalg.norm(np.array(P)[2, 0:3]) for P in Ps]
    tri_uv, tri_P, tri_X, tri_n = [], [], [], []
    for X in pts:
        n = rng.randint(2, 8)
        idx = rng.choice(7, n, replace=False)
        uv, PP = [], []
        for j in idx:
            x = Pn[j].dot(np.hstack([X, 1.0]))
            uv.append(x[0:2] / x[2] * 1.0 + rng.normal(0, 0.5, 2))
            # the demo triangulates with f0-scaled matrices: rows 0,1 divided by f0
            Pj = Pn[j].copy()
            Pj[0:2, :] /= f0
            PP.append(Pj)
        Xt = obs_geom.Triangulate3DPointByLeastSquares(uv, PP, f0, 0)
        tri_n.append(n)
        tri_uv.append(np.vstack(uv + [np.zeros(2)] * (7 - n)))
        tri_P.append(np.array(PP + [np.zeros((3, 4))] * (7 - n)))
        tri_X.append(Xt)
    np.savez_compressed(os.path.join(OUT, "pyproto_dino_prep.npz"), P=np.array(Ps), scale=np.array(scales),
                        K=np.array(Ks), R_direct=np.array(Rd), T_direct=np.array(Td), f0=np.float64(f0),
                        tri_n=np.array(tri_n), tri_uv=np.array(tri_uv), tri_P=np.array(tri_P), tri_X=np.array(tri_X))
    print("wrote dino prep")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    run_case("pyproto_case_a", seed=11, M=6, N=40, f0=1.0, k22=1.0, with_error=True)
    run_case("pyproto_case_b", seed=23, M=9, N=64, f0=600.0, k22=600.0, with_error=False)
    rodrigues_case()
    dino_prep_case()
