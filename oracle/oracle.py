"""ctypes binding of the CPU ORACLE (oracle/libba_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from surikatoko_amd/ (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libba_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


class Report(C.Structure):
    _fields_ = [("status", C.c_int32), ("optimized", C.c_int32), ("iterations", C.c_int64),
                ("attempts", C.c_int64), ("seen", C.c_int64), ("err_initial", C.c_double),
                ("err_final", C.c_double), ("hessian_factor", C.c_double), ("world_scale", C.c_double),
                ("sec_derivatives", C.c_double), ("sec_schur", C.c_double), ("sec_solve", C.c_double),
                ("sec_backsub", C.c_double), ("sec_apply", C.c_double), ("sec_error", C.c_double)]


class Normalizer(C.Structure):
    _fields_ = [("R0", C.c_double * 9), ("T0", C.c_double * 3), ("world_scale", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_reproj_error.restype = C.c_double
        _lib.orc_status_string.restype = C.c_char_p
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Scene:
    """Flat scene arrays in the layout of include/srk_ba.h (copies are made)."""

    def __init__(self, points, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv):
        self.points = _f64(points).reshape(-1, 3).copy()
        self.cam_R = _f64(cam_R).reshape(-1, 9).copy()
        self.cam_T = _f64(cam_T).reshape(-1, 3).copy()
        self.K = _f64(K).reshape(-1, 9).copy()
        self.shared_k = int(bool(shared_k))
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64).copy()
        self.obs_frame = np.ascontiguousarray(obs_frame, dtype=np.int32).copy()
        self.obs_uv = _f64(obs_uv).reshape(-1, 2).copy()

    @property
    def N(self):
        return self.points.shape[0]

    @property
    def M(self):
        return self.cam_R.shape[0]

    @property
    def O(self):
        return int(self.row_ptr[-1])

    def copy(self):
        return Scene(self.points, self.cam_R, self.cam_T, self.K, self.shared_k, self.row_ptr, self.obs_frame,
                     self.obs_uv)

    def _args(self):
        return (C.c_int64(self.N), _d(self.points), C.c_int32(self.M), _d(self.cam_R), _d(self.cam_T), _d(self.K),
                C.c_int32(self.shared_k), _i64(self.row_ptr), _i32(self.obs_frame), _d(self.obs_uv))


def project_onto_so3(R_noisy, T_noisy):
    """(ok, R, T) of ProjectOntoSO3 (multi-view-factorization.cpp:79-104); R row-major 3x3."""
    Rn = np.ascontiguousarray(R_noisy, dtype=np.float64).reshape(9)
    Tn = np.ascontiguousarray(T_noisy, dtype=np.float64).reshape(3)
    R, T = np.zeros(9), np.zeros(3)
    f = lib().orc_project_onto_so3
    f.restype = C.c_int
    ok = f(_d(Rn), _d(Tn), _d(R), _d(T))
    return bool(ok), R.reshape(3, 3), T


def mvf_relative_motion(x_anchor, x_target, depth_anchor):
    """(ok, R, T) of FindRelativeMotionMultiPoints (multi-view-factorization.cpp:107-189)."""
    xa = np.ascontiguousarray(x_anchor, dtype=np.float64).reshape(-1, 3)
    xt = np.ascontiguousarray(x_target, dtype=np.float64).reshape(-1, 3)
    dp = np.ascontiguousarray(depth_anchor, dtype=np.float64).reshape(-1)
    R, T = np.zeros(9), np.zeros(3)
    f = lib().orc_mvf_relative_motion
    f.restype = C.c_int
    ok = f(C.c_int64(xa.shape[0]), _d(xa), _d(xt), _d(dp), _d(R), _d(T))
    return bool(ok), R.reshape(3, 3), T


def mvf_point_depth(frame, x_meter, cam_R, cam_T):
    """Estimate3DPointDepthFromFrames (multi-view-factorization.cpp:223-253) for one track."""
    fr = np.ascontiguousarray(frame, dtype=np.int32)
    xm = np.ascontiguousarray(x_meter, dtype=np.float64).reshape(-1, 3)
    R = np.ascontiguousarray(cam_R, dtype=np.float64).reshape(-1, 9)
    T = np.ascontiguousarray(cam_T, dtype=np.float64).reshape(-1, 3)
    f = lib().orc_mvf_point_depth
    f.restype = C.c_double
    return float(f(C.c_int64(fr.shape[0]), _i32(fr), _d(xm), _d(R), _d(T)))


def reproj_error_mvf(f0, sc):
    """(ok, err, summands) of MultiViewIterativeFactorizer::ReprojError (multi-view-factorization.cpp:415-475)."""
    e, n = C.c_double(0), C.c_int64(0)
    f = lib().orc_reproj_error_mvf
    f.restype = C.c_int
    ok = f(C.c_double(f0), *sc._args(), C.byref(e), C.byref(n))
    return bool(ok), e.value, n.value


def reproj_error(f0, sc):
    seen = C.c_int64(0)
    e = lib().orc_reproj_error(C.c_double(f0), *sc._args(), C.byref(seen))
    return float(e), int(seen.value)


def normalize(sc, t1y=1.0, comp=1):
    nrm = Normalizer()
    ok = lib().orc_normalize_scene(C.c_int64(sc.N), _d(sc.points), C.c_int32(sc.M), _d(sc.cam_R), _d(sc.cam_T),
                                   C.c_double(t1y), C.c_int32(comp), C.byref(nrm))
    return bool(ok), nrm


def revert(sc, nrm):
    lib().orc_revert_normalization(C.c_int64(sc.N), _d(sc.points), C.c_int32(sc.M), _d(sc.cam_R), _d(sc.cam_T),
                                   C.byref(nrm))


def check_normalized(sc, t1y=1.0, comp=1):
    return bool(lib().orc_check_world_is_normalized(C.c_int32(sc.M), _d(sc.cam_R), _d(sc.cam_T), C.c_double(t1y),
                                                    C.c_int32(comp)))


def derivatives(f0, sc):
    N, M, O = sc.N, sc.M, sc.O
    gradE = np.zeros(3 * N + 10 * M)
    V = np.zeros((N, 3, 3))
    U = np.zeros((M, 10, 10))
    W = np.zeros((O, 3, 10))
    lib().orc_derivatives(C.c_double(f0), *sc._args(), _d(gradE), _d(V), _d(U), _d(W))
    return gradE, V, U, W


def two_phase(sc, gradE, V, U, W, c, comp=1, dense_literal=False, want_system=False):
    N, M = sc.N, sc.M
    n = 10 * M - 7
    corr = np.zeros(3 * N + 10 * M)
    S = np.zeros((n, n)) if want_system else None
    rhs = np.zeros(n) if want_system else None
    ok = lib().orc_two_phase(C.c_int64(N), C.c_int32(M), _i64(sc.row_ptr), _i32(sc.obs_frame), _d(gradE), _d(V),
                             _d(U), _d(W), C.c_double(c), C.c_int32(comp), C.c_int32(int(dense_literal)), _d(corr),
                             _d(S) if want_system else None, _d(rhs) if want_system else None, None, None, None)
    if want_system:
        return bool(ok), corr, S, rhs
    return bool(ok), corr


def two_phase_skyline(sc, gradE, V, U, W, c, comp=1, sel_rows=None, want_rhs=False):
    """Baseline variant (ii): the two-phase step on skyline storage with a skyline Cholesky (ba_oracle.c:
    orc_two_phase_skyline).  sel_rows: reduced row indices whose rows of the system (before the factorisation, columns <=
    row filled) are returned as an [len(sel_rows), 10M-7] array.  Returns (ok, corrections[, rows][, rhs])."""
    N, M = sc.N, sc.M
    n = 10 * M - 7
    corr = np.zeros(3 * N + 10 * M)
    sel = np.ascontiguousarray(sel_rows, dtype=np.int64) if sel_rows is not None else None
    rows = np.zeros((len(sel), n)) if sel is not None else None
    rhs = np.zeros(n) if want_rhs else None
    f = lib().orc_two_phase_skyline
    f.restype = C.c_int
    ok = f(C.c_int64(N), C.c_int32(M), _i64(sc.row_ptr), _i32(sc.obs_frame), _d(gradE), _d(V), _d(U), _d(W), C.c_double(c),
           C.c_int32(comp), _d(corr), _i64(sel) if sel is not None else None, C.c_int64(len(sel) if sel is not None else 0),
           _d(rows) if rows is not None else None, _d(rhs) if rhs is not None else None, None, None, None)
    out = (bool(ok), corr)
    if rows is not None:
        out += (rows,)
    if rhs is not None:
        out += (rhs,)
    return out


def set_solver(mode):
    """0 = the reference's Householder QR (default, what the parity tests pin); 1 = skyline Cholesky inside compute_inplace
    (BASELINE.md baseline variant (ii); bench.py's cpu_baseline leg)."""
    lib().orc_set_solver(C.c_int(int(mode)))


def naive_solve(sc, gradE, V, U, W, c, comp=1):
    corr = np.zeros(3 * sc.N + 10 * sc.M)
    ok = lib().orc_naive_solve(C.c_int64(sc.N), C.c_int32(sc.M), _i64(sc.row_ptr), _i32(sc.obs_frame), _d(gradE),
                               _d(V), _d(U), _d(W), C.c_double(c), C.c_int32(comp), _d(corr))
    return bool(ok), corr


def apply_corrections(sc, corr):
    corr = _f64(corr)
    lib().orc_apply_corrections(C.c_int64(sc.N), _d(sc.points), C.c_int32(sc.M), _d(sc.cam_R), _d(sc.cam_T), _d(corr))


def compute_inplace(f0, sc, allowed_err_change=None, max_hessian_factor=None, max_iterations=0, dense_literal=False):
    rep = Report()
    a = C.byref(C.c_double(allowed_err_change)) if allowed_err_change is not None else None
    m = C.byref(C.c_double(max_hessian_factor)) if max_hessian_factor is not None else None
    rc = lib().orc_compute_inplace(C.c_double(f0), *sc._args(), a, m, C.c_int64(max_iterations),
                                   C.c_int32(int(dense_literal)), C.byref(rep))
    return rc, rep


def fd_point(f0, sc, pi, eps=1e-5):
    """finite-difference first [3] and second [3][3] derivatives of the error w.r.t. landmark pi (BA:895-946)"""
    d1, d2 = np.zeros(3), np.zeros((3, 3))
    lib().orc_fd_point(C.c_double(f0), *sc._args(), C.c_int64(pi), C.c_double(eps), _d(d1), _d(d2))
    return d1, d2


def fd_frame(f0, sc, fj, eps=1e-5):
    """finite-difference first [10] and second [10][10] derivatives w.r.t. the variables of frame fj (BA:948-1086)"""
    d1, d2 = np.zeros(10), np.zeros((10, 10))
    lib().orc_fd_frame(C.c_double(f0), *sc._args(), C.c_int32(fj), C.c_double(eps), _d(d1), _d(d2))
    return d1, d2


def fd_point_frame(f0, sc, pi, fj, eps=1e-5):
    """finite-difference mixed second derivatives [3][10] of landmark pi and frame fj (BA:1088-1138)"""
    d2 = np.zeros((3, 10))
    lib().orc_fd_point_frame(C.c_double(f0), *sc._args(), C.c_int64(pi), C.c_int32(fj), C.c_double(eps), _d(d2))
    return d2


def set_threads(n):
    """Baseline variant (iii) only: OpenMP threads of the derivative / Schur / QR-update / back-substitution loops
    (1 = the reference's sequential order, what every parity test uses)."""
    lib().orc_set_threads(C.c_int(int(n)))


def set_w_storage_f32(mode):
    """0 / False: off; 1 / True: the 30 products of every point-frame block rounded to float once; 2: its rank-2 FACTORS rounded
    to float and the products formed in double -- what the HIP path's f32 storage mode stores"""
    lib().orc_set_w_storage_f32(C.c_int(2 if mode == 2 else int(bool(mode))))


def set_skip_solve(on):
    """bench.py's cpu_baseline leg only: time derivatives + Schur + back-substitution without the dense QR."""
    lib().orc_set_skip_solve(C.c_int(int(bool(on))))


def get_threads():
    return int(lib().orc_get_threads())


def status_string(status):
    return lib().orc_status_string(C.c_int(status)).decode()


def rot_from_axis_angle(w):
    w = _f64(w)
    R = np.zeros(9)
    ok = lib().orc_rot_from_axis_angle(_d(w), _d(R))
    return bool(ok), R.reshape(3, 3)


def rot_from_unity_dir_and_angle(d, ang, check_input=True):
    d = _f64(d)
    R = np.zeros(9)
    ok = lib().orc_rot_from_unity_dir_and_angle(_d(d), C.c_double(ang), _d(R), C.c_int(int(check_input)))
    return bool(ok), R.reshape(3, 3)


def axis_angle_from_rot(R):
    R = _f64(R).reshape(9)
    w = np.zeros(3)
    ok = lib().orc_axis_angle_from_rot(_d(R), _d(w))
    return bool(ok), w


def skew(v):
    v = _f64(v)
    S = np.zeros(9)
    lib().orc_skew(_d(v), _d(S))
    return S.reshape(3, 3)


def circle_camera_shots(center, radius, ascent_z, angles):
    center = _f64(center)
    angles = _f64(angles)
    n = angles.shape[0]
    R = np.zeros((n, 9))
    T = np.zeros((n, 3))
    lib().orc_circle_camera_shots(_d(center), C.c_double(radius), C.c_double(ascent_z), C.c_int32(n), _d(angles),
                                  _d(R), _d(T))
    return R, T


def remove_rows_cols(mat, rm_rows, rm_cols):
    m = np.ascontiguousarray(mat, dtype=np.int64).copy()
    rows, cols = m.shape
    rr = np.ascontiguousarray(rm_rows, dtype=np.int64)
    rc = np.ascontiguousarray(rm_cols, dtype=np.int64)
    nr, nc = C.c_int64(0), C.c_int64(0)
    flat = m.reshape(-1).copy()
    lib().orc_remove_rows_cols(C.c_int64(rows), C.c_int64(cols), _i64(flat), C.c_int32(len(rr)), _i64(rr),
                               C.c_int32(len(rc)), _i64(rc), C.byref(nr), C.byref(nc))
    return flat[: nr.value * nc.value].reshape(nr.value, nc.value)


def householder_qr_solve(A, b):
    A = np.asfortranarray(_f64(A)).copy(order="F")
    n = A.shape[0]
    b = _f64(b)
    x = np.zeros(n)
    ok = lib().orc_householder_qr_solve(C.c_int64(n), A.ctypes.data_as(C.POINTER(C.c_double)), _d(b), _d(x))
    return bool(ok), x


def inverse3x3(A):
    A = _f64(A).reshape(9)
    Ai = np.zeros(9)
    det = C.c_double(0)
    ok = lib().orc_inverse3x3_with_check(_d(A), _d(Ai), C.byref(det))
    return bool(ok), Ai.reshape(3, 3), det.value
