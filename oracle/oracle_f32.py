"""ctypes binding of the f32 build of the CPU ORACLE (oracle/libba_oracle_f32.so).

TEST INFRASTRUCTURE ONLY (tests/ and nothing else).  The library is oracle/ba_oracle.c compiled with -DORC_F32: every
scalar of the restatement -- interface arrays, intermediates, the report -- is a C float, i.e. the reference built with
`suriko_scalar_type_string=f32` (cpp_impl/suriko-engine/CMakeLists.txt:14-15,76-82; `Scalar` of rt-config.h:41-48).
It pins the f32 row of SURVEY 8(f): what the reference's own float arithmetic gives on a scene, as the yardstick for the
product's two f32 entry points (the f32 boundary `srk_ba_compute_inplace_f32` and the f32 storage mode).
Only the entry points those tests need are bound.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libba_oracle_f32.so")
_lib = None


class Report(C.Structure):
    _fields_ = [("status", C.c_int32), ("optimized", C.c_int32), ("iterations", C.c_int64),
                ("attempts", C.c_int64), ("seen", C.c_int64), ("err_initial", C.c_float),
                ("err_final", C.c_float), ("hessian_factor", C.c_float), ("world_scale", C.c_float),
                ("sec_derivatives", C.c_float), ("sec_schur", C.c_float), ("sec_solve", C.c_float),
                ("sec_backsub", C.c_float), ("sec_apply", C.c_float), ("sec_error", C.c_float)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libba_oracle_f32.so"])
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_reproj_error.restype = C.c_float
        _lib.orc_scalar_bytes.restype = C.c_int
        assert _lib.orc_scalar_bytes() == 4
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class SceneF32:
    """A scene in float arrays (copies; the layout of include/srk_ba.h)."""

    def __init__(self, sc):
        self.points = _f32(sc.points).reshape(-1, 3).copy()
        self.cam_R = _f32(sc.cam_R).reshape(-1, 9).copy()
        self.cam_T = _f32(sc.cam_T).reshape(-1, 3).copy()
        self.K = _f32(sc.K).reshape(-1, 9).copy()
        self.shared_k = int(bool(sc.shared_k))
        self.row_ptr = np.ascontiguousarray(sc.row_ptr, dtype=np.int64).copy()
        self.obs_frame = np.ascontiguousarray(sc.obs_frame, dtype=np.int32).copy()
        self.obs_uv = _f32(sc.obs_uv).reshape(-1, 2).copy()

    @property
    def N(self):
        return self.points.shape[0]

    @property
    def M(self):
        return self.cam_R.shape[0]

    @property
    def O(self):
        return int(self.row_ptr[-1])

    def _args(self):
        return (C.c_int64(self.N), _f(self.points), C.c_int32(self.M), _f(self.cam_R), _f(self.cam_T), _f(self.K),
                C.c_int32(self.shared_k), self.row_ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                self.obs_frame.ctypes.data_as(C.POINTER(C.c_int32)), _f(self.obs_uv))


def reproj_error(f0, sc):
    seen = C.c_int64(0)
    e = lib().orc_reproj_error(C.c_float(f0), *sc._args(), C.byref(seen))
    return float(e), seen.value


def derivatives(f0, sc):
    N, M, O = sc.N, sc.M, sc.O
    gradE = np.zeros(3 * N + 10 * M, dtype=np.float32)
    V = np.zeros((N, 3, 3), dtype=np.float32)
    U = np.zeros((M, 10, 10), dtype=np.float32)
    W = np.zeros((O, 3, 10), dtype=np.float32)
    lib().orc_derivatives(C.c_float(f0), *sc._args(), _f(gradE), _f(V), _f(U), _f(W))
    return gradE, V, U, W


def compute_inplace(f0, sc, allowed_err_change=None, max_hessian_factor=None, max_iterations=0):
    """ComputeInplace of the reference's f32 build on `sc` (a SceneF32, modified in place): (rc, Report)."""
    rep = Report()
    a = C.byref(C.c_float(allowed_err_change)) if allowed_err_change is not None else None
    m = C.byref(C.c_float(max_hessian_factor)) if max_hessian_factor is not None else None
    rc = lib().orc_compute_inplace(C.c_float(f0), *sc._args(), a, m, C.c_int64(max_iterations), C.c_int32(0), C.byref(rep))
    return rc, rep
