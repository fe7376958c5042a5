"""The multi-view-factorization steps on either side of the BA call: batched depth estimation
(MultiViewIterativeFactorizer::Estimate3DPointDepthFromFrames, multi-view-factorization.cpp:223-253), relative motion
from common points (FindRelativeMotionMultiPoints, :107-189) and the SO(3) projection (ProjectOntoSO3, :79-104)."""
import ctypes as C

import numpy as np

from ._lib import lib


def _d(a):
    return a.ctypes.data_as(C.c_void_p)


def project_onto_so3(R_noisy, T_noisy):
    """(ok, R, T); host code, no GPU needed."""
    Rn = np.ascontiguousarray(R_noisy, dtype=np.float64).reshape(9)
    Tn = np.ascontiguousarray(T_noisy, dtype=np.float64).reshape(3)
    R, T = np.zeros(9), np.zeros(3)
    rc = lib().srk_mvf_project_onto_so3(_d(Rn), _d(Tn), _d(R), _d(T))
    if rc < 0:
        raise ValueError("srk_mvf_project_onto_so3: bad arguments")
    return rc == 1, R.reshape(3, 3), T


def estimate_depths(ba, row_ptr, frame, x_meter, cam_R, cam_T):
    """Depth of every track in its base (first observed) frame; NaN for tracks seen fewer than twice."""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    fr = np.ascontiguousarray(frame, dtype=np.int32)
    xm = np.ascontiguousarray(x_meter, dtype=np.float64).reshape(-1, 3)
    R = np.ascontiguousarray(cam_R, dtype=np.float64).reshape(-1, 9)
    T = np.ascontiguousarray(cam_T, dtype=np.float64).reshape(-1, 3)
    n = rp.shape[0] - 1
    out = np.zeros(max(n, 0))
    ba._raise(lib().srk_mvf_estimate_depths(C.c_void_p(ba._h), C.c_int64(n), _d(rp), _d(fr), _d(xm), C.c_int32(R.shape[0]),
                                            _d(R), _d(T), _d(out)))
    return out


def relative_motion(ba, x_anchor, x_target, depth_anchor):
    """(ok, R, T): target_from_anchor camera motion from >= 6 common points."""
    xa = np.ascontiguousarray(x_anchor, dtype=np.float64).reshape(-1, 3)
    xt = np.ascontiguousarray(x_target, dtype=np.float64).reshape(-1, 3)
    dp = np.ascontiguousarray(depth_anchor, dtype=np.float64).reshape(-1)
    if not (xa.shape[0] == xt.shape[0] == dp.shape[0]):
        raise ValueError("relative_motion: array lengths differ")
    R, T = np.zeros(9), np.zeros(3)
    rc = lib().srk_mvf_relative_motion(C.c_void_p(ba._h), C.c_int64(xa.shape[0]), _d(xa), _d(xt), _d(dp), _d(R), _d(T))
    if rc < 0:
        ba._raise(rc)
    return rc == 1, R.reshape(3, 3), T
