"""The f32 build of the CPU oracle (oracle/libba_oracle_f32.so = ba_oracle.c with the reference's Scalar = float,
rt-config.h:41-48): it must BE float arithmetic (scalar size 4, results a float-sized distance from the fp64 build, not
identical to it) and still the same algorithm (blocks, error and LM decisions of the fp64 build to float accuracy).
No GPU involved."""
import numpy as np

import surikatoko_amd as sa
from conftest import rel_err


def _scenes(orc, o32, spec):
    sc = sa.generate_scene(spec)
    f32 = {k: np.ascontiguousarray(getattr(sc, k), dtype=np.float32) for k in ("points", "cam_R", "cam_T", "K", "obs_uv")}
    s64 = orc.Scene(f32["points"], f32["cam_R"], f32["cam_T"], f32["K"], sc.shared_k, sc.row_ptr, sc.obs_frame, f32["obs_uv"])
    return s64, o32.SceneF32(s64)


def test_f32_build_is_float_arithmetic_of_the_same_algorithm(orc):
    from oracle import oracle_f32 as o32
    assert o32.lib().orc_scalar_bytes() == 4 and orc.lib().orc_scalar_bytes() == 8
    spec = sa.SceneSpec(n_frames=12, grid_nx=12, grid_ny=8, vis_window=6, noise_uv_pix=0.3)
    f0 = float(np.float32(spec.f0))
    s64, s32 = _scenes(orc, o32, spec)
    e64, n64 = orc.reproj_error(f0, s64)
    e32, n32 = o32.reproj_error(f0, s32)
    assert n32 == n64 and e32 != e64 and abs(e32 - e64) < 1e-4 * e64
    g64, V64, U64, W64 = orc.derivatives(f0, s64)
    g32, V32, U32, W32 = o32.derivatives(f0, s32)
    for a, b, tol in ((V32, V64, 1e-5), (U32, U64, 1e-4), (W32, W64, 1e-5), (g32, g64, 1e-3)):
        assert a.dtype == np.float32 and 0 < rel_err(a, b) < tol
    rc64, r64 = orc.compute_inplace(f0, s64, None, None, 6)
    rc32, r32 = o32.compute_inplace(f0, s32, None, None, 6)
    assert (r32.iterations, r32.attempts) == (r64.iterations, r64.attempts)      # no tie on this scene
    assert 1e-7 < abs(r32.err_final - r64.err_final) / r64.err_final < 0.05
    assert 1e-7 < np.abs(s32.points - s64.points).max() < 0.01


def test_f32_build_uses_eigens_float_invertibility_threshold(orc):
    """computeInverseAndDetWithCheck (bundle-adj-kanatani.cpp:1876) takes NumTraits<Scalar>::dummy_precision(): 1e-12 for
    double, 1e-5 for float.  A point block with |det| = 1e-8 is inverted by the fp64 build and skipped by the f32 build."""
    import ctypes as C
    from oracle import oracle_f32 as o32
    A = np.diag([1e-2, 1e-3, 1e-3])  # det 1e-8
    ok64, _, det64 = orc.inverse3x3(A)
    assert ok64 and det64 == np.float64(1e-8) or abs(det64 - 1e-8) < 1e-20
    A32 = np.ascontiguousarray(A, dtype=np.float32).reshape(9)
    inv = np.zeros(9, dtype=np.float32)
    det = C.c_float(0)
    f = o32.lib().orc_inverse3x3_with_check
    f.restype = C.c_int
    ok32 = f(A32.ctypes.data_as(C.POINTER(C.c_float)), inv.ctypes.data_as(C.POINTER(C.c_float)), C.byref(det))
    assert ok32 == 0 and abs(det.value - 1e-8) < 1e-12
