"""Host-side mirror of the reference's BA operator interface over the C ABI (include/srk_ba.h).

Same names, argument meaning and error behaviour as suriko's
  BundleAdjustmentKanatani            cpp_impl/suriko-engine/include/suriko/bundle-adj-kanatani.h:96-261
  BundleAdjustmentKanataniTermCriteria                                             .h:68-92
  NormalizeSceneInplace / CheckWorldIsNormalized                                   .h:61-65
so that tests/ read like the reference's own tests.  Scenes are flat arrays (class Scene) instead of
FragmentMap / CornerTrackRepository / std::vector<SE3Transform>; the C++ adapter that converts those
containers is include/suriko_amd/bundle-adj-kanatani.hpp.

Everything computes on the GPU through libsrk_ba.so.  Nothing here imports oracle/.
"""
import ctypes as C
import math

import numpy as np

from ._lib import Normalizer, Report, lib

BUF_GRAD, BUF_POINT_BLOCKS, BUF_FRAME_BLOCKS, BUF_POINT_FRAME, BUF_RCS, BUF_RCS_RHS, BUF_CORRECTIONS, BUF_POINTS, \
    BUF_CAM_R, BUF_CAM_T = range(10)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def status_string(status):
    return lib().srk_ba_status_string(int(status)).decode()


def device_count():
    return int(lib().srk_ba_device_count())


class Scene:
    """Flat scene arrays in the C-ABI layout (include/srk_ba.h); arrays are owned copies, mutated in place by BA."""

    def __init__(self, points, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv):
        self.points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3).copy()
        self.cam_R = np.ascontiguousarray(cam_R, dtype=np.float64).reshape(-1, 9).copy()
        self.cam_T = np.ascontiguousarray(cam_T, dtype=np.float64).reshape(-1, 3).copy()
        self.K = np.ascontiguousarray(K, dtype=np.float64).reshape(-1, 9).copy()
        self.shared_k = int(bool(shared_k))
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64).copy()
        self.obs_frame = np.ascontiguousarray(obs_frame, dtype=np.int32).copy()
        self.obs_uv = np.ascontiguousarray(obs_uv, dtype=np.float64).reshape(-1, 2).copy()

    N = property(lambda self: self.points.shape[0])
    M = property(lambda self: self.cam_R.shape[0])
    O = property(lambda self: int(self.row_ptr[-1]))

    def copy(self):
        return Scene(self.points, self.cam_R, self.cam_T, self.K, self.shared_k, self.row_ptr, self.obs_frame,
                     self.obs_uv)

    def shard(self, rank, world):
        """Landmark shard of this scene for `rank` of `world` (contiguous pnt_ind range balanced by observation
        count, cameras replicated) -- SURVEY 8e partitioning."""
        lo, hi = shard_bounds(self.row_ptr, rank, world)
        o0, o1 = int(self.row_ptr[lo]), int(self.row_ptr[hi])
        return Scene(self.points[lo:hi], self.cam_R, self.cam_T, self.K, self.shared_k,
                     self.row_ptr[lo:hi + 1] - o0, self.obs_frame[o0:o1], self.obs_uv[o0:o1]), (lo, hi)

    def scene_args(self):
        return (C.c_int64(self.N), _p(self.points), C.c_int32(self.M), _p(self.cam_R), _p(self.cam_T), _p(self.K),
                C.c_int(self.shared_k), _p(self.row_ptr), _p(self.obs_frame), _p(self.obs_uv))


def covisibility(scene, to_internal=None):
    """min_cv[j] = smallest frame index that shares a landmark with frame j (the skyline of the reduced camera
    system); computed on the WHOLE scene before sharding.  to_internal: the frame numbering given to
    set_frame_order() -- the result is then in that numbering, as srk_ba_set_covisibility wants it."""
    M = scene.M
    f = scene.obs_frame if to_internal is None else np.asarray(to_internal, np.int32)[scene.obs_frame]
    counts = np.diff(scene.row_ptr)
    first = np.minimum.reduceat(f, scene.row_ptr[:-1][counts > 0]) if scene.O else np.zeros(0, np.int32)
    first_per_obs = np.repeat(first, counts[counts > 0])
    min_cv = np.arange(M, dtype=np.int64)
    np.minimum.at(min_cv, f, first_per_obs)
    return min_cv.astype(np.int32)


def frame_order(scene, mode=-1):
    """srk_frame_order on the host: to_internal[frame] when renumbering the frames pays (mode -1) / differs (mode 1), else
    None.  For landmark shards: find it on the whole scene, give it to every rank's set_frame_order()."""
    to_int = np.zeros(scene.M, np.int32)
    rc = lib().srk_frame_order(C.c_int(mode), C.c_int64(scene.N), C.c_int32(scene.M), _p(scene.row_ptr), _p(scene.obs_frame), _p(to_int))
    if rc < 0:
        raise ValueError("srk_frame_order: bad arguments")
    return to_int if rc == 1 else None


def shard_bounds(row_ptr, rank, world):
    """Contiguous pnt_ind range [lo, hi) of `rank`: cut points chosen so every rank gets ~O/world observations."""
    row_ptr = np.asarray(row_ptr)
    N = len(row_ptr) - 1
    O = int(row_ptr[-1])
    cuts = [0]
    for r in range(1, world):
        target = O * r // world
        cuts.append(int(np.searchsorted(row_ptr, target, side="left")))
    cuts.append(N)
    cuts = [min(max(c, 0), N) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


class BundleAdjustmentKanataniTermCriteria:
    """bundle-adj-kanatani.h:68-92 -- two optionals; None = unset ('potentially optimize forever')."""

    def __init__(self):
        self._allowed_reproj_err_rel_change = None
        self._max_hessian_factor = None

    def AllowedReprojErrRelativeChange(self, value="__get__"):
        if value == "__get__":
            return self._allowed_reproj_err_rel_change
        self._allowed_reproj_err_rel_change = value

    def MaxHessianFactor(self, value="__get__"):
        if value == "__get__":
            return self._max_hessian_factor
        self._max_hessian_factor = value


def normalize_scene_inplace(scene, t1y_dist=1.0, unity_comp_ind=1):
    """NormalizeSceneInplace (bundle-adj-kanatani.h:61-62, .cpp:277-286).  Returns (success, normalizer)."""
    if not (0 <= unity_comp_ind < 3):
        raise ValueError("Can normalize only one of [T1x, T1y, Tz] components")  # CHECK at .cpp:129
    nrm = Normalizer()
    ok = lib().srk_ba_normalize_scene(C.c_int64(scene.N), _p(scene.points), C.c_int32(scene.M), _p(scene.cam_R),
                                      _p(scene.cam_T), C.c_double(t1y_dist), C.c_int32(unity_comp_ind), C.byref(nrm))
    return bool(ok), nrm


def revert_normalization(scene, nrm):
    """SceneNormalizer::RevertNormalization (.cpp:249-270)."""
    lib().srk_ba_revert_normalization(C.c_int64(scene.N), _p(scene.points), C.c_int32(scene.M), _p(scene.cam_R),
                                      _p(scene.cam_T), C.byref(nrm))


def check_world_is_normalized(scene, t1y=1.0, unity_comp_ind=1):
    """CheckWorldIsNormalized (.cpp:288-333)."""
    return bool(lib().srk_ba_check_world_is_normalized(C.c_int32(scene.M), _p(scene.cam_R), _p(scene.cam_T),
                                                        C.c_double(t1y), C.c_int32(unity_comp_ind)))


class BundleAdjustmentKanatani:
    """Mirror of suriko::BundleAdjustmentKanatani on top of one srk_ba handle (one GPU)."""

    kPointVarsCount = 3
    kIntrinsicVarsCount = 4
    kTVarsCount = 3
    kWVarsCount = 3
    kMaxFrameVarsCount = 10

    def __init__(self, device=0):
        self._lib = lib()
        self._h = self._lib.srk_ba_create(int(device))
        if not self._h:
            raise RuntimeError("srk_ba_create failed: no usable HIP device (there is no CPU fallback)")
        self._f0 = 0.0
        self._scene = None
        self._status = ""
        self.report = Report()
        self._hook = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.srk_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API
    def ComputeInplace(self, f0, scene, term_crit=None, max_iterations=0):
        """bool ComputeInplace(f0, map, inverse_orient_cams, track_rep, shared_K | Ks, term_crit)
        (bundle-adj-kanatani.h:179-184).  `scene` is updated in place.  Raises ValueError where the reference
        CHECK-aborts (f0 ~ 0; fewer than two frames)."""
        a, m = self._criteria(term_crit)
        self._f0 = float(f0)
        self._scene = scene
        self.report = Report()
        rc = self._lib.srk_ba_compute_inplace(C.c_void_p(self._h), C.c_double(f0), *scene.scene_args(), a, m,
                                              C.c_int64(max_iterations), C.byref(self.report))
        self._raise(rc)
        self._status = status_string(self.report.status)
        return rc == 0

    def ComputeInplaceF32(self, f0, points, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv, term_crit=None,
                          max_iterations=0):
        """The call of a reference built with Scalar = float: float32 arrays (updated in place) at the boundary, the
        fp64 pipeline inside (srk_ba_compute_inplace_f32)."""
        for arr in (points, cam_R, cam_T, K, obs_uv):
            if arr.dtype != np.float32 or not arr.flags["C_CONTIGUOUS"]:
                raise ValueError("ComputeInplaceF32 takes C-contiguous float32 arrays")
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        fr = np.ascontiguousarray(obs_frame, dtype=np.int32)
        a64, m64 = self._criteria(term_crit)
        a32 = C.c_float(term_crit.AllowedReprojErrRelativeChange()) if a64 is not None else None
        m32 = C.c_float(term_crit.MaxHessianFactor()) if m64 is not None else None
        self._f0 = float(f0)
        self.report = Report()
        rc = self._lib.srk_ba_compute_inplace_f32(
            C.c_void_p(self._h), C.c_float(f0), C.c_int64(points.shape[0]), _p(points), C.c_int32(cam_R.shape[0]), _p(cam_R),
            _p(cam_T), _p(K), C.c_int(int(bool(shared_k))), _p(rp), _p(fr), _p(obs_uv),
            C.byref(a32) if a32 is not None else None, C.byref(m32) if m32 is not None else None,
            C.c_int64(max_iterations), C.byref(self.report))
        self._raise(rc)
        self._status = status_string(self.report.status)
        return rc == 0

    def ReprojError(self, f0, scene):
        """static Scalar ReprojError(...) (.h:167-172, .cpp:410-490) -> (err, seen_points_count)."""
        seen = C.c_int64(0)
        e = self._lib.srk_ba_reproj_error(C.c_void_p(self._h), C.c_double(f0), *scene.scene_args(), C.byref(seen))
        if math.isnan(e):
            msg = self.last_error()
            if "hip" in msg.lower():
                raise RuntimeError("srk_ba_reproj_error: " + msg)
            raise ValueError("srk_ba_reproj_error: " + msg)  # the reference CHECK-aborts on bad arguments (:420-421)
        self._f0 = float(f0)
        return float(e), int(seen.value)

    def ReprojErrorMvf(self, f0, scene, z_tol=1e-5):
        """MultiViewIterativeFactorizer::ReprojError (multi-view-factorization.cpp:415-475) -> (ok, err, summands):
        the score the MVF driver takes before deciding to run BA; observations with |z| <= z_tol are skipped and
        ok is False when nothing was summed."""
        e, n = C.c_double(0), C.c_int64(0)
        rc = self._lib.srk_ba_reproj_error_mvf(C.c_void_p(self._h), C.c_double(f0), *scene.scene_args(), C.c_double(z_tol),
                                               C.byref(e), C.byref(n))
        if rc < 0:
            msg = self.last_error()
            if "hip" in msg.lower():
                raise RuntimeError("srk_ba_reproj_error_mvf: " + msg)
            raise ValueError("srk_ba_reproj_error_mvf: " + msg)
        return rc == 1, float(e.value), int(n.value)

    def ReprojErrorPixPerPoint(self, reproj_err, seen_points_count):
        """f0 * sqrt(err / seen) (.cpp:602-615)."""
        return self._f0 * math.sqrt(reproj_err / float(seen_points_count))

    def OptimizationStatusString(self):
        return self._status

    def PointsCount(self):
        return self._scene.N

    def FramesCount(self):
        return self._scene.M

    def VarsCount(self):
        return 3 * self._scene.N + 10 * self._scene.M

    def NormalizedVarsCount(self):
        return self.VarsCount() - 7

    # ---- staged API (resident scene)
    def upload(self, f0, scene, already_normalized=False):
        self._f0 = float(f0)
        self._scene = scene
        rc = self._lib.srk_ba_upload_scene(C.c_void_p(self._h), C.c_double(f0), *scene.scene_args(),
                                           C.c_int(int(already_normalized)))
        if rc == 1:
            return False
        self._raise(rc)
        return True

    def optimize(self, term_crit=None, max_iterations=0):
        a, m = self._criteria(term_crit)
        self.report = Report()
        rc = self._lib.srk_ba_optimize(C.c_void_p(self._h), a, m, C.c_int64(max_iterations), C.byref(self.report))
        self._raise(rc)
        self._status = status_string(self.report.status)
        return rc == 0

    def download(self, scene=None, revert_normalization=True):
        scene = scene or self._scene
        self._raise(self._lib.srk_ba_download_scene(C.c_void_p(self._h), _p(scene.points), _p(scene.cam_R),
                                                    _p(scene.cam_T), C.c_int(int(revert_normalization))))
        return scene

    def reset(self):
        self._raise(self._lib.srk_ba_reset_scene(C.c_void_p(self._h)))

    def set_speculation(self, on=True):
        """Run the next damping factor beside the current attempt (one rank, instrumentation off); next upload."""
        self._raise(self._lib.srk_ba_set_speculation(C.c_void_p(self._h), C.c_int(int(bool(on)))))

    def set_deterministic(self, on=True):
        """ordered sums instead of fp64 atomics (srk_ba_set_deterministic); takes effect at the next upload"""
        self._raise(self._lib.srk_ba_set_deterministic(C.c_void_p(self._h), C.c_int(int(bool(on)))))

    def deterministic(self):
        """True when the uploaded scene runs in deterministic mode (srk_ba_deterministic)"""
        return int(self._lib.srk_ba_deterministic(C.c_void_p(self._h))) == 1

    MULTI_SCHEDULES = {"allreduce": 0, "dp": 1, "dp_force": 2}

    def set_multi_schedule(self, schedule="dp"):
        """exchange schedule with several ranks (srk_ba_set_multi_schedule): "dp" (default), "allreduce", "dp_force"; before upload"""
        self._raise(self._lib.srk_ba_set_multi_schedule(C.c_void_p(self._h), C.c_int(self.MULTI_SCHEDULES[schedule])))

    def multi_schedule(self):
        """schedule in effect (srk_ba_multi_schedule): 'allreduce', 'dp', 'dp (self-check passed)', 'allreduce (dp self-check failed)'"""
        return {0: "allreduce", 1: "dp", 2: "dp (self-check passed)", 3: "allreduce (dp self-check failed)"}[
            int(self._lib.srk_ba_multi_schedule(C.c_void_p(self._h)))]

    def set_frame_reordering(self, mode=-1):
        """-1 automatic, 0 = keep the caller's frame order, 1 = renumber whenever the ordering differs (next upload)"""
        self._raise(self._lib.srk_ba_set_frame_reordering(C.c_void_p(self._h), C.c_int(mode)))

    def set_frame_order(self, to_internal=None):
        """The frame numbering to use at the next upload (None: automatic); landmark shards: the same on every rank, found on
        the whole scene with surikatoko_amd.frame_order()."""
        if to_internal is None:
            self._raise(self._lib.srk_ba_set_frame_order(C.c_void_p(self._h), None, C.c_int32(0)))
        else:
            a = np.ascontiguousarray(to_internal, dtype=np.int32)
            self._raise(self._lib.srk_ba_set_frame_order(C.c_void_p(self._h), _p(a), C.c_int32(a.size)))

    def frame_order(self):
        """None when the frames are stored in the caller's order, else to_internal[caller's frame]"""
        to_int = np.zeros(self._scene.M if self._scene is not None else 0, np.int32)
        rc = self._lib.srk_ba_frame_order(C.c_void_p(self._h), _p(to_int))
        if rc < 0:
            self._raise(rc)
        return to_int if rc == 1 else None

    def set_solver_fusion(self, on=True):
        """Outer steps of the blocked Cholesky as one launch each (default) or as the panel / update launch sequence."""
        self._raise(self._lib.srk_ba_set_solver_fusion(C.c_void_p(self._h), C.c_int(int(bool(on)))))

    def solver_fusion(self):
        """True while the fused outer step is in use (False after a hand-off timeout, until the next upload / optimise call)"""
        return bool(self._lib.srk_ba_solver_fusion(C.c_void_p(self._h)))

    def solver_sync_timeouts(self):
        self._lib.srk_ba_solver_sync_timeouts.restype = C.c_int64
        return int(self._lib.srk_ba_solver_sync_timeouts(C.c_void_p(self._h)))

    def iteration_log(self):
        """accepted iterations of the last optimise / ComputeInplace call: dict of arrays attempts, ms (host time since the
        call began), err, hessian_factor (srk_ba_iteration_log)"""
        fn = self._lib.srk_ba_iteration_log
        fn.restype = C.c_int64
        n = int(fn(C.c_void_p(self._h), C.c_int64(0), None, None, None, None))
        att = np.zeros(max(n, 1), dtype=np.int32)
        ms = np.zeros(max(n, 1))
        err = np.zeros(max(n, 1))
        fac = np.zeros(max(n, 1))
        fn(C.c_void_p(self._h), C.c_int64(n), att.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p),
           err.ctypes.data_as(C.c_void_p), fac.ctypes.data_as(C.c_void_p))
        return {"attempts": att[:n], "ms": ms[:n], "err": err[:n], "hessian_factor": fac[:n]}

    def set_jacobian_mode(self, mode=-1):
        """-1 automatic, 0 = per-observation kernels only, 1 = run-based (uniform runs) whenever possible, 2 = run-based over
        frame unions (ragged tracks) whenever possible (next upload)"""
        self._raise(self._lib.srk_ba_set_jacobian_mode(C.c_void_p(self._h), C.c_int(mode)))

    def jacobian_kernel(self):
        return int(self._lib.srk_ba_jacobian_kernel(C.c_void_p(self._h)))

    def set_storage_precision(self, f32=False):
        """W stored as float (next upload); arithmetic stays fp64 -- see include/srk_ba.h"""
        self._raise(self._lib.srk_ba_set_storage_precision(C.c_void_p(self._h), C.c_int(int(bool(f32)))))

    def set_schur_precision(self, fp32=False):
        """Opt-in mixed precision: fp32 run sums in the grouped Schur kernel (everything else stays fp64)."""
        self._raise(self._lib.srk_ba_set_schur_precision(C.c_void_p(self._h), C.c_int(int(bool(fp32)))))

    def set_profile(self, level=2):
        """0 = no device events (default of the library), 1 = per-phase events, 2 / True = + MFMA update events."""
        if level is True:
            level = 2
        self._raise(self._lib.srk_ba_set_profile(C.c_void_p(self._h), C.c_int(int(level))))

    def set_covisibility(self, min_cv):
        """Global covisibility for sharded runs (see covisibility()); None = dense."""
        if min_cv is None:
            self._raise(self._lib.srk_ba_set_covisibility(C.c_void_p(self._h), None))
        else:
            a = np.ascontiguousarray(min_cv, dtype=np.int32)
            self._raise(self._lib.srk_ba_set_covisibility(C.c_void_p(self._h), _p(a)))

    def set_rcs_mode(self, mode=2):
        """0 / False = dense, 1 = skyline as one chain, 2 / True = skyline cut into chunks (default)."""
        if mode is True:
            mode = 2
        self._raise(self._lib.srk_ba_set_rcs_mode(C.c_void_p(self._h), C.c_int(int(mode))))

    def rcs_chunks(self):
        return int(self._lib.srk_ba_rcs_chunks(C.c_void_p(self._h)))

    def solve_mfma_flops(self):
        return float(self._lib.srk_ba_solve_mfma_flops(C.c_void_p(self._h)))

    def rcs_fill(self):
        return float(self._lib.srk_ba_rcs_fill(C.c_void_p(self._h)))

    def set_stream(self, hip_stream_handle):
        self._raise(self._lib.srk_ba_set_stream(C.c_void_p(self._h), C.c_void_p(hip_stream_handle)))

    def set_allreduce(self, hook, rank, world):
        """hook: an ALLREDUCE_FN instance (see surikatoko_amd/dist.py); kept alive by this object."""
        self._hook = hook
        self._raise(self._lib.srk_ba_set_allreduce(C.c_void_p(self._h), hook, None, int(rank), int(world)))

    # native RCCL exchange (include/srk_ba.h): no Python in the data path
    def rccl_unique_id(self):
        """128 bytes of a fresh ncclUniqueId (rank 0 calls this and hands the bytes to the other ranks)."""
        buf = (C.c_ubyte * 128)()
        self._raise(self._lib.srk_ba_rccl_get_unique_id(buf))
        return bytes(buf)

    def rccl_init(self, unique_id, rank, world):
        """Collective: every rank creates the communicator on its handle's device; replaces an all-reduce callback."""
        assert len(unique_id) == 128
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self._raise(self._lib.srk_ba_rccl_init(C.c_void_p(self._h), buf, C.c_int(int(rank)), C.c_int(int(world))))
        self._hook = None

    def rccl_init_second(self, unique_id):
        """Collective, after rccl_init: the second attempt slot's own communicator (keeps the attempt pairs with several ranks)."""
        assert len(unique_id) == 128
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self._raise(self._lib.srk_ba_rccl_init_second(C.c_void_p(self._h), buf))

    def phase_error(self):
        e, seen = C.c_double(0), C.c_int64(0)
        self._raise(self._lib.srk_ba_phase_error(C.c_void_p(self._h), C.byref(e), C.byref(seen)))
        return e.value, seen.value

    def phase_derivatives(self):
        self._raise(self._lib.srk_ba_phase_derivatives(C.c_void_p(self._h)))

    def phase_schur(self, hessian_factor):
        self._raise(self._lib.srk_ba_phase_schur(C.c_void_p(self._h), C.c_double(hessian_factor)))

    def phase_solve(self):
        rc = self._lib.srk_ba_phase_solve(C.c_void_p(self._h))
        self._raise(rc)
        return rc == 0

    def phase_backsub(self, hessian_factor):
        self._raise(self._lib.srk_ba_phase_backsub(C.c_void_p(self._h), C.c_double(hessian_factor)))

    def phase_accept(self):
        self._raise(self._lib.srk_ba_phase_accept(C.c_void_p(self._h)))

    def buffer(self, which):
        n = self._lib.srk_ba_buffer_size(C.c_void_p(self._h), int(which))
        if n < 0:
            self._raise(int(n))
        out = np.zeros(max(n, 0))
        self._raise(self._lib.srk_ba_download(C.c_void_p(self._h), int(which), _p(out), C.c_int64(n)))
        return out

    def rcs_rows(self, rows):
        """rows of the padded reduced camera system (full frame-variable indexing, columns <= row filled)"""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.zeros((len(rows), 10 * self._scene.M))
        self._raise(self._lib.srk_ba_download_rcs_rows(C.c_void_p(self._h), rows.ctypes.data_as(C.c_void_p),
                                                       C.c_int64(len(rows)), _p(out)))
        return out

    def dense_spd_solve(self, A, b):
        A = np.ascontiguousarray(A, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        n = A.shape[0]
        x = np.zeros(n)
        ms = C.c_double(0)
        rc = self._lib.srk_ba_dense_spd_solve(C.c_void_p(self._h), C.c_int64(n), _p(A), _p(b), _p(x), C.byref(ms))
        self._raise(rc)
        return rc == 0, x, ms.value

    # ---- helpers
    def last_error(self):
        return self._lib.srk_ba_last_error(C.c_void_p(self._h)).decode()

    def _criteria(self, term_crit):
        a = m = None
        self._keep = []
        if term_crit is not None:
            if term_crit.AllowedReprojErrRelativeChange() is not None:
                v = C.c_double(term_crit.AllowedReprojErrRelativeChange())
                self._keep.append(v)
                a = C.byref(v)
            if term_crit.MaxHessianFactor() is not None:
                v = C.c_double(term_crit.MaxHessianFactor())
                self._keep.append(v)
                m = C.byref(v)
        return a, m

    def _raise(self, rc):
        if rc >= 0:
            return
        msg = self.last_error()
        if rc == -1:
            raise ValueError("srk_ba: bad argument: " + msg)
        if rc == -3:
            raise RuntimeError("srk_ba: not possible in this state (no scene uploaded?): " + msg)
        if rc == -4:
            raise MemoryError("srk_ba: out of device memory: " + msg)
        raise RuntimeError("srk_ba: device error: " + msg)
