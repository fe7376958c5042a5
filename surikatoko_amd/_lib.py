"""ctypes loader of libsrk_ba.so (built in-tree by surikatoko_amd/csrc/Makefile)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRK_BA_LIBRARY: development only (tools/*.sh load ablation / stamp builds from a temporary path, so that the product
# library in the tree is never overwritten)
_PATH = os.environ.get("SRK_BA_LIBRARY") or os.path.join(_HERE, "libsrk_ba.so")
_lib = None


class LibraryNotBuilt(RuntimeError):
    pass


def library_path():
    return _PATH


def build_library(jobs=4):
    """hipcc --offload-arch=gfx950 build of every HIP translation unit (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j", str(jobs), "-C", os.path.join(_HERE, "csrc")])
    return _PATH


class Report(C.Structure):
    _fields_ = [("status", C.c_int32), ("optimized", C.c_int32), ("iterations", C.c_int64),
                ("attempts", C.c_int64), ("seen", C.c_int64), ("err_initial", C.c_double),
                ("err_final", C.c_double), ("hessian_factor", C.c_double), ("world_scale", C.c_double),
                ("ms_jacobian", C.c_double), ("ms_schur", C.c_double), ("ms_solve", C.c_double),
                ("ms_backsub", C.c_double), ("ms_apply", C.c_double), ("ms_error", C.c_double),
                ("ms_total", C.c_double), ("schur_launches", C.c_int64), ("jacobian_launches", C.c_int64),
                ("ms_jacobian_kernel", C.c_double), ("ms_solve_syrk", C.c_double),
                ("solve_mfma_flops", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Normalizer(C.Structure):
    _fields_ = [("R0", C.c_double * 9), ("T0", C.c_double * 3), ("world_scale", C.c_double)]


class SceneSpecC(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("grid_nx", C.c_int32), ("grid_ny", C.c_int32), ("vis_window", C.c_int32),
                ("half_extent_x", C.c_double), ("half_extent_y", C.c_double), ("f0", C.c_double),
                ("noise_x3d_hi", C.c_double), ("noise_r_hi", C.c_double), ("noise_uv_pix", C.c_double),
                ("seed", C.c_uint32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)

# every symbol include/srk_ba.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "srk_ba_create", "srk_ba_destroy", "srk_ba_last_error", "srk_ba_status_string", "srk_ba_device_count",
    "srk_ba_set_stream", "srk_ba_set_allreduce", "srk_ba_rccl_get_unique_id", "srk_ba_rccl_init", "srk_ba_rccl_init_second", "srk_ba_rccl_set_comm", "srk_ba_compute_inplace", "srk_ba_compute_inplace_f32", "srk_ba_reproj_error", "srk_ba_reproj_error_mvf", "srk_mvf_estimate_depths", "srk_mvf_relative_motion",
    "srk_mvf_project_onto_so3", "srk_ba_set_schur_precision", "srk_ba_set_storage_precision", "srk_ba_set_speculation", "srk_ba_set_deterministic", "srk_ba_deterministic", "srk_ba_set_multi_schedule", "srk_ba_multi_schedule", "srk_ba_set_frame_reordering", "srk_ba_set_frame_order", "srk_ba_frame_order", "srk_frame_order", "srk_ba_set_solver_fusion", "srk_ba_solver_sync_timeouts", "srk_ba_iteration_log", "srk_ba_solver_fusion", "srk_ba_set_jacobian_mode", "srk_ba_jacobian_kernel",
    "srk_ba_normalize_scene", "srk_ba_revert_normalization", "srk_ba_check_world_is_normalized",
    "srk_ba_upload_scene", "srk_ba_optimize", "srk_ba_download_scene", "srk_ba_reset_scene", "srk_ba_phase_error",
    "srk_ba_phase_derivatives", "srk_ba_phase_schur", "srk_ba_phase_solve", "srk_ba_phase_backsub",
    "srk_ba_phase_accept", "srk_ba_buffer_size", "srk_ba_download", "srk_ba_download_rcs_rows", "srk_ba_set_profile", "srk_ba_dense_spd_solve",
    "srk_ba_set_covisibility", "srk_ba_set_rcs_mode", "srk_ba_rcs_fill", "srk_ba_solve_mfma_flops", "srk_ba_rcs_chunks",
    "srk_scene_num_observations", "srk_scene_generate", "srk_circle_camera_shots",
    "srk_read_matrix_file", "srk_decompose_proj_mat", "srk_triangulate_least_squares", "srk_dino_load",
]


def lib():
    """Load libsrk_ba.so; raises LibraryNotBuilt (never falls back to anything else)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever is
    # loaded first serves both, and torch refuses to see the GPU when it is not its own.  Importing torch first makes
    # device pointers of this library usable by torch.distributed / RCCL (surikatoko_amd/dist.py).  Pure C/C++ users
    # of libsrk_ba.so never load torch and run on the ROCm runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(_PATH):
        raise LibraryNotBuilt(
            f"{_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C surikatoko_amd/csrc).  There is no CPU fallback.")
    L = C.CDLL(_PATH)
    L.srk_ba_create.restype = C.c_void_p
    L.srk_ba_create.argtypes = [C.c_int]
    L.srk_ba_destroy.argtypes = [C.c_void_p]
    L.srk_ba_destroy.restype = None
    L.srk_ba_last_error.restype = C.c_char_p
    L.srk_ba_last_error.argtypes = [C.c_void_p]
    L.srk_ba_status_string.restype = C.c_char_p
    L.srk_ba_status_string.argtypes = [C.c_int]
    L.srk_ba_reproj_error.restype = C.c_double
    L.srk_ba_reproj_error_mvf.restype = C.c_int
    L.srk_ba_buffer_size.restype = C.c_int64
    L.srk_ba_buffer_size.argtypes = [C.c_void_p, C.c_int]
    L.srk_scene_num_observations.restype = C.c_int64
    L.srk_ba_rcs_fill.restype = C.c_double
    L.srk_ba_rcs_fill.argtypes = [C.c_void_p]
    L.srk_ba_solve_mfma_flops.restype = C.c_double
    L.srk_ba_solve_mfma_flops.argtypes = [C.c_void_p]
    L.srk_ba_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.srk_ba_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int]
    L.srk_ba_revert_normalization.restype = None
    L.srk_circle_camera_shots.restype = None
    _lib = L
    return L
