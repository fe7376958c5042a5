/*
 * srk_ba.h -- C ABI of the MI355X-native bundle-adjustment core (libsrk_ba.so).
 *
 * Drop-in boundary for whigg/surikatoko's BundleAdjustmentKanatani
 * (cpp_impl/suriko-engine/include/suriko/bundle-adj-kanatani.h:165-193).  The reference has no
 * FFI layer: its boundary is the C++ class API.  These entry points are what a thin adapter with
 * that class's signature binds (see include/suriko_amd/bundle-adj-kanatani.hpp and INTEGRATION.md):
 *
 *   reference (file:line)                                       C ABI
 *   ----------------------------------------------------------  ------------------------------------
 *   BundleAdjustmentKanatani::ComputeInplace  .h:179-184        srk_ba_compute_inplace
 *   BundleAdjustmentKanatani::ReprojError     .h:167-172        srk_ba_reproj_error
 *   ...::OptimizationStatusString             .h:193            srk_ba_status_string
 *   NormalizeSceneInplace / SceneNormalizer   .h:16-62          srk_ba_normalize_scene / srk_ba_revert_normalization
 *   CheckWorldIsNormalized                    .h:64-65          srk_ba_check_world_is_normalized
 *   BundleAdjustmentKanataniTermCriteria      .h:68-92          the two optional-double pointers
 *
 * Flat, caller-owned, row-major doubles; no exceptions cross the ABI; one srk_ba per thread/GPU.
 *   points      [N][3]   world coordinates, index = pnt_ind (order of reconstructed tracks, .cpp:1161-1169)
 *   cam_R,cam_T [M][9],[M][3]  world->camera ("inverse orientation", as on the reference API)
 *   K           [M][9] or [1][9] when shared_k != 0 (exactly one of shared_K / Ks in the reference, .cpp:421)
 *   obs_row_ptr [N+1] int64; obs_frame [O] int32 strictly ascending inside a point; obs_uv [O][2] pixels
 * Per-frame variable order [fx fy u0 v0 Tx Ty Tz Wx Wy Wz] (bundle-adj-kanatani.h:113-118).
 *
 * Return convention of the optimisation calls: 0 = optimised (reference `true`), 1 = not optimised
 * (reference `false`, see report.status), negative = SRK_E_* argument / device error.
 */
#ifndef SRK_BA_H
#define SRK_BA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct srk_ba srk_ba; /* opaque: owns device buffers, stream, events */

enum srk_status {
    SRK_STATUS_NONE = 0,               /* "" : normalisation failed (.cpp:681-682) or not run */
    SRK_STATUS_ABS_ERR_THRESHOLD = 1,  /* "abs err threshold"           -> true  (.cpp:749-753) */
    SRK_STATUS_SMALL_ERR_CHANGE = 2,   /* "small relative err change"   -> true  (.cpp:880-884) */
    SRK_STATUS_HESSIAN_OVERFLOW = 3,   /* "hessian overflow"            -> false (.cpp:843-847,866) */
    SRK_STATUS_ERR_CONVERGED = 4,      /* "err converged to limit value"-> false (.cpp:828-838,868) */
    SRK_STATUS_MAX_ITERATIONS = 5,     /* harness addition (the reference has no cap, .cpp:756) */
    SRK_STATUS_DEVICE_ERROR = 6        /* "device error" */
};

enum srk_error {
    SRK_OK = 0,
    SRK_E_ARGS = -1,     /* bad argument (f0 ~ 0, M < 2, NULL arrays, unsorted frames, ...) */
    SRK_E_DEVICE = -2,   /* HIP error, see srk_ba_last_error */
    SRK_E_STATE = -3,    /* staged call without an uploaded scene */
    SRK_E_NOMEM = -4
};

typedef struct srk_ba_report {
    int32_t status;          /* enum srk_status */
    int32_t optimized;       /* reference bool */
    int64_t iterations;      /* accepted outer LM iterations (.cpp:756-891) */
    int64_t attempts;        /* solve+apply+error attempts (.cpp:775-850) */
    int64_t seen;            /* observation count over all ranks (.cpp:483) */
    double err_initial;      /* (pix/f0)^2 units */
    double err_final;
    double hessian_factor;   /* value at exit */
    double world_scale;      /* SceneNormalizer::WorldScale */
    /* device time per phase, milliseconds, summed over the call (hipEvent pairs) */
    double ms_jacobian;      /* residuals + all normal-equation blocks (.cpp:1140-1448) */
    double ms_schur;         /* reduced camera system build (.cpp:1780-1908) */
    double ms_solve;         /* dense factor + solve (.cpp:1911) */
    double ms_backsub;       /* point back-substitution (.cpp:1919-1960) */
    double ms_apply;         /* apply corrections (.cpp:1997-2063) */
    double ms_error;         /* reprojection error (.cpp:410-490) */
    double ms_total;         /* wall time of the optimise call */
    int64_t schur_launches, jacobian_launches; /* kernel launches of the two HBM-bound phases */
    double ms_jacobian_kernel; /* time of the point-major Jacobian kernel alone (roofline numerator) */
    double ms_solve_syrk;      /* time inside the MFMA trailing-update kernels of the solve (srk_ba_set_profile) */
    double solve_mfma_flops;   /* flops those launches executed, summed over the call (srk_ba_set_profile) */
} srk_ba_report;

typedef struct srk_ba_normalizer {
    double R0[9], T0[3];   /* cam0 before normalisation (SceneNormalizer::prenorm_cam0_from_world) */
    double world_scale;
} srk_ba_normalizer;

/* ---- lifetime ---- */
srk_ba* srk_ba_create(int device_id);            /* NULL on failure (no HIP device, bad id) */
void srk_ba_destroy(srk_ba*);
const char* srk_ba_last_error(const srk_ba*);    /* text of the last SRK_E_DEVICE / SRK_E_ARGS */
const char* srk_ba_status_string(int status);    /* the four reference strings (+ harness additions) */
int srk_ba_device_count(void);                   /* HIP devices visible; 0 when none */
/* run on a caller-provided hipStream_t (e.g. torch's current stream) instead of the handle's own */
int srk_ba_set_stream(srk_ba*, void* hip_stream);

/* ---- multi-GPU (landmark sharding) ----
 * Every rank owns a contiguous pnt_ind range and passes only that range to the scene calls; cameras
 * are replicated.  With world_size >= 2 an LM iteration runs in ROUNDS of the next two or three damping
 * factors: every rank builds each factor's reduced camera system on its landmarks (the
 * frame blocks are added before the exchange, so they need none of their own), the packed band of
 * factor k with its rhs behind it is reduced to rank k, rank k solves and broadcasts the frame
 * corrections, every rank back-substitutes and scores every factor on its shard, and one all-reduce
 * carries the {error, solver status, point-update status} words of all factors; plus one exchange at
 * the start of an optimise call (initial error, observation count).  (SRK_MULTI_SCHEDULE=allreduce in
 * the environment of srk_ba_create: the older schedule -- the band all-reduced, every rank solves.)
 * The callback `fn` sums `count` doubles in place across ranks; it is all the library needs (a reduce
 * is a sum the other ranks ignore, a broadcast a sum of a buffer they zeroed).
 * Buffers downloaded with srk_ba_download (GRAD, UG, ...) hold this rank's partial sums.
 * `dev_ptr` is device memory; the hook is called with everything queued for the buffer complete and
 * must return after the reduction is complete.  Returns 0 on success. */
typedef int (*srk_allreduce_fn)(void* ctx, double* dev_ptr, int64_t count);
int srk_ba_set_allreduce(srk_ba*, srk_allreduce_fn fn, void* ctx, int rank, int world_size);
/* The same exchanges natively: ncclReduce / ncclBroadcast groups and ncclAllReduce (sum, fp64) on a stream of the handle,
 * ordered against the attempts' streams by events, with no host synchronisation and no Python -- what a C++ caller on the
 * 8 GPUs of a node uses (one process or thread per GPU, one handle each).  librccl.so is opened on first use.
 *   srk_ba_rccl_get_unique_id: rank 0 fills 128 bytes (ncclUniqueId) and hands them to the other ranks by any means;
 *   srk_ba_rccl_init:          every rank creates the communicator on its handle's device (collective call);
 *   srk_ba_rccl_init_second:   (optional, collective, after srk_ba_rccl_init, with a second unique id) a communicator of
 *                              its own for the second attempt slot of the `allreduce` schedule (speculative attempt pairs
 *                              with several ranks, each slot's all-reduces on its own stream and communicator); the
 *                              default schedule needs one communicator only;
 *   srk_ba_rccl_set_comm:      use a communicator (ncclComm_t) the caller owns instead; NULL detaches.
 * Either replaces a callback set with srk_ba_set_allreduce.  Call before srk_ba_upload_scene. */
int srk_ba_rccl_get_unique_id(void* id128 /* out: 128 bytes */);
int srk_ba_rccl_init(srk_ba*, const void* id128, int rank, int world_size);
int srk_ba_rccl_init_second(srk_ba*, const void* id128);
int srk_ba_rccl_set_comm(srk_ba*, void* nccl_comm, int rank, int world_size);

/* ---- the reference API, one call ---- */
int srk_ba_compute_inplace(srk_ba*, double f0,
                           int64_t n_points, double* points_xyz,
                           int32_t n_frames, double* cam_R, double* cam_T,
                           const double* K, int shared_k,
                           const int64_t* obs_row_ptr, const int32_t* obs_frame, const double* obs_uv,
                           const double* allowed_err_change /* NULL = unset */,
                           const double* max_hessian_factor /* NULL = unset */,
                           int64_t max_iterations /* <= 0 = unlimited (reference behaviour) */,
                           srk_ba_report* out);

/* The same call for a reference built with Scalar = float (rt-config.h:41-48; CMake option suriko_scalar_type_string =
 * f32, suriko-engine/CMakeLists.txt:14-15,76-82): float arrays at the boundary, widened to the fp64 pipeline and
 * rounded back (strictly more accurate than the reference's f32 arithmetic; there is no f32 device pipeline). */
int srk_ba_compute_inplace_f32(srk_ba*, float f0,
                               int64_t n_points, float* points_xyz,
                               int32_t n_frames, float* cam_R, float* cam_T,
                               const float* K, int shared_k,
                               const int64_t* obs_row_ptr, const int32_t* obs_frame, const float* obs_uv,
                               const float* allowed_err_change, const float* max_hessian_factor,
                               int64_t max_iterations, srk_ba_report* out);

double srk_ba_reproj_error(srk_ba*, double f0,
                           int64_t n_points, const double* points_xyz,
                           int32_t n_frames, const double* cam_R, const double* cam_T,
                           const double* K, int shared_k,
                           const int64_t* obs_row_ptr, const int32_t* obs_frame, const double* obs_uv,
                           int64_t* seen /* may be NULL */); /* NaN on error */

/* MultiViewIterativeFactorizer::ReprojError (multi-view-factorization.cpp:415-475): the score the MVF driver takes
 * before deciding to run BA (:372-379).  Observations whose homogeneous image point has |z| <= z_tol (reference:
 * 1e-5, :455) are skipped.  Returns 1 = ok (*reproj_err and *summands set), 0 = nothing was summed (the reference
 * returns false), negative = argument / device error.  Both scorers leave an uploaded BA scene untouched and do no
 * gauge normalisation, sorting or solver planning. */
int srk_ba_reproj_error_mvf(srk_ba*, double f0,
                            int64_t n_points, const double* points_xyz,
                            int32_t n_frames, const double* cam_R, const double* cam_T,
                            const double* K, int shared_k,
                            const int64_t* obs_row_ptr, const int32_t* obs_frame, const double* obs_uv,
                            double z_tol, double* reproj_err, int64_t* summands /* may be NULL */);

/* ---- multi-view-factorization steps (the caller on the other side of the BA path) ----
 * Estimate3DPointDepthFromFrames (multi-view-factorization.cpp:223-253, MASKS 8.44), batched over tracks: track i has
 * observations [row_ptr[i], row_ptr[i+1]) = (frame, metric homogeneous image point); its FIRST observation is the
 * base frame (:195-215) and depth_out[i] is the depth in that frame (NaN for tracks with fewer than 2 observations).
 * cam_R / cam_T: world -> camera of every frame. */
int srk_mvf_estimate_depths(srk_ba*, int64_t n_tracks, const int64_t* row_ptr, const int32_t* frame,
                            const double* x_meter /* [O][3] */, int32_t n_frames, const double* cam_R, const double* cam_T,
                            double* depth_out /* [n_tracks] */);
/* FindRelativeMotionMultiPoints (:107-189) + ProjectOntoSO3 (:79-104): camera motion anchor -> target from the common
 * points' homogeneous image coordinates in both frames and their depths in the anchor frame.  The Gram matrix of the
 * 3P x 12 system is reduced on the device; its smallest eigenvector replaces the reference's JacobiSVD (same vector up
 * to sign, and the projection is sign-invariant).  At least 6 points (each gives two independent equations of the 11
 * needed).  Returns 1 ok, 0 = projection failed (det S ~ 0), negative = error.
 * Accuracy: forming A^T A squares the condition number of A, so with NOISY tracks the null vector carries about half
 * the digits a one-sided SVD of A would (relative error ~ eps * cond(A)^2 instead of eps * cond(A)); with exact
 * geometry (the reference demo's matcher projects ground truth) both are exact to rounding -- the demo's poses agree
 * with the ground truth to 1e-11..1e-5 over its first 27 frames.  No reference fixture exists for this function
 * (parity unpinned); the oracle restates it with a one-sided Jacobi SVD and the tests use noise-free geometry. */
int srk_mvf_relative_motion(srk_ba*, int64_t n_points, const double* x_anchor /* [P][3] */, const double* x_target /* [P][3] */,
                            const double* depth_anchor /* [P] */, double* R_out /* [9] row-major */, double* T_out /* [3] */);
/* ProjectOntoSO3 alone (host code, no GPU needed): 1 ok, 0 = det S ~ 0 */
int srk_mvf_project_onto_so3(const double* R_noisy, const double* T_noisy, double* R_out, double* T_out);

/* host-side gauge normalisation (bundle-adj-kanatani.cpp:203-270); no GPU needed */
int srk_ba_normalize_scene(int64_t n_points, double* points_xyz, int32_t n_frames, double* cam_R, double* cam_T,
                           double t1y, int32_t unity_comp_ind, srk_ba_normalizer* out); /* 1 = ok, 0 = failed */
void srk_ba_revert_normalization(int64_t n_points, double* points_xyz, int32_t n_frames, double* cam_R,
                                 double* cam_T, const srk_ba_normalizer* nrm);
int srk_ba_check_world_is_normalized(int32_t n_frames, const double* cam_R, const double* cam_T, double t1y,
                                     int32_t unity_comp_ind);

/* ---- staged API: scene resident in HBM (bench, parity tests, repeated solves) ----
 * upload = gauge-normalise on the host (unless already_normalized) + copy to the device;
 * optimize = the device-resident LM loop (.cpp:720-893); download = copy back + revert normalisation. */
int srk_ba_upload_scene(srk_ba*, double f0,
                        int64_t n_points, const double* points_xyz,
                        int32_t n_frames, const double* cam_R, const double* cam_T,
                        const double* K, int shared_k,
                        const int64_t* obs_row_ptr, const int32_t* obs_frame, const double* obs_uv,
                        int already_normalized);
int srk_ba_optimize(srk_ba*, const double* allowed_err_change, const double* max_hessian_factor,
                    int64_t max_iterations, srk_ba_report* out);
int srk_ba_download_scene(srk_ba*, double* points_xyz, double* cam_R, double* cam_T, int revert_normalization);
/* restore the scene that was uploaded (device-side copy), so that a bench can repeat identical steps */
int srk_ba_reset_scene(srk_ba*);

/* single phases on the resident scene (parity tests and per-kernel timing) */
int srk_ba_phase_error(srk_ba*, double* err, int64_t* seen);
int srk_ba_phase_derivatives(srk_ba*);
int srk_ba_phase_schur(srk_ba*, double hessian_factor);
int srk_ba_phase_solve(srk_ba*);                            /* 0 ok, 1 non-finite / not positive definite */
int srk_ba_phase_backsub(srk_ba*, double hessian_factor);   /* also forms the trial scene (apply) */
int srk_ba_phase_accept(srk_ba*);                           /* trial scene becomes current */

/* copies of device buffers for the parity tests, expanded to the oracle's layouts */
enum srk_buffer {
    SRK_BUF_GRAD = 0,        /* [3N + 10M] gradE */
    SRK_BUF_POINT_BLOCKS,    /* [N][3][3] */
    SRK_BUF_FRAME_BLOCKS,    /* [M][10][10] */
    SRK_BUF_POINT_FRAME,     /* [O][3][10] */
    SRK_BUF_RCS,             /* [10M][10M] padded reduced camera system (fixed variables: identity rows) */
    SRK_BUF_RCS_RHS,         /* [10M] */
    SRK_BUF_CORRECTIONS,     /* [3N + 10M] corrections with zero gaps */
    SRK_BUF_POINTS,          /* [N][3] current (normalised) points */
    SRK_BUF_CAM_R,           /* [M][9] */
    SRK_BUF_CAM_T            /* [M][3] */
};
int64_t srk_ba_buffer_size(srk_ba*, int which);                 /* doubles; negative on error */
int srk_ba_download(srk_ba*, int which, double* dst, int64_t count);
/* selected rows of the padded reduced camera system (SRK_BUF_RCS is 12.8 GB at 4000 frames): dst[n_rows][10M], row
 * rows[k] of the system with its columns <= the row filled (the lower triangle is authoritative), zeros right of it */
int srk_ba_download_rcs_rows(srk_ba*, const int64_t* rows, int64_t n_rows, double* dst);

/* The reduced camera system is non-zero only where two frames share a landmark; the solver skips everything
 * outside that skyline (exact: Cholesky fill stays inside it).  With landmark shards every rank must be given the
 * GLOBAL covisibility after upload: min_cv[j] = smallest frame index sharing a landmark with frame j. */
int srk_ba_set_covisibility(srk_ba*, const int32_t* min_cv /* [M], NULL = dense */);
int srk_ba_set_rcs_mode(srk_ba*, int mode /* 0 = dense lower triangle, 1 = skyline as one chain,
                                              2 = skyline cut into independent chunks + separator system (default) */);
int srk_ba_rcs_chunks(srk_ba*); /* number of chunks of the current plan (0 = one chain) */
double srk_ba_rcs_fill(srk_ba*); /* skyline size / lower-triangle size */
double srk_ba_solve_mfma_flops(srk_ba*); /* flops of the MFMA trailing updates of one solve (current mode / plan) */

/* Solver launch structure (harness knob; the reference has one dense solve, bundle-adj-kanatani.cpp:1911): 1 (default) =
 * each 256-column outer step of the blocked Cholesky is ONE launch whose workgroups hand factored tiles to one another
 * (bounded spins; a timed-out hand-off makes the LM loop repeat that attempt with the unfused sequence and stay there),
 * 0 = one launch per 64-column panel and per rank-64 update.  Each is bit-reproducible from run to run; they agree with each
 * other to rounding (since round 4 the fused step eliminates the diagonal block's own rows in the factorisation's unscaled
 * form).  Takes effect at once.
 * srk_ba_solver_sync_timeouts: how many solves had to be repeated (0 in every run so far).  A timeout is a scheduling event
 * (another process on the GPU, a debugger), so the unfused sequence is kept only for the rest of that call: the next upload /
 * optimise call uses the fused step again -- until the handle has seen three timeouts since the last
 * srk_ba_set_solver_fusion(h, 1); then the unfused sequence stays.  srk_ba_solver_fusion: 1 = fused right now, 0 = unfused. */
int srk_ba_set_solver_fusion(srk_ba*, int on);
int64_t srk_ba_solver_sync_timeouts(srk_ba*);
int srk_ba_solver_fusion(srk_ba*);

/* The accepted outer iterations of the last srk_ba_optimize / srk_ba_compute_inplace call (the reference logs them with
 * VLOG, bundle-adj-kanatani.cpp:756-891): for iteration k the attempts it needed (solve + apply + error each), the host
 * time in ms since the call began at which it was accepted, the error it reached and the damping factor that was
 * accepted.  Returns the number of iterations (arrays, each may be NULL, receive at most cap entries). */
int64_t srk_ba_iteration_log(srk_ba*, int64_t cap, int32_t* attempts, double* ms_since_start, double* err,
                             double* hessian_factor);

/* Speculative attempts (default on; takes effect at the next upload): with the instrumentation off
 * (srk_ba_set_profile 0, the default) the LM loop runs the next damping factor on a second stream beside the current
 * one and judges the attempts in the reference's order, so results are those of the sequential loop; costs a second
 * reduced camera system in memory.  With several ranks every rank takes the same decisions, so the exchanges of the
 * two attempts are issued in the same order everywhere.  0 = strictly one attempt at a time -- with several ranks too: one
 * attempt slot, i.e. the all-reduce schedule without pairs (the damping-parallel schedule runs two or three factors a round). */
int srk_ba_set_speculation(srk_ba*, int on);

/* Deterministic mode (default off; takes effect at the next upload).  The reference is sequential: it adds every landmark's
 * contribution in point order (bundle-adj-kanatani.cpp:1862-1898) and two runs give the same bits.  By default the derivative
 * and the Schur kernels here combine partial sums with fp64 atomics, whose arrival order differs from run to run: results
 * agree to rounding, and where two attempts' errors tie at that level the accept / reject sequence can fork.  With the mode
 * on, a derivative task's frame sums and a run's Schur sum go to staging buffers and an ordered second pass adds them (per
 * frame in task order, per block in run order): the same scene gives the same bits every time, at the cost of that pass
 * (+ ~0.16 GB of staging per attempt slot at 1000 frames).  Covered: scenes whose tracks span at most 20 frames (the
 * run-based derivative kernel and the MFMA Schur kernel; every BASELINE configuration), one rank, fp64 run sums.
 * srk_ba_deterministic: 1 when the uploaded scene runs that way, 0 when the mode is off or the scene is not covered (the
 * default kernels then run). */
int srk_ba_set_deterministic(srk_ba*, int on);
int srk_ba_deterministic(srk_ba*);

/* Exchange schedule of the LM loop with several ranks (landmark shards; takes effect at the next upload):
 *   1 (default) damping-parallel: a round builds the next min(3, world) damping factors c, 10c, 100c on every shard, band k is
 *     REDUCED to rank k, rank k solves factor k and broadcasts its corrections, every rank scores all of them and one
 *     all-reduce carries the status words -- an iteration that needs <= 3 attempts costs about one solve;
 *   0 all-reduce: the band of each attempt is all-reduced and every rank solves the same system redundantly (round 2);
 *   2 the damping-parallel schedule at world size 1 as well (rehearsal of its collectives on one GPU).
 * The native (RCCL) form of schedule 1 issues groups of ncclReduce / ncclBroadcast rooted at different ranks on one
 * communicator.  Its FIRST round on a handle checks itself: checksums of every band and of every corrections vector travel
 * beside the rooted collectives through plain all-reduces; on a mismatch or an RCCL error the handle switches to schedule 0
 * for good, the round is repeated that way and srk_ba_last_error says why.
 * srk_ba_multi_schedule: 0 all-reduce (asked for), 1 damping-parallel (no native round has run yet), 2 damping-parallel
 * with its self-check passed, 3 all-reduce after a failed self-check. */
int srk_ba_set_multi_schedule(srk_ba*, int mode);
int srk_ba_multi_schedule(srk_ba*);

/* Internal frame order.  The reference's dense solve (bundle-adj-kanatani.cpp:1911) does not care how the frames are
 * numbered; the skyline solver, its nested dissection and the derivative kernels' frame windows here want covisible frames
 * to have nearby indices.  When the caller's numbering is far from banded (an unordered image set, a sequence that closes
 * a loop) the frames are renumbered internally by reverse Cuthill-McKee on the covisibility graph; the gauge stays on the
 * CALLER's frames 0 and 1, every download and the scene itself come back in the caller's order, results are those of the
 * caller's order up to summation order.  One rank only (landmark shards keep the caller's order).
 * mode -1 = automatic (default), 0 = never, 1 = whenever the ordering differs; takes effect at the next upload.
 * srk_ba_frame_order: 1 = renumbered (to_internal[caller's frame] = internal index, may be NULL), 0 = caller's order.
 * srk_frame_order: the decision and the numbering alone, on the host (no device needed). */
int srk_ba_set_frame_reordering(srk_ba*, int mode);
/* The numbering to use at the next upload instead of the automatic one (to_internal = NULL: automatic again).  This is how
 * landmark shards get a renumbering: the caller finds it on the WHOLE scene (srk_frame_order) and gives every rank the same
 * one; srk_ba_set_covisibility then takes min_cv in THAT numbering (covisibility of the renumbered scene). */
int srk_ba_set_frame_order(srk_ba*, const int32_t* to_internal /* [n_frames] or NULL */, int32_t n_frames);
int srk_ba_frame_order(srk_ba*, int32_t* to_internal /* [M] or NULL */);
int srk_frame_order(int mode, int64_t n_points, int32_t n_frames, const int64_t* obs_row_ptr, const int32_t* obs_frame,
                    int32_t* to_internal /* [M] */);

/* Derivative kernel selection (harness knob, the reference has one code path: bundle-adj-kanatani.cpp:1140-1448).
 * mode -1 = automatic: the run-based kernel (a lane keeps one frame's sums in registers over a run of landmarks with
 * identical frame lists) when the runs are long enough, else the per-observation kernels; 0 = per-observation kernels
 * only; 1 = run-based whenever the scene allows it (tracks of <= 64 frames, narrow frame windows); 2 = run-based over the
 * UNION of the frame lists of a run of landmarks (ragged feature tracks: a lane per (landmark, frame slot) cell, masks) whenever
 * the scene allows it (every track of <= 24 frames) -- automatic mode takes it when the uniform runs are too short to pay.
 * Takes effect at the next upload.  srk_ba_jacobian_kernel: 3 = run-based over frame unions, 2 = run-based, 1 = fused
 * per-observation, 0 = two-kernel path, -1 = no scene. */
int srk_ba_set_jacobian_mode(srk_ba*, int mode);
int srk_ba_jacobian_kernel(srk_ba*);

/* f32 storage mode (the reference's suriko_scalar_type_string = f32 switch, suriko-engine/
 * CMakeLists.txt:14-15,76-82, rt-config.h:41-48, applied where the bytes are): f32 = 1 stores the point-frame blocks W
 * (3 x 10 per observation: 240 of the 260 algorithmic bytes per observation of the derivative pass, and what the Schur
 * and back-substitution passes read; the library keeps each block as its 21 rank-2 factors) as float, 84 bytes an
 * observation; they are widened on load, every sum, the reduced camera system and the solve stay fp64.  Takes effect at the next upload.  Tolerances against the fp64 path and against the oracle with W
 * rounded the same way: tests/test_gpu_parity.py::test_f32_storage_*.  Default 0. */
int srk_ba_set_storage_precision(srk_ba*, int f32);

/* Opt-in mixed precision for the reduced camera system (the reference's suriko_scalar_type_string = f32 switch,
 * suriko-engine/CMakeLists.txt:14-15, applied where it pays on this hardware): fp32 = 1 rounds W and E^-1 W to fp32
 * when they are staged and accumulates each run of <= 128 landmarks with packed fp32 FMAs; the runs' sums, the frame
 * blocks, the factorisation and everything else stay fp64.  Default 0 = the reference's fp64 arithmetic (the only
 * mode the parity tests and the benchmark's headline use). */
int srk_ba_set_schur_precision(srk_ba*, int fp32);

/* device-time instrumentation of srk_ba_optimize / srk_ba_compute_inplace: 0 = none (default; report.ms_* stay 0
 * except ms_total), 1 = one HIP event pair per phase (fills report.ms_*), 2 = additionally event pairs around
 * every MFMA trailing-update launch (fills report.ms_solve_syrk / solve_mfma_flops).  Every event costs a few
 * microseconds on the stream, and levels >= 1 serialise the attempts (no speculation).  Default 0. */
int srk_ba_set_profile(srk_ba*, int level);

/* dense SPD solve A x = b on the device (the reduced-camera-system solver on its own; A row-major
 * n x n, lower triangle read).  returns 0 ok, 1 not positive definite, negative on error. */
int srk_ba_dense_spd_solve(srk_ba*, int64_t n, const double* A, const double* b, double* x, double* ms_factor);

/* ---- synthetic scenes (host; restates demos/demo-bundle-adj-circle-grid.cpp:86-257 and
 * src/virt-world/scene-generator.cpp:9-55, with the visibility window of SURVEY 8d) ---- */
typedef struct srk_scene_spec {
    int32_t n_frames;        /* M */
    int32_t grid_nx, grid_ny;/* N = nx * ny */
    int32_t vis_window;      /* L consecutive frames per point; <= 0 or >= M: every frame (the demo) */
    double half_extent_x, half_extent_y; /* points on [-hx,hx] x [-hy,hy] */
    double f0;               /* 600 */
    double noise_x3d_hi;     /* 0.005 */
    double noise_r_hi;       /* 0.005 */
    double noise_uv_pix;     /* 0 (the demo projects exactly) */
    uint32_t seed;           /* 1234 */
} srk_scene_spec;
int64_t srk_scene_num_observations(const srk_scene_spec*);
/* fills caller arrays: points [N][3] (noisy), points_gt [N][3] (may be NULL), cam_R/T [M][..] (noisy),
 * cam_R_gt/cam_T_gt (may be NULL), K [M][9], row_ptr [N+1], obs_frame [O], obs_uv [O][2] */
int srk_scene_generate(const srk_scene_spec*, double* points, double* points_gt, double* cam_R, double* cam_T,
                       double* cam_R_gt, double* cam_T_gt, double* K, int64_t* row_ptr, int32_t* obs_frame,
                       double* obs_uv);
void srk_circle_camera_shots(const double center[3], double radius, double ascent_z, int32_t n,
                             const double* angles, double* cam_R, double* cam_T);

/* ---- callers of the BA path: the dinosaur loader and its pre-processing (host only) ----
 * srk_read_matrix_file          ReadMatrixFromFile        cpp_impl/suriko-engine/src/mat-serialization.cpp:12-87
 * srk_decompose_proj_mat        DecomposeProjMat          cpp_impl/suriko-engine/src/obs-geom.cpp:606-677
 * srk_triangulate_least_squares Triangulate3DPointByLeastSquares              obs-geom.cpp:679-727
 * srk_dino_load                 DinoDemo's scene build    cpp_impl/demos/demo-bundle-adj-dinosaur.cpp:24-54,85-230
 * all return 1 on success, 0 on failure (err receives the reference's message where it has one). */
int srk_read_matrix_file(const char* path, char delimiter, double* data /* NULL = count only */, int64_t capacity,
                         int64_t* rows, int64_t* cols, char* err, int errlen);
int srk_decompose_proj_mat(const double* P /* [3][4] */, double* scale_factor, double* K /* [9] */,
                           double* R_direct /* [9] */, double* T_direct /* [3] */);
int srk_triangulate_least_squares(int32_t n, const double* uv /* [n][2] */, const double* P /* [n][12] */, double f0,
                                  double* X /* [3] */);
int srk_dino_load(const char* dir, double f0, int64_t* n_points, int32_t* n_frames, int64_t* n_obs,
                  double* points /* NULL = sizes only */, double* cam_R, double* cam_T, double* K, int64_t* row_ptr,
                  int32_t* obs_frame, double* obs_uv, char* err, int errlen);

#ifdef __cplusplus
}
#endif
#endif
