// srk_dev.hpp -- internal declarations shared by the HIP translation units of libsrk_ba.so.
// Device data layout (all fp64 unless noted; see DESIGN.md "Data layout in HBM"):
//   pts      [N][3]            landmark coordinates (normalised world), two buffers: current / trial
//   cam      [M][SRK_CAM_PACK] per-frame pack recomputed from (R,T,K) once per pose change:
//              0-8 R  9-11 T  12-20 K  21-29 K*R  30-32 Td (direct translation)
//              33-35 rot1  36-38 rot2  39-41 rot3  (bundle-adj-kanatani.cpp:1511-1513)
//              42 1/fx  43 u0/(f0 fx)  44 1/fy  45 v0/(f0 fy)  46 1/f0  47 f0
//   obs      point-major CSR: row_ptr[N+1] i64, obs_frame[O] i32, obs_pt[O] i32, obs_uv[O][2]
//            frame-major copy:  col_ptr[M+1] i64, fobs_pt[O] i32, fobs_uv[O][2]
//   W        fp64 storage (default): [21][Os] the rank-2 FACTORS of the point-frame blocks (SRK_WF_* below), structure-of-
//            arrays, plane k of observation o at W[k*Os + o] (Os = O rounded up to 64): every store/load is lane-contiguous;
//            f32 storage mode: the same 21 planes as floats ([21][Os], plane k of observation o at W[k*Os + o]); widened on load
//   Vg       [9][Ns]   per point: V00 V01 V02 V11 V12 V22 g0 g1 g2 (SoA, Ns = N rounded up to 64)
//   Ug       [M][65]   per frame: 55 upper-triangle entries of the 10x10 block (row-major order) + 10 gradient
//   S        [ld][ld]  padded reduced camera system, row-major, LOWER triangle authoritative;
//            variable index = 10*frame + var; the 7 gauge-fixed variables and the padding rows
//            carry an identity diagonal and zero rhs, so their correction is exactly 0
//   rhs,dc   [ld]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#define SRK_CAM_PACK 48
#define SRK_UG 65
// fp64 storage of the point-frame blocks: the rank-2 factors of W[pv][fv] = Ap[pv] Af[fv] + Bp[pv] Bf[fv] as SoA planes
// (srk_ba_kernels.hip: "storage of the point-frame blocks"); Af[1] = Af[3] = Bf[0] = Bf[2] = 0, Af[2] = Bf[3] = G
#define SRK_WF_AP 0   // planes 0..2   Ap[0..2]
#define SRK_WF_BP 3   // planes 3..5   Bp[0..2]
#define SRK_WF_AF0 6  // Af[0]
#define SRK_WF_G 7    // Af[2] = Bf[3]
#define SRK_WF_AF4 8  // planes 8..13  Af[4..9]
#define SRK_WF_BF1 14 // Bf[1]
#define SRK_WF_BF4 15 // planes 15..20 Bf[4..9]
#define SRK_WF_PLANES 21

struct SrkDims {
    int64_t N, O, Os, Ns;
    int32_t M;
    int64_t ld;          // padded RCS dimension (multiple of SRK_CHOL_NB)
    int32_t comp;        // unity_t1_comp_ind (gauge), 1 by default
    int32_t w_f32;       // 1: the point-frame blocks W are stored as float (W then points at floats), arithmetic stays fp64
    int32_t g0, g1;      // internal indices of the caller's frames 0 and 1 (the gauge-fixed ones): 0, 1 unless the frames were reordered
};

#define SRK_CHOL_NB 256 // outer panel of the blocked Cholesky; ld is a multiple of it

// ---- deterministic mode (srk_ba_set_deterministic): the sums that take fp64 atomics by default go through staging buffers and
// an ordered second pass instead.  Derivative kernel: a task's 65 frame sums per frame slot, then per frame the tasks' sums in
// task order.  Schur kernel: a run's sum as the 10 x 10 blocks (slot a >= slot b) of its lower block triangle and its
// right-hand-side terms, then per block / per frame the runs' contributions in run order.
#define SRK_DET_LD 200       // local variables of a run: SRK_WS_NF_HOST frames x 10
#define SRK_DET_STRIDE 21000 // doubles of a run's staged sum: 20 * 21 / 2 blocks of 100
struct SrkDetJac {
    double* stage;            // [task][64][SRK_UG]
    const int32_t* ptr;       // [M + 1]
    const int32_t* ent;       // task * 64 + frame slot, by frame, tasks ascending
};
struct SrkDetSchur {
    double* stage;            // [run][SRK_DET_STRIDE]
    double* stage_rhs;        // [run][SRK_DET_LD]
    const int32_t *pair_ptr, *pair_fa, *pair_fb, *pair_ent; // blocks (fa >= fb) that receive something; ent: run | slot a << 20 | slot b << 25
    int32_t n_pairs;
    const int32_t *f_ptr, *f_ent; // per frame: run | slot << 20
};

// ---- BA kernels (srk_ba_kernels.hip) ----
void srk_launch_cam_pack(hipStream_t s, int32_t M, const double* R, const double* T, const double* K, double f0,
                         double* pack);
void srk_launch_jac_points(hipStream_t s, const SrkDims& d, const double* pts, const double* cam,
                           const int32_t* obs_frame, const int32_t* obs_pt, const double* obs_uv, double* W,
                           double* Vg);
// fused single pass (point blocks + frame blocks); usable when every workgroup's frame range fits SRK_JF_SLOTS_HOST
#define SRK_JF_OBS_HOST 1024
#define SRK_JF_SLOTS_HOST 48
#define SRK_JF_PMAX_HOST 448
void srk_launch_jac_fused(hipStream_t s, const SrkDims& d, const double* pts, const double* cam,
                          const int32_t* obs_frame, const int32_t* obs_pt, const double* obs_uv, double* W,
                          double* Vg, double* Ug, const int32_t* wg_jmin);
// run-based single pass: one wave per task = consecutive landmarks with identical frame lists (nf <= 64 frames), about
// SRK_JR_TASK_PTS_MIN_HOST .. MAX_HOST of them; four consecutive tasks (one workgroup) must touch fewer than
// SRK_JF_SLOTS_HOST consecutive frames
#define SRK_JR_TASK_PTS_MIN_HOST 12
#define SRK_JR_TASK_PTS_MAX_HOST 96 // = SRK_JR_XMAX of the kernel
void srk_launch_jac_runs(hipStream_t s, const SrkDims& d, const double* pts, const double* cam, const int64_t* row_ptr,
                         const int32_t* obs_frame, const double* obs_uv, double* W, double* Vg, double* Ug,
                         const int32_t* task_first, const int32_t* task_count, int32_t n_tasks, const int32_t* wg_jmin,
                         const int32_t* task_group /* NULL: uniform runs; else the Schur run (grp_*) each task is a piece of */,
                         const int32_t* grp_nf, const int32_t* grp_frames, const uint32_t* pt_mask,
                         const SrkDetJac* det = nullptr /* deterministic mode */,
                         int frames_stride = 24 /* row length of grp_frames: SRK_GRP_MAXNF_HOST (the Schur runs) or 32 (the derivative kernel's own) */);
void srk_launch_jac_frames(hipStream_t s, const SrkDims& d, int64_t max_frame_obs, const double* pts,
                           const double* cam, const int64_t* col_ptr, const int32_t* fobs_pt, const double* fobs_uv,
                           double* Ug);
void srk_launch_schur(hipStream_t s, const SrkDims& d, double c, const int64_t* row_ptr, const int32_t* obs_frame,
                      const double* W, const double* Vg, double* S, double* rhs, const int32_t* pt_list,
                      int64_t n_list);
#define SRK_GRP_MAXNF_HOST 24   // must match SRK_GRP_MAXNF in srk_ba_kernels.hip
#define SRK_GRP_MAXPTS_HOST 128 // landmarks per workgroup run
#define SRK_GRP_NF1_HOST 21     // must match SRK_GRP_NF1
#define SRK_WS_NF_HOST 20       // must match SRK_WS_NF (runs the MFMA kernel k_schur_mm takes)
void srk_launch_schur_grouped(hipStream_t s, const SrkDims& d, double c, const int64_t* row_ptr, const int32_t* obs_pt,
                              const uint8_t* obs_slot /* [O] slot of the observation's frame in its run's frame set */,
                              const uint32_t* pt_mask /* [N] slots a landmark sees */, const double* W, const double* Vg,
                              double* S, double* rhs, const int32_t* grp_first, const int32_t* grp_count,
                              const int32_t* grp_nf /* size of the run's frame set; negative = ragged run */,
                              const int32_t* grp_frames /* [n_groups][SRK_GRP_MAXNF_HOST] */, int64_t n_groups,
                              int64_t n_wide /* runs with more than SRK_GRP_NF1_HOST frames */,
                              int64_t n_mid /* runs with SRK_WS_NF_HOST < frames <= SRK_GRP_NF1_HOST */,
                              int fp32_accumulate /* 0 = fp64 (reference arithmetic), 1 = packed fp32 run sums */,
                              int32_t* irr /* [0] count + list of landmarks k_schur_mm hands back to the inverse path */,
                              int64_t n_mm_uniform, int64_t n_mm_ragged /* runs of <= SRK_WS_NF_HOST frames by kind */,
                              const SrkDetSchur* det = nullptr /* deterministic mode (every run must be one of those) */);
// tracks longer than SRK_GRP_MAXNF_HOST frames: runs of <= SRK_LONG_PTS_HOST landmarks over a frame set of
// <= SRK_LONG_MAXNF_HOST frames, one workgroup per pair of 8-frame blocks (k_schur_long); longer tracks stay with k_schur.
// (Round 3: 4096 -- the limit is only the row length of the run_frames table; it was 256, and a track over more frames fell
// back to the per-landmark global-atomics kernel, a 30x cliff on all-visible scenes of more than 256 frames.)
#define SRK_LONG_PTS_HOST 128
#define SRK_LONG_MAXNF_HOST 4096
#define SRK_LONG_FB_HOST 8
void srk_launch_schur_long(hipStream_t s, const SrkDims& d, double c, const double* W, const double* Vg, double* S, double* rhs,
                           const int32_t* item /* [n_items][4]: run, row block, column block (<= row block), 0 */,
                           int64_t n_items, const int32_t* run_np, const int32_t* run_nf,
                           const int32_t* run_pts /* [run][SRK_LONG_PTS_HOST] landmarks (internal order) */,
                           const int32_t* run_frames /* [run][SRK_LONG_MAXNF_HOST] the run's frame set, ascending */,
                           const int64_t* run_obs_off, const int32_t* run_obs /* [off + landmark * fb ceil(nf / fb) + slot]: observation or -1 */,
                           int fb = SRK_LONG_FB_HOST /* frames per block: 8 or 16 */);
void srk_launch_assemble(hipStream_t s, const SrkDims& d, double c, const double* Ug, double* S, double* rhs,
                         double ident /* diagonal of fixed / padding variables */, const int64_t* row_ptr,
                         const int32_t* obs_frame, const double* W, const double* Vg,
                         const int32_t* irr /* landmarks handed back by k_schur_mm: served by the tail workgroups */);
void srk_launch_backsub(hipStream_t s, const SrkDims& d, double c, const int32_t* obs_frame, const int32_t* obs_pt,
                        const double* W, const double* Vg, const double* dc, double* acc, const double* pts,
                        double* pts_trial, double* dx);
void srk_launch_cam_apply(hipStream_t s, int32_t M, const double* R, const double* T, const double* dc, double* Rn,
                          double* Tn, const double* K, double f0, double* pack /* camera packs of the new poses, or NULL */);
void srk_launch_error(hipStream_t s, const SrkDims& d, const double* pts, const double* cam,
                      const int32_t* obs_frame, const int32_t* obs_pt, const double* obs_uv, double* partial,
                      int32_t n_partial, double* err_out,
                      const int32_t* wg_jmin /* fused-Jacobian frame windows, or NULL: gather the cameras */,
                      int* info = nullptr, int* info2 = nullptr /* given: packed into err_out[1..2] and cleared */);
int32_t srk_error_partials(const SrkDims& d);
int64_t srk_error_partials_staged(const SrkDims& d); // partial sums written when wg_jmin is given
void srk_launch_error_score(hipStream_t s, int64_t O, const double* pts, const double* cam, const int32_t* obs_frame,
                            const int32_t* obs_pt, const double* obs_uv, double z_tol /* < 0: keep every observation */,
                            double* partial /* 2 n_partial */, int32_t n_partial, double* out2 /* {error, count} */);
void srk_launch_expand_ug(hipStream_t s, int32_t M, const double* Ug, double* U_full, double* g_full);
void srk_launch_symmetrize(hipStream_t s, int64_t n, int64_t ld, double* S);

void srk_launch_env_zero(hipStream_t s, int64_t ld, const int64_t* env_col, double* S, double* rhs /* zeroed too, or null */,
                         int32_t* irr = nullptr /* hand-back counter of k_schur_mm, cleared too */);
void srk_launch_env_pack(hipStream_t s, int64_t ld, const int64_t* env_col, const int64_t* env_off, double* S,
                         double* packed, int dir);
void srk_launch_band_pack(hipStream_t s, int64_t ld, const int64_t* band_col, const int64_t* band_off, double* S,
                          double* packed, int dir /* 0 pack, 1 unpack */);

void srk_launch_checksum(hipStream_t s, const double* p, int64_t n, double* part /* 512 doubles of scratch */, double* out2 /* {sum, sum |.|} */);

// ---- multi-view-factorization steps (srk_ba_kernels.hip) ----
void srk_launch_mvf_depth(hipStream_t s, int64_t n_tracks, const int64_t* row_ptr, const int32_t* frame, const double* x_meter,
                          const double* cam_R, const double* cam_T, double* depth);
void srk_launch_mvf_gram(hipStream_t s, int64_t n_points, const double* x_anchor, const double* x_target, const double* depth,
                         double* partial /* [ceil(P / 256)][78] upper triangle of A^T A, row by row */);

// optional profile of one solve: event pairs around every MFMA trailing-update launch (n pairs recorded, at most
// cap / 2) and the flops those launches execute; dry = count only, launch nothing
struct SrkSolveProf {
    hipEvent_t* ev = nullptr;
    size_t cap = 0, n = 0;
    double flops = 0;
    bool dry = false;
};

// state of the fused outer-step kernel's in-launch hand-offs (k_step256): flag words (device, zeroed once, 16 per batch
// item) that hold the epoch of the launch that set them, and the host's launch counter.  One per stream.
#define SRK_SYNC_WORDS (16 * 32) // 16 words x SRK_MAX_CHUNKS items
struct SrkCholSync {
    unsigned* flags = nullptr;
    unsigned epoch = 0;
    bool fused = true; // false: the unfused k_panel / k_upd64 sequence
};

// ---- dense SPD solver (srk_chol.hip) ----
// In-place blocked Cholesky of the lower triangle of A (row-major, ld x ld, ld % SRK_CHOL_NB == 0) with the forward
// substitution folded in, then the backward substitution.  w: rhs (destroyed), y: scratch, x: solution.
// info (device int) is set non-zero when a pivot is not positive/finite or the solution is not finite.
// row_end / col_begin: optional host arrays describing the skyline of A (see srk_chol.hip); NULL = dense.
void srk_chol_solve(hipStream_t s, int64_t ld, double* A, double* w, double* y, double* x, int* d_info,
                    const int64_t* row_end, const int64_t* col_begin, double* dinv /* (ld / 64) * 4096 doubles */,
                    struct SrkSolveProf* prof /* may be NULL */, struct SrkCholSync* sync /* NULL: unfused kernels */,
                    int64_t n_real = 0 /* > 0: rows / columns from there on are padding (identity diagonal, zero rhs) */);

// ---- chunked (bordered block-diagonal) solve of a banded reduced camera system (srk_chol.hip) ----
#include <vector>
#define SRK_MAX_CHUNKS 32
#define SRK_MAX_SEPW 1024 // widest separator (= covisibility bandwidth in variables) the chunked solve takes on
struct SrkChunkPlan {
    int P = 0;                       // number of chunks; < 2 = not used
    int64_t sepw = 256;              // separator width (variables), >= bandwidth
    int64_t a[SRK_MAX_CHUNKS]{};     // first global variable of chunk c
    int64_t n[SRK_MAX_CHUNKS]{};     // interior size of chunk c (multiple of 256)
    int64_t ldc[SRK_MAX_CHUNKS]{};   // n + 2 sepw
    double* Ac[SRK_MAX_CHUNKS]{};    // chunk matrices (ldc x ldc)
    double* wc[SRK_MAX_CHUNKS]{};
    double* yc[SRK_MAX_CHUNKS]{};
    double* xc[SRK_MAX_CHUNKS]{};
    double* dinvc[SRK_MAX_CHUNKS]{};
    std::vector<int64_t> row_end[SRK_MAX_CHUNKS], col_begin[SRK_MAX_CHUNKS]; // local skylines
    int64_t lds = 0;                 // separator system size = sepw (P - 1)
    double *Cs = nullptr, *ws = nullptr, *ys = nullptr, *xs = nullptr, *dinvs = nullptr;
    std::vector<int64_t> s_row_end, s_col_begin;
    int64_t* d_sep_start = nullptr;  // device: first global variable of separator c
    int64_t* d_sep_env = nullptr;    // device: skyline (first column per 128-row tile) of the separator system
    SrkChunkPlan* child = nullptr;   // plan of the separator system itself (nested dissection); NULL = direct solve
};
void srk_chol_solve_chunked(hipStream_t s, const SrkChunkPlan& pl, int64_t ld, const double* S, const double* rhs,
                            double* x, const int64_t* d_env_col, int* d_info, struct SrkSolveProf* prof /* may be NULL */,
                            struct SrkCholSync* sync /* NULL: unfused kernels */);
