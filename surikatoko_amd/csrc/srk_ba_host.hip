// srk_ba_host.hip -- host side of libsrk_ba.so: the C ABI of include/srk_ba.h, the gauge normalisation, the
// device-resident Levenberg-Marquardt loop (two attempt slots: the next damping factor runs speculatively beside the
// current one), the nested-dissection plan of the reduced camera system and the buffer management.
//
// Mirrors whigg/surikatoko cpp_impl/suriko-engine/src/bundle-adj-kanatani.cpp:
//   ComputeInplace :617-718, ComputeOnNormalizedWorld :720-893 (LM control), SceneNormalizer :123-333.
// There is NO CPU fallback: every compute call needs a HIP device and fails loudly (SRK_E_DEVICE) without one.
#include "../../include/srk_ba.h"
#include "srk_dev.hpp"
#include "srk_geom.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <numeric>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <iterator>
#include <memory>
#include <vector>
#include <functional>
#include <unordered_set>
#include <thread>

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

} // namespace

#define SRK_FUSION_RETRIES 3
#define SRK_SLOTS 3 // attempt slots: two on one GPU (speculative pairs), up to three with several ranks (one damping factor each)
struct srk_ba {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string last_error;

    // scene
    bool have_scene = false;
    SrkDims d{};
    double f0 = 0;
    int64_t max_frame_obs = 0;
    srk_ba_normalizer nrm{};
    bool normalized_on_upload = false;

    // device buffers
    DevBuf pts[SRK_SLOTS + 1], camR[SRK_SLOTS + 1], camT[SRK_SLOTS + 1], K, cam[SRK_SLOTS + 1]; // the current scene + one trial scene per attempt slot
    DevBuf pts0, camR0, camT0; // copy of the uploaded (normalised) scene for srk_ba_reset_scene
    DevBuf row_ptr, obs_frame, obs_pt, obs_uv, col_ptr, fobs_pt, fobs_uv;
    DevBuf W, Vg, Ug, scratch;
    // landmarks are stored sorted by frame list (internal order); perm[internal] = caller's pnt_ind
    std::vector<int64_t> perm, row_ptr_user, row_ptr_int;
    // Frames may be stored in another order than the caller's (frame_reorder below: unordered image sets, loop
    // closures).  Both empty = the caller's order.  frame_int[caller's frame] = internal index, frame_user = its inverse;
    // obs_rank[caller's observation] = its place inside its landmark's internal (re-sorted) observation list.
    std::vector<int32_t> frame_int, frame_user, obs_rank;
    int frame_order_mode = -1; // srk_ba_set_frame_reordering: -1 automatic, 0 never, 1 whenever the ordering differs from the caller's
    std::vector<int32_t> frame_order_given; // srk_ba_set_frame_order: the numbering to use (several ranks: the same on every rank)
    bool frame_order_supplied = false;      // the uploaded scene uses frame_order_given
    DevBuf grp_first, grp_count, grp_nf, grp_frames, obs_slot, pt_mask, gen_list, wg_jmin;
    // long tracks (more than SRK_GRP_MAXNF_HOST frames): runs over frame-block pairs, k_schur_long
    DevBuf lg_item, lg_np, lg_nf, lg_pts, lg_frames, lg_obs_off, lg_obs;
    int64_t n_long_items = 0, n_long_runs = 0;
    DevBuf sc_pts, sc_R, sc_T, sc_K, sc_cam, sc_frame, sc_pt, sc_uv, sc_partial, sc_out; // standalone scoring path
    int64_t n_groups = 0, n_groups_wide = 0, n_groups_mid = 0, n_generic = 0;
    int64_t n_mm_uniform = 0, n_mm_ragged = 0; // runs the MFMA kernel takes (<= SRK_WS_NF_HOST frames), by kind
    bool jac_fused = false; // every 1024-observation workgroup touches < SRK_JF_SLOTS_HOST consecutive frames
    // run-based Jacobian kernel (k_jac_runs): tasks = pieces of runs of landmarks with identical frame lists
    DevBuf jr_first, jr_count, jr_jmin, jr_group;
    // the derivative kernel's OWN runs (round 4): when a scene holds tracks over more than SRK_GRP_MAXNF_HOST frames the Schur
    // kernels' runs do not cover every landmark; runs over unions of <= 32 frames (one mask word) built for the derivative kernel
    // alone do, as long as no track is longer than that
    DevBuf jd_nf, jd_frames, jd_mask;
    bool jr_own_runs = false;
    // deterministic mode (srk_ba_set_deterministic; srk_dev.hpp: SrkDetJac / SrkDetSchur): index tables of the ordered second
    // passes and the derivative kernel's staging buffer (the Schur kernel's are per attempt slot)
    bool deterministic = false;       // asked for (takes effect at the next upload)
    bool det_active = false;          // the uploaded scene runs that way (every landmark through k_jac_runs / k_schur_mm)
    DevBuf dj_ptr, dj_ent, dj_stage, ds_pair_ptr, ds_pair_fa, ds_pair_fb, ds_pair_ent, ds_f_ptr, ds_f_ent;
    int32_t ds_n_pairs = 0;
    bool jac_runs_masked = false; // the tasks are pieces of the Schur kernel's runs over UNIONS of frame lists (ragged tracks)
    int32_t jr_tasks = 0, jr_min_nf = 64;
    int long_fb = SRK_LONG_FB_HOST; // frames per block of k_schur_long's pairs: 8, or 16 when the scene has enough of them
    bool jac_runs = false;  // the tasks are long enough to pay and every workgroup's frame window fits
    int jac_mode = -1;      // -1 = automatic, 0 = never k_jac_runs, 1 = whenever possible (srk_ba_set_jacobian_mode)
    // skyline of the reduced camera system (see k_env_zero): host + device copies
    std::vector<int32_t> min_cv;                       // [M] smallest frame sharing a landmark with frame j
    std::vector<int64_t> env_col_h, env_off_h, row_end_h, col_begin_h;
    DevBuf env_col, env_off, band_col, band_off;
    int64_t env_packed = 0, band_packed = 0; // doubles inside the factorisation skyline / the pre-factorisation band
    bool use_envelope = true;
    // chunked solve of a banded system (srk_chol.hip): plan + its buffers
    bool use_chunks = true;
    // Everything one LM attempt writes lives in an attempt slot: the reduced camera system and its solver plan, the
    // corrections, the trial scene, the status words, and the stream it runs on.  Two slots let the loop run the next
    // damping factor speculatively beside the current one (the solve is a latency chain that leaves the chip idle).
    struct Attempt {
        DevBuf S, rhs, wy, dc, acc, dx, err_partial, info, dinv, packed, sync_flags;
        DevBuf irr; // [0] count + landmarks the SYRK form of k_schur_mm hands back to the per-landmark inverse path
        DevBuf det_stage, det_rhs; // deterministic mode: the runs' staged sums (SrkDetSchur)
        SrkChunkPlan plan;
        SrkCholSync sync;            // in-launch hand-offs of the fused outer-step kernel (srk_chol.hip: k_step256)
        std::vector<DevBuf> plan_bufs;
        std::vector<char> plan_zeroed;   // plan_bufs[i] is a matrix / vector that must be zero outside what a solve writes
        std::vector<std::unique_ptr<SrkChunkPlan>> plan_children; // plans of the nested separator systems
        std::vector<int64_t> plan_sig;   // what the plan was built for (build_chunk_plan keeps it when the next scene's skyline is the same)
        SrkSolveProf solve_prof;     // event pairs / flops of the last profiled solve
        double* host_back = nullptr; // pinned: {error, solver info, point-update info} of one attempt
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        hipEvent_t ev_a = nullptr, ev_b = nullptr; // cross-stream hand-offs of the damping-parallel schedule
        double* err_dst = nullptr;   // {error, solver info, point-update info} of this slot: 8 doubles inside srk_ba::status_all
        int trial = 1;               // index of this slot's trial scene buffers
        bool allocated = false;
    };
    Attempt att[SRK_SLOTS];
    Attempt* A = &att[0]; // the slot the phase functions work on (select_attempt)
    hipStream_t main_stream = nullptr; // = att[0].stream
    hipEvent_t ev_jac = nullptr;       // derivatives done (the second slot's stream waits for it)
    bool speculate = true;             // single rank, instrumentation off: run two damping factors side by side
    int cur = 0; // index of the current scene buffers

    // multi-GPU exchange
    srk_allreduce_fn allreduce = nullptr;
    void* allreduce_ctx = nullptr;
    int rank = 0, world = 1;
    // native exchange: RCCL (librccl.so, loaded on first use) all-reduces on the stream of the attempt that needs them --
    // no host round trip, no Python.  comm_owned: created by srk_ba_rccl_init (destroyed with the handle).
    ncclComm_t comm = nullptr;
    bool comm_owned = false;
    // a second communicator for the second attempt slot (srk_ba_rccl_init_second): with it the speculative pairs stay on
    // with several ranks -- each slot's all-reduces run on its own stream AND its own communicator, so the two slots'
    // collectives never share a communicator's queue.  Without it (or with the all-reduce callback, which blocks the
    // host) pairs stay on as well when spec_multi allows: the callback serialises the exchanges on the host.
    ncclComm_t comm2 = nullptr;
    bool spec_multi = true; // SRK_MULTI_SPECULATION=0: one attempt at a time with several ranks
    // Damping-parallel schedule (world >= 2): an iteration's attempts c, 10c, 100c are built by every rank on its
    // shard, band k is REDUCED to rank k, rank k solves factor k and broadcasts its corrections, every rank scores all of
    // them.  All collectives of that schedule go through ONE communicator on ONE stream (comm_stream) in one program order.
    bool dp_schedule = true;          // srk_ba_set_multi_schedule(h, 0): the round-2 schedule (all-reduce, redundant solves)
    bool dp_force = false;            // srk_ba_set_multi_schedule(h, 2): the schedule at world size 1 as well (three slots, every
                                      // collective issued; all one GPU can rehearse of the native path)
    // The native form of that schedule (groups of ncclReduce / ncclBroadcast rooted at different ranks on one communicator)
    // has never run on more than one GPU in this repository's tests: its FIRST round on a handle checks itself against plain
    // all-reduces of checksums (dp_selfcheck below); a mismatch or an RCCL error switches the handle to the all-reduce schedule.
    bool dp_verified = false;         // the first native round passed its self-check
    bool dp_selfcheck_failed = false; // ... or did not: the handle runs the all-reduce schedule (srk_ba_multi_schedule: 3)
    DevBuf dp_chk;                    // scratch of the self-check: [512] partial sums, [2] checksum, [16] all-reduce staging
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_comm = nullptr;
    DevBuf status_all;                // [SRK_SLOTS][8] doubles: the slots' {error, solver info, point-update info}
    double* dp_back = nullptr;        // pinned copy of status_all
    int64_t seen_global = -1; // observation count over all ranks of the uploaded scene (-1 = not yet exchanged)

    // timing
    hipEvent_t ev[16]{};
    std::vector<hipEvent_t> chol_ev;
    bool schur_fp32 = false; // opt-in mixed precision: fp32 run sums in the grouped Schur kernel
    bool store_f32 = false;  // opt-in: the point-frame blocks W are STORED as float (next upload); arithmetic stays fp64
    int profile_level = 0; // 0 = no events, 1 = phase events (report.ms_*), 2 = + event pairs around the MFMA updates
    bool chol_fused = true; // the solve's outer steps as one launch each (k_step256); srk_ba_set_solver_fusion
    // what the caller asked for.  A hand-off timeout switches chol_fused off for the rest of that call; the next upload /
    // optimise call switches it back on (a timeout is a scheduling event: another process on the GPU, a debugger) until the
    // handle has seen SRK_FUSION_RETRIES of them -- then the unfused sequence stays, and srk_ba_solver_sync_timeouts says so.
    bool chol_fused_wanted = true;
    int fusion_rearms_left = 3; // SRK_FUSION_RETRIES
    int64_t sync_timeouts = 0; // solves repeated with the unfused kernels after a hand-off timed out
    struct IterLog { int32_t attempts; double ms, err, factor; };
    std::vector<IterLog> iter_log; // the accepted iterations of the last optimise call (srk_ba_iteration_log)
    int last_slot = 0;     // attempt slot of the last judged attempt (what SRK_BUF_RCS / RHS / CORRECTIONS download)
    double last_hessian_factor = 0;
    bool lean_resets = false; // srk_ba_optimize: no per-attempt memsets (see phase_solve)
    // A failed factorisation (non-positive / non-finite pivot) leaves NaNs in the slot's system and chunk matrices, also
    // OUTSIDE the parts the next attempt rewrites (k_panel's dead rows: 0 * NaN).  The flag makes the next entry point
    // re-zero every slot's system and plan buffers (clear_poison) before anything is computed from them.
    bool poisoned = false;
};

// librccl.so is opened on first use: the library itself carries no link-time dependency on it (a single-GPU caller
// never needs it), and a Python caller keeps torch's own copy to itself
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    bool rooted = false; // Reduce / Broadcast / GroupStart / GroupEnd resolved (else they are emulated by all-reduces)
};
RcclApi& rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names)
            if ((a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!a.lib) return a;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        a.Reduce = reinterpret_cast<decltype(a.Reduce)>(dlsym(a.lib, "ncclReduce"));
        a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(a.lib, "ncclBroadcast"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(a.lib, "ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(a.lib, "ncclGroupEnd"));
        a.rooted = a.Reduce && a.Broadcast && a.GroupStart && a.GroupEnd;
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
        return a;
    }();
    return api;
}
} // namespace

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            (h)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                     \
            return SRK_E_DEVICE;                                                                     \
        }                                                                                            \
    } while (0)

// SRK_DEBUG=1: plan and per-attempt traces on stderr
static bool srk_debug()
{
    static const bool on = getenv("SRK_DEBUG") != nullptr;
    return on;
}

static int dev_alloc(srk_ba* h, DevBuf& b, size_t bytes)
{
    if (bytes == 0) bytes = 8;
    if (b.p && b.bytes >= bytes) return SRK_OK;
    if (b.p) {
        hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        h->last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
        b.p = nullptr;
        return e == hipErrorOutOfMemory ? SRK_E_NOMEM : SRK_E_DEVICE;
    }
    b.bytes = bytes;
    return SRK_OK;
}
static void dev_free(DevBuf& b)
{
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
template <typename T> static T* P(const DevBuf& b) { return reinterpret_cast<T*>(b.p); }

extern "C" {

int srk_ba_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

srk_ba* srk_ba_create(int device_id)
{
    int n = srk_ba_device_count();
    if (n <= 0 || device_id < 0 || device_id >= n) {
        fprintf(stderr, "srk_ba_create: no usable HIP device (count=%d, requested=%d); there is no CPU fallback\n", n,
                device_id);
        return nullptr;
    }
    if (hipSetDevice(device_id) != hipSuccess) return nullptr;
    srk_ba* h = new srk_ba();
    h->device = device_id;
    if (hipStreamCreate(&h->stream) != hipSuccess) {
        delete h;
        return nullptr;
    }
    h->main_stream = h->stream;
#ifdef SRK_DEV // development switches (tools/): the default build reads no environment but SRK_DEBUG (the trace)
    if (const char* e = getenv("SRK_CHOL_FUSED")) h->chol_fused = h->chol_fused_wanted = e[0] != '0'; // the unfused launch sequence
    if (const char* e = getenv("SRK_MULTI_SPECULATION")) h->spec_multi = e[0] != '0';
    if (const char* e = getenv("SRK_MULTI_SCHEDULE")) {
        h->dp_schedule = std::strcmp(e, "allreduce") != 0;
        h->dp_force = std::strcmp(e, "dp_force") == 0;
    }
#endif
    h->att[0].stream = h->stream;
    bool ok = hipEventCreateWithFlags(&h->ev_jac, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&h->dp_back), 64 * SRK_SLOTS, hipHostMallocDefault) == hipSuccess &&
              hipMalloc(&h->status_all.p, 64 * SRK_SLOTS) == hipSuccess;
    if (ok) h->status_all.bytes = 64 * SRK_SLOTS, ok = hipMemset(h->status_all.p, 0, 64 * SRK_SLOTS) == hipSuccess;
    for (int sl = 0; sl < SRK_SLOTS; ++sl) {
        auto& a = h->att[sl];
        a.trial = sl + 1;
        a.err_dst = ok ? P<double>(h->status_all) + 8 * sl : nullptr;
        if (sl > 0) ok = ok && hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&a.done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&a.ev_a, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&a.ev_b, hipEventDisableTiming) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void**>(&a.host_back), 64, hipHostMallocDefault) == hipSuccess;
    }
    for (auto& e : h->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    if (!ok) {
        delete h;
        return nullptr;
    }
    return h;
}

// the phase functions work on h->A and enqueue on h->stream: point both at one attempt slot
static void select_attempt(srk_ba* h, int slot)
{
    h->A = &h->att[slot];
    h->stream = h->A->stream;
}

void srk_ba_destroy(srk_ba* h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (int sl = 1; sl < SRK_SLOTS; ++sl)
        if (h->att[sl].stream) hipStreamSynchronize(h->att[sl].stream);
    if (h->comm_stream) hipStreamSynchronize(h->comm_stream);
    if (h->comm2) rccl().CommDestroy(h->comm2); // the second slot's communicator is always owned by the handle
    if (h->comm && h->comm_owned) rccl().CommDestroy(h->comm);
    h->comm = h->comm2 = nullptr;
    for (int w = 0; w < SRK_SLOTS + 1; ++w)
        for (DevBuf* b : { &h->pts[w], &h->camR[w], &h->camT[w], &h->cam[w] }) dev_free(*b);
    dev_free(h->status_all);
    if (h->dp_back) hipHostFree(h->dp_back);
    DevBuf* all[] = { &h->K, &h->pts0, &h->camR0, &h->camT0, &h->row_ptr,
                      &h->obs_frame, &h->obs_pt, &h->obs_uv, &h->col_ptr, &h->fobs_pt, &h->fobs_uv, &h->W, &h->Vg, &h->Ug,
                      &h->scratch, &h->grp_first, &h->grp_count, &h->grp_nf, &h->grp_frames, &h->obs_slot, &h->pt_mask,
                      &h->gen_list, &h->env_col, &h->env_off, &h->wg_jmin, &h->band_col, &h->band_off,
                      &h->dj_ptr, &h->dj_ent, &h->dj_stage, &h->ds_pair_ptr, &h->ds_pair_fa, &h->ds_pair_fb, &h->ds_pair_ent, &h->ds_f_ptr, &h->ds_f_ent,
                      &h->jd_nf, &h->jd_frames, &h->jd_mask,
                      &h->jr_first, &h->jr_count, &h->jr_jmin, &h->jr_group, &h->lg_item, &h->lg_np, &h->lg_nf, &h->lg_pts, &h->lg_frames,
                      &h->lg_obs_off, &h->lg_obs };
    for (DevBuf* b : all) dev_free(*b);
    for (auto& a : h->att) {
        for (DevBuf* b : { &a.S, &a.rhs, &a.wy, &a.dc, &a.acc, &a.dx, &a.err_partial, &a.info, &a.dinv, &a.packed, &a.sync_flags, &a.irr, &a.det_stage, &a.det_rhs }) dev_free(*b);
        for (DevBuf& b : a.plan_bufs) dev_free(b);
        if (a.host_back) hipHostFree(a.host_back);
        if (a.done) hipEventDestroy(a.done);
        if (a.ev_a) hipEventDestroy(a.ev_a);
        if (a.ev_b) hipEventDestroy(a.ev_b);
    }
    for (DevBuf* b : { &h->sc_pts, &h->sc_R, &h->sc_T, &h->sc_K, &h->sc_cam, &h->sc_frame, &h->sc_pt, &h->sc_uv, &h->sc_partial, &h->sc_out })
        dev_free(*b);
    for (auto& e : h->ev)
        if (e) hipEventDestroy(e);
    for (auto& e : h->chol_ev) hipEventDestroy(e);
    if (h->ev_jac) hipEventDestroy(h->ev_jac);
    if (h->ev_comm) hipEventDestroy(h->ev_comm);
    if (h->comm_stream) hipStreamDestroy(h->comm_stream);
    for (int sl = 1; sl < SRK_SLOTS; ++sl)
        if (h->att[sl].stream) hipStreamDestroy(h->att[sl].stream);
    if (h->own_stream && h->main_stream) hipStreamDestroy(h->main_stream);
    delete h;
}

const char* srk_ba_last_error(const srk_ba* h) { return h ? h->last_error.c_str() : "null handle"; }

const char* srk_ba_status_string(int status)
{
    switch (status) {
    case SRK_STATUS_ABS_ERR_THRESHOLD: return "abs err threshold";
    case SRK_STATUS_SMALL_ERR_CHANGE: return "small relative err change";
    case SRK_STATUS_HESSIAN_OVERFLOW: return "hessian overflow";
    case SRK_STATUS_ERR_CONVERGED: return "err converged to limit value";
    case SRK_STATUS_MAX_ITERATIONS: return "max iterations";
    case SRK_STATUS_DEVICE_ERROR: return "device error";
    default: return "";
    }
}

int srk_ba_set_stream(srk_ba* h, void* hip_stream)
{
    if (!h) return SRK_E_ARGS;
    hipSetDevice(h->device);
    select_attempt(h, 0);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    h->stream = h->main_stream = h->att[0].stream = reinterpret_cast<hipStream_t>(hip_stream);
    h->own_stream = false;
    return SRK_OK;
}

// landmark shards must agree on the frame numbering: each rank's own renumbering (made from its shard at upload) would not
static bool exchange_after_reordered_upload(srk_ba* h, int world_size)
{
    if (world_size < 2 || !h->have_scene || h->frame_int.empty() || h->frame_order_supplied) return false;
    h->last_error = "the uploaded scene's frames were renumbered for one rank; configure the exchange before the upload";
    return true;
}

int srk_ba_set_allreduce(srk_ba* h, srk_allreduce_fn fn, void* ctx, int rank, int world_size)
{
    if (!h || world_size < 1 || rank < 0 || rank >= world_size) return SRK_E_ARGS;
    if (exchange_after_reordered_upload(h, world_size)) return SRK_E_STATE;
    // either exchange replaces the other: exchange() prefers a communicator, so a callback set after srk_ba_rccl_init
    // would never be called unless the communicators are detached here
    if (h->comm || h->comm2) {
        hipSetDevice(h->device);
        if (h->stream) hipStreamSynchronize(h->stream);
        for (int sl = 1; sl < SRK_SLOTS; ++sl)
            if (h->att[sl].stream) hipStreamSynchronize(h->att[sl].stream);
        if (h->comm_stream) hipStreamSynchronize(h->comm_stream);
        if (h->comm2) rccl().CommDestroy(h->comm2);
        if (h->comm && h->comm_owned) rccl().CommDestroy(h->comm);
        h->comm = h->comm2 = nullptr;
        h->comm_owned = false;
    }
    h->allreduce = fn;
    h->allreduce_ctx = ctx;
    h->rank = rank;
    h->world = world_size;
    h->seen_global = -1;
    return SRK_OK;
}

// ---- native RCCL exchange
int srk_ba_rccl_get_unique_id(void* id128)
{
    if (!id128 || !rccl().ok) return SRK_E_DEVICE;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
    return rccl().GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)) == ncclSuccess ? SRK_OK : SRK_E_DEVICE;
}
static int rccl_attach(srk_ba* h, ncclComm_t comm, bool owned, int rank, int world_size)
{
    if (exchange_after_reordered_upload(h, world_size)) {
        if (owned) rccl().CommDestroy(comm);
        return SRK_E_STATE;
    }
    if (h->comm && h->comm_owned) rccl().CommDestroy(h->comm);
    if (h->comm2) rccl().CommDestroy(h->comm2);
    h->comm2 = nullptr;
    h->comm = comm;
    h->comm_owned = owned;
    h->allreduce = nullptr;
    h->allreduce_ctx = nullptr;
    h->rank = rank;
    h->world = world_size;
    h->seen_global = -1;
    return SRK_OK;
}
int srk_ba_rccl_init(srk_ba* h, const void* id128, int rank, int world_size)
{
    if (!h || !id128 || world_size < 1 || rank < 0 || rank >= world_size) return SRK_E_ARGS;
    if (!rccl().ok) { h->last_error = "librccl.so could not be loaded"; return SRK_E_DEVICE; }
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    ncclResult_t r = rccl().CommInitRank(&comm, world_size, id, rank);
    if (r != ncclSuccess) { h->last_error = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r); return SRK_E_DEVICE; }
    return rccl_attach(h, comm, true, rank, world_size);
}
// collective, after srk_ba_rccl_init: a second communicator (its own ncclUniqueId) for the second attempt slot
int srk_ba_rccl_init_second(srk_ba* h, const void* id128)
{
    if (!h || !id128) return SRK_E_ARGS;
    if (!h->comm || !rccl().ok) { h->last_error = "srk_ba_rccl_init_second: srk_ba_rccl_init comes first"; return SRK_E_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    ncclResult_t r = rccl().CommInitRank(&comm, h->world, id, h->rank);
    if (r != ncclSuccess) { h->last_error = std::string("ncclCommInitRank (second): ") + rccl().GetErrorString(r); return SRK_E_DEVICE; }
    if (h->comm2) rccl().CommDestroy(h->comm2);
    h->comm2 = comm;
    return SRK_OK;
}
int srk_ba_rccl_set_comm(srk_ba* h, void* nccl_comm, int rank, int world_size)
{
    if (!h || world_size < 1 || rank < 0 || rank >= world_size) return SRK_E_ARGS;
    if (nccl_comm && !rccl().ok) { h->last_error = "librccl.so could not be loaded"; return SRK_E_DEVICE; }
    return rccl_attach(h, reinterpret_cast<ncclComm_t>(nccl_comm), false, rank, world_size);
}

// ------------------------------------------------------------------ host normalisation (bundle-adj-kanatani.cpp:123-333)

int srk_ba_check_world_is_normalized(int32_t M, const double* cam_R, const double* cam_T, double t1y, int32_t comp)
{
    if (M < 2 || !cam_R || !cam_T || comp < 0 || comp > 2) return 0; // :291-292
    const double atol = 1e-3;                                           // :297
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            if (!srk::is_close(r == c ? 1.0 : 0.0, cam_R[3 * r + c], atol, atol)) return 0; // :299 IsIdentity
    if (srk::norm3(cam_T) >= atol) return 0;                                                  // :308-309
    double Ri[9], Ti[3];
    srk::se3_inv(cam_R + 9, cam_T + 3, Ri, Ti);                                               // :320
    return srk::is_close(t1y, std::fabs(Ti[comp]), atol) ? 1 : 0;                             // :323
}

int srk_ba_normalize_scene(int64_t N, double* pts, int32_t M, double* cam_R, double* cam_T, double t1y, int32_t comp,
                           srk_ba_normalizer* out)
{
    if (M < 2 || comp < 0 || comp > 2 || !out || !cam_R || !cam_T || (N > 0 && !pts)) return 0;
    // cam0_from1 = SE3AFromB(cam0, cam1)  (:208, obs-geom.cpp:141-150)
    double R1i[9], T1i[3], v[3], T01[3];
    srk::se3_inv(cam_R + 9, cam_T + 3, R1i, T1i);
    srk::mat3_vec(cam_R, T1i, v);
    for (int i = 0; i < 3; ++i) T01[i] = v[i] + cam_T[i];
    double shift = T01[comp];
    if (srk::is_close(0.0, shift, 1e-5)) return 0; // :215-217 (the lone third argument is rtol)
    double s = t1y / std::fabs(shift);             // :219
    std::memcpy(out->R0, cam_R, sizeof out->R0);
    std::memcpy(out->T0, cam_T, sizeof out->T0);
    out->world_scale = s;
    double R0t[9];
    srk::mat3_tr(out->R0, R0t);
    for (int32_t j = 0; j < M; ++j) { // NormalizeRT :143-162
        double RR[9], u[3];
        srk::mat3_mul(cam_R + 9 * (int64_t)j, R0t, RR);
        srk::mat3_vec(RR, out->T0, u);
        for (int i = 0; i < 3; ++i) cam_T[3 * (int64_t)j + i] = (cam_T[3 * (int64_t)j + i] - u[i]) * s;
        std::memcpy(cam_R + 9 * (int64_t)j, RR, sizeof RR);
    }
    for (int64_t i = 0; i < N; ++i) { // :179-199
        double y[3];
        srk::se3_apply(out->R0, out->T0, pts + 3 * i, y);
        pts[3 * i] = y[0] * s;
        pts[3 * i + 1] = y[1] * s;
        pts[3 * i + 2] = y[2] * s;
    }
    return 1;
}

void srk_ba_revert_normalization(int64_t N, double* pts, int32_t M, double* cam_R, double* cam_T,
                                 const srk_ba_normalizer* nrm)
{
    double s = nrm->world_scale;
    double R0t[9];
    srk::mat3_tr(nrm->R0, R0t);
    for (int64_t i = 0; i < N; ++i) { // :187-191
        double is = 1 / s;
        double t[3] = { pts[3 * i] * is - nrm->T0[0], pts[3 * i + 1] * is - nrm->T0[1], pts[3 * i + 2] * is - nrm->T0[2] };
        srk::mat3_vec(R0t, t, pts + 3 * i);
    }
    for (int32_t j = 0; j < M; ++j) { // RevertRT :164-177
        double RR[9], u[3];
        double* R = cam_R + 9 * (int64_t)j;
        double* T = cam_T + 3 * (int64_t)j;
        srk::mat3_mul(R, nrm->R0, RR);
        srk::mat3_vec(R, nrm->T0, u);
        for (int i = 0; i < 3; ++i) T[i] = T[i] / s + u[i];
        std::memcpy(R, RR, sizeof RR);
    }
}

} // extern "C"

// ------------------------------------------------------------------ scene upload

static int validate_scene(srk_ba* h, double f0, int64_t N, const double* pts, int32_t M, const double* cam_R,
                          const double* cam_T, const double* K, const int64_t* row_ptr, const int32_t* obs_frame,
                          const double* obs_uv, int32_t min_frames = 2)
{
    if (!h) return SRK_E_ARGS;
    // CHECK(!IsClose(0, f0)) (:420)
    if (srk::is_close(0.0, f0)) { h->last_error = "f0 must not be ~0"; return SRK_E_ARGS; }
    if (M < min_frames) { h->last_error = min_frames > 1 ? "need at least two frames" : "need at least one frame"; return SRK_E_ARGS; }
    if (N < 0 || !cam_R || !cam_T || !K || !row_ptr) { h->last_error = "null scene array"; return SRK_E_ARGS; }
    if (N > 0 && !pts) { h->last_error = "null points"; return SRK_E_ARGS; }
    if (N > 2147483000LL) { h->last_error = "too many points for int32 indices"; return SRK_E_ARGS; }
    if (row_ptr[0] != 0) { h->last_error = "obs_row_ptr[0] != 0"; return SRK_E_ARGS; }
    int64_t O = row_ptr[N];
    if (O > 0 && (!obs_frame || !obs_uv)) { h->last_error = "null observation arrays"; return SRK_E_ARGS; }
    for (int64_t i = 0; i < N; ++i) {
        if (row_ptr[i + 1] < row_ptr[i]) { h->last_error = "obs_row_ptr not monotone"; return SRK_E_ARGS; }
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o) {
            int32_t j = obs_frame[o];
            if (j < 0 || j >= M) { h->last_error = "obs_frame out of range"; return SRK_E_ARGS; }
            if (o > row_ptr[i] && obs_frame[o - 1] >= j) {
                h->last_error = "obs_frame must be strictly ascending inside a point";
                return SRK_E_ARGS;
            }
        }
    }
    return SRK_OK;
}

static int compute_cam_packs(srk_ba* h, int which)
{
    srk_launch_cam_pack(h->stream, h->d.M, P<double>(h->camR[which]), P<double>(h->camT[which]), P<double>(h->K),
                        h->f0, P<double>(h->cam[which]));
    HIPCHK(h, hipGetLastError());
    return SRK_OK;
}

// Chunked solve plan: P chunks of the variable range separated by P-1 separators at least as wide as the
// bandwidth (so chunks are decoupled).  The separator system is block tridiagonal in units of sepw, so it is chunked
// again (cuts aligned to those blocks) by a child plan: nested dissection of a banded system, every level one batch.
//
// Cost model, in units of one 256-column outer step of the blocked Cholesky (~0.17 ms on MI355X): a chunked level
// pays its longest chunk (border rows make a step ~10% dearer), a fixed overhead for gather / reduce / scatter,
// and the cost of its separator system.
static double plan_cost(int64_t ld, int64_t sepw, int64_t unit, int* best_P)
{
    const double n256 = (double)(ld / SRK_CHOL_NB);
    double best = n256;
    int bp = 0;
    for (int p = 2; p <= SRK_MAX_CHUNKS; ++p) {
        const int64_t interior = ld - sepw * (p - 1);
        if (interior < (int64_t)p * sepw) break; // every chunk at least one separator wide
        const int64_t units = interior / unit;
        const int64_t longest = ((units + p - 1) / p) * unit;
        const double cst = 1.1 * (double)((longest + SRK_CHOL_NB - 1) / SRK_CHOL_NB) + 0.6 + plan_cost(sepw * (p - 1), sepw, sepw, nullptr);
        if (cst < best - 1e-9) best = cst, bp = p;
    }
    if (best_P) *best_P = bp;
    return best;
}

// row_end (per 256 block, multiple of 128) / col_begin (per 64 tile): the skyline of the system to be chunked
static int make_plan(srk_ba* h, SrkChunkPlan& pl, int64_t ld, int64_t sepw, int64_t unit,
                     const std::vector<int64_t>& row_end, const std::vector<int64_t>& col_begin)
{
    pl.P = 0;
    int P = 0;
    plan_cost(ld, sepw, unit, &P);
    if (P < 2) return SRK_OK;
    const int64_t interior = ld - sepw * (P - 1);
    const int64_t blocks = interior / unit; // ld, sepw are multiples of unit
    if (srk_debug())
        fprintf(stderr, "srk_ba chunk plan: system %lld -> %d chunks of <= %lld + %d separators of %lld\n", (long long)ld, P,
                (long long)(((blocks + P - 1) / P) * unit), P - 1, (long long)sepw);
    pl.sepw = sepw;
    pl.lds = sepw * (P - 1);
    std::vector<int64_t> sep_start((size_t)(P - 1));
    int64_t pos = 0;
    auto alloc = [&](size_t bytes, bool zero) -> void* {
        h->A->plan_bufs.emplace_back();
        h->A->plan_zeroed.push_back(zero ? 1 : 0);
        if (dev_alloc(h, h->A->plan_bufs.back(), bytes) != SRK_OK) return nullptr;
        if (zero) hipMemsetAsync(h->A->plan_bufs.back().p, 0, bytes, h->stream);
        return h->A->plan_bufs.back().p;
    };
    for (int c = 0; c < P; ++c) {
        int64_t nb = blocks / P + (c < blocks % P ? 1 : 0);
        pl.a[c] = pos;
        pl.n[c] = nb * unit;
        pl.ldc[c] = pl.n[c] + 2 * sepw;
        pos += pl.n[c];
        if (c < P - 1) {
            sep_start[(size_t)c] = pos;
            pos += sepw;
        }
        const int64_t nc = pl.n[c], ldc = pl.ldc[c];
        pl.Ac[c] = (double*)alloc((size_t)(8 * ldc * ldc), true);
        pl.wc[c] = (double*)alloc((size_t)(8 * ldc), true);
        pl.yc[c] = (double*)alloc((size_t)(8 * ldc), true);
        pl.xc[c] = (double*)alloc((size_t)(8 * ldc), true);
        pl.dinvc[c] = (double*)alloc((size_t)(8 * 64 * ldc), false);
        if (!pl.Ac[c] || !pl.wc[c] || !pl.yc[c] || !pl.xc[c] || !pl.dinvc[c]) return SRK_E_NOMEM;
        pl.row_end[c].assign((size_t)(nc / SRK_CHOL_NB), 0);
        for (int64_t K = 0; K < nc / SRK_CHOL_NB; ++K) {
            int64_t rg = row_end[(size_t)(pl.a[c] / SRK_CHOL_NB + K)] - pl.a[c];
            pl.row_end[c][(size_t)K] = std::min<int64_t>(std::max<int64_t>(rg, SRK_CHOL_NB * (K + 1)), nc);
        }
        pl.col_begin[c].assign((size_t)(nc / 64), 0);
        for (int64_t q = 0; q < nc / 64; ++q)
            pl.col_begin[c][(size_t)q] = std::max<int64_t>(col_begin[(size_t)(pl.a[c] / 64 + q)] - pl.a[c], 0);
    }
    const int64_t lds = pl.lds;
    pl.Cs = (double*)alloc((size_t)(8 * lds * lds), true);
    pl.ws = (double*)alloc((size_t)(8 * lds), true);
    pl.ys = (double*)alloc((size_t)(8 * lds), true);
    pl.xs = (double*)alloc((size_t)(8 * lds), true);
    pl.dinvs = (double*)alloc((size_t)(8 * 64 * lds), false);
    pl.d_sep_start = (int64_t*)alloc((size_t)(8 * (P - 1)), false);
    pl.d_sep_env = (int64_t*)alloc((size_t)(8 * (lds / 128)), false);
    if (!pl.Cs || !pl.ws || !pl.ys || !pl.xs || !pl.dinvs || !pl.d_sep_start || !pl.d_sep_env) return SRK_E_NOMEM;
    HIPCHK(h, hipMemcpyAsync(pl.d_sep_start, sep_start.data(), (size_t)(8 * (P - 1)), hipMemcpyHostToDevice, h->stream));
    // separators only couple with their neighbours (through the chunk between them): block tridiagonal skyline
    pl.s_row_end.assign((size_t)(lds / SRK_CHOL_NB), 0);
    for (int64_t K = 0; K < lds / SRK_CHOL_NB; ++K) {
        int64_t c = (SRK_CHOL_NB * K) / sepw;
        pl.s_row_end[(size_t)K] = std::min<int64_t>(lds, (c + 2) * sepw);
    }
    pl.s_col_begin.assign((size_t)(lds / 64), 0);
    for (int64_t q = 0; q < lds / 64; ++q) {
        int64_t c = (64 * q) / sepw;
        pl.s_col_begin[(size_t)q] = std::max<int64_t>(c - 1, 0) * sepw;
    }
    std::vector<int64_t> sep_env((size_t)(lds / 128));
    for (int64_t t = 0; t < lds / 128; ++t) sep_env[(size_t)t] = pl.s_col_begin[(size_t)(2 * t)];
    HIPCHK(h, hipMemcpyAsync(pl.d_sep_env, sep_env.data(), (size_t)(8 * (lds / 128)), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // the separator system, chunked again when that shortens its chain
    h->A->plan_children.emplace_back(new SrkChunkPlan());
    SrkChunkPlan* child = h->A->plan_children.back().get();
    int rc = make_plan(h, *child, lds, sepw, sepw, pl.s_row_end, pl.s_col_begin);
    if (rc != SRK_OK) return rc;
    pl.child = child->P >= 2 ? child : nullptr;
    pl.P = P;
    return SRK_OK;
}

static int build_chunk_plan(srk_ba* h)
{
    const SrkDims& d = h->d;
    SrkChunkPlan& pl = h->A->plan;
    // the same skyline as the plan in place was built for (a scene uploaded again, the next call of a caller that adjusts the
    // same tracks): keep plan and buffers -- freeing and allocating them again costs ~9 ms a slot at 1000 frames -- and
    // bring the buffers that must be zero outside what a solve writes back to zero
    std::vector<int64_t> sig = { d.ld, h->use_envelope ? 1 : 0, h->use_chunks ? 1 : 0 };
    sig.insert(sig.end(), h->min_cv.begin(), h->min_cv.end());
    if (!h->A->plan_sig.empty() && sig == h->A->plan_sig) {
        for (size_t i = 0; i < h->A->plan_bufs.size(); ++i)
            if (h->A->plan_zeroed[i]) HIPCHK(h, hipMemsetAsync(h->A->plan_bufs[i].p, 0, h->A->plan_bufs[i].bytes, h->stream));
        return SRK_OK;
    }
    h->A->plan_sig = sig;
    pl.P = 0;
    pl.child = nullptr;
    for (DevBuf& b : h->A->plan_bufs) dev_free(b);
    h->A->plan_bufs.clear();
    h->A->plan_zeroed.clear();
    h->A->plan_children.clear();
    if (!h->use_envelope || !h->use_chunks) return SRK_OK;
    int64_t maxdist = 0;
    for (int32_t j = 0; j < d.M; ++j) maxdist = std::max<int64_t>(maxdist, 10 * (int64_t)(j - h->min_cv[(size_t)j]) + 9);
    // separators at least one bandwidth wide, in units of the 256-column outer panel (k_bwd_border stages 2 sepw values)
    const int64_t sepw = (maxdist + SRK_CHOL_NB - 1) / SRK_CHOL_NB * SRK_CHOL_NB;
    if (sepw > SRK_MAX_SEPW) return SRK_OK;
    const int rc = make_plan(h, pl, d.ld, sepw, SRK_CHOL_NB, h->row_end_h, h->col_begin_h);
    if (rc != SRK_OK) h->A->plan_sig.clear();
    return rc;
}

// Skyline of the RCS from the covisibility (min_cv[j] = smallest frame index sharing a landmark with frame j).
// All quantities are aligned to the solver's blocking: env_col multiples of 256, row_end multiples of 128.
static int build_envelope(srk_ba* h)
{
    const SrkDims& d = h->d;
    const int64_t nt = d.ld / 128, nk = d.ld / SRK_CHOL_NB, n64 = d.ld / 64;
    h->env_col_h.assign((size_t)nt, 0);
    for (int64_t t = 0; t < nt; ++t) {
        int64_t r0 = 128 * t, r1 = 128 * t + 127;
        int64_t fc = r0; // padding rows and the diagonal itself
        if (h->use_envelope) {
            for (int64_t j = r0 / 10; j <= r1 / 10 && j < d.M; ++j) fc = std::min<int64_t>(fc, 10 * (int64_t)h->min_cv[(size_t)j]);
        } else {
            fc = 0;
        }
        h->env_col_h[(size_t)t] = (fc / SRK_CHOL_NB) * SRK_CHOL_NB;
    }
    h->row_end_h.assign((size_t)nk, 0);
    for (int64_t K = 0; K < nk; ++K) {
        int64_t last = -1;
        for (int64_t t = nt - 1; t >= 0; --t)
            if (h->env_col_h[(size_t)t] <= SRK_CHOL_NB * K) { last = t; break; }
        int64_t re = 128 * (last + 1);
        re = std::max<int64_t>(re, SRK_CHOL_NB * (K + 1));
        h->row_end_h[(size_t)K] = std::min<int64_t>(re, d.ld);
    }
    h->col_begin_h.assign((size_t)n64, 0);
    for (int64_t q = 0; q < n64; ++q) h->col_begin_h[(size_t)q] = h->env_col_h[(size_t)(q / 2)];
    h->env_off_h.assign((size_t)nt + 1, 0);
    for (int64_t t = 0; t < nt; ++t)
        h->env_off_h[(size_t)t + 1] = h->env_off_h[(size_t)t] + 128 * (128 * (t + 1) - h->env_col_h[(size_t)t]);
    h->env_packed = h->env_off_h[(size_t)nt];
    int rc;
    if ((rc = dev_alloc(h, h->env_col, (size_t)(8 * nt))) != SRK_OK) return rc;
    if ((rc = dev_alloc(h, h->env_off, (size_t)(8 * (nt + 1)))) != SRK_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(h->env_col.p, h->env_col_h.data(), (size_t)(8 * nt), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->env_off.p, h->env_off_h.data(), (size_t)(8 * (nt + 1)), hipMemcpyHostToDevice, h->stream));
    // exchange format of the assembled system (landmark shards): the exact pre-factorisation band of every row
    {
        std::vector<int64_t> bc((size_t)d.ld), bo((size_t)d.ld + 1, 0);
        for (int64_t r = 0; r < d.ld; ++r) {
            int64_t c0 = r; // padding rows: the diagonal only
            if (r < 10 * (int64_t)d.M) c0 = h->use_envelope ? 10 * (int64_t)h->min_cv[(size_t)(r / 10)] : 0;
            bc[(size_t)r] = c0;
            bo[(size_t)r + 1] = bo[(size_t)r] + (r - c0 + 1);
        }
        h->band_packed = bo[(size_t)d.ld];
        if ((rc = dev_alloc(h, h->band_col, (size_t)(8 * d.ld))) != SRK_OK) return rc;
        if ((rc = dev_alloc(h, h->band_off, (size_t)(8 * (d.ld + 1)))) != SRK_OK) return rc;
        HIPCHK(h, hipMemcpyAsync(h->band_col.p, bc.data(), (size_t)(8 * d.ld), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->band_off.p, bo.data(), (size_t)(8 * (d.ld + 1)), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream)); // bc / bo are locals
    }
    for (int sl = 0; sl < SRK_SLOTS; ++sl) { // every allocated attempt slot: its own zeroed system and solver plan
        select_attempt(h, 0);
        if (!h->att[sl].allocated) continue;
        HIPCHK(h, hipMemsetAsync(h->att[sl].S.p, 0, (size_t)(8 * d.ld * d.ld), h->stream)); // outside the skyline stays 0
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->A = &h->att[sl]; // plan buffers of slot sl, allocated and zeroed on the main stream
        rc = build_chunk_plan(h);
        h->A = &h->att[0];
        if (rc != SRK_OK) return rc;
    }
    return SRK_OK;
}

// Internal frame order.  The reference treats the reduced camera system as a dense matrix (bundle-adj-kanatani.cpp:1911), so
// the order of the frames means nothing to it.  Here everything fast depends on covisible frames having NEARBY indices: the
// skyline of the system, its nested dissection (separators one bandwidth wide), the frame windows of the derivative kernels.
// An image sequence in time order has that property; the same frames in any other order (an unordered image set), or a
// sequence that closes a loop (the last frames see the first frames' landmarks), do not -- the solve alone then takes 5x
// longer on the 1000-frame scene (one skyline chain instead of chunks).  So, when the caller's order is far from banded,
// the frames are renumbered by reverse Cuthill-McKee on the covisibility graph (two frames adjacent iff they share a
// landmark), started from a pseudo-peripheral frame: a shuffled sequence gets its band back, a closed loop becomes a band
// of two to three times the width (twice is the optimum for a ring).  Only the numbering changes -- arithmetic per block, gauge (the caller's frames 0 and 1, wherever they
// land: SrkDims::g0, g1) and results are those of the caller's order; every download maps back.
// Returns true and fills to_int[caller's frame] = internal index when renumbering pays.  mode: see srk_ba::frame_order_mode.
static bool frame_reorder(int mode, int64_t N, int32_t M, const int64_t* row_ptr, const int32_t* obs_frame, std::vector<int32_t>& to_int)
{
    if (mode == 0 || M < 3 || M > 16384) return false;
    int64_t bw_nat = 0, lmax = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t k = row_ptr[i + 1] - row_ptr[i];
        if (k < 1) continue;
        lmax = std::max(lmax, k);
        bw_nat = std::max<int64_t>(bw_nat, obs_frame[row_ptr[i + 1] - 1] - obs_frame[row_ptr[i]]); // lists ascend
    }
    if (mode < 0 && bw_nat <= 2 * lmax) return false; // as banded as tracks of that length allow
    // covisibility graph as a bit matrix (M <= 16384: 32 MB); every distinct frame list once
    const size_t wpr = ((size_t)M + 63) / 64;
    std::vector<uint64_t> adj((size_t)M * wpr, 0);
    std::unordered_set<uint64_t> seen;
    int64_t pair_work = 0;
    const int64_t pair_budget = 400000000; // ~1 s of host time
    for (int64_t i = 0; i < N; ++i) {
        const int64_t k = row_ptr[i + 1] - row_ptr[i];
        if (k < 2) continue;
        const int32_t* f = obs_frame + row_ptr[i];
        uint64_t hsh = 1469598103934665603ull ^ (uint64_t)k;
        for (int64_t a = 0; a < k; ++a) hsh = (hsh ^ (uint64_t)(uint32_t)f[a]) * 1099511628211ull;
        if (!seen.insert(hsh).second) continue; // (a collision only costs ordering quality: the skyline is built from the observations)
        auto link = [&](int32_t u, int32_t v) {
            adj[(size_t)u * wpr + (size_t)(v >> 6)] |= 1ull << (v & 63);
            adj[(size_t)v * wpr + (size_t)(u >> 6)] |= 1ull << (u & 63);
        };
        // all pairs of a list while the work stays bounded (long ragged tracks in an unordered set: ~1e6 distinct lists of
        // ~500 frames would be 1e11 insertions before the first kernel); beyond the budget a list's chain of consecutive
        // frames plus its first-last pair, which keeps the graph connected along every track
        if (pair_work + k * (k - 1) / 2 <= pair_budget) {
            pair_work += k * (k - 1) / 2;
            for (int64_t a = 0; a < k; ++a)
                for (int64_t b = a + 1; b < k; ++b) link(f[a], f[b]);
        } else {
            for (int64_t a = 0; a + 1 < k; ++a) link(f[a], f[a + 1]);
            link(f[0], f[k - 1]);
        }
    }
    std::vector<int32_t> deg((size_t)M, 0);
    for (int32_t j = 0; j < M; ++j)
        for (size_t w = 0; w < wpr; ++w) deg[(size_t)j] += __builtin_popcountll(adj[(size_t)j * wpr + w]);
    std::vector<int32_t> order, level((size_t)M), nb;
    std::vector<char> done((size_t)M, 0);
    // breadth-first levels of the component of `root` among the frames not yet numbered; returns the last level's
    // frame of smallest degree and the depth
    std::vector<int32_t> stamp((size_t)M, 0); // visited in THIS search: stamp == bfs_id (no copy of an M-byte array per search)
    int32_t bfs_id = 0;
    auto bfs = [&](int32_t root, std::vector<int32_t>& out, int32_t& depth) -> int32_t {
        out.clear();
        out.push_back(root);
        ++bfs_id;
        struct Vis {
            const std::vector<char>& done;
            std::vector<int32_t>& stamp;
            int32_t id;
            struct Ref {
                Vis& v;
                size_t i;
                operator bool() const { return v.done[i] || v.stamp[i] == v.id; }
                Ref& operator=(int) { v.stamp[i] = v.id; return *this; }
            };
            Ref operator[](size_t i) { return Ref{ *this, i }; }
        } vis{ done, stamp, bfs_id };
        vis[(size_t)root] = 1;
        level[(size_t)root] = 0;
        for (size_t q = 0; q < out.size(); ++q) {
            const int32_t u = out[q];
            nb.clear();
            for (size_t w = 0; w < wpr; ++w)
                for (uint64_t bits = adj[(size_t)u * wpr + w]; bits; bits &= bits - 1) {
                    const int32_t v = (int32_t)(64 * w) + __builtin_ctzll(bits);
                    if (!vis[(size_t)v]) { vis[(size_t)v] = 1; nb.push_back(v); }
                }
            std::sort(nb.begin(), nb.end(), [&](int32_t a, int32_t b) { return deg[(size_t)a] != deg[(size_t)b] ? deg[(size_t)a] < deg[(size_t)b] : a < b; });
            for (int32_t v : nb) { level[(size_t)v] = level[(size_t)u] + 1; out.push_back(v); }
        }
        depth = level[(size_t)out.back()];
        int32_t far = out.back();
        for (size_t q = out.size(); q-- > 0 && level[(size_t)out[q]] == depth;)
            if (deg[(size_t)out[q]] < deg[(size_t)far] || (deg[(size_t)out[q]] == deg[(size_t)far] && out[q] < far)) far = out[q];
        return far;
    };
    std::vector<int32_t> comp;
    for (int32_t start = 0; start < M; ++start) {
        if (done[(size_t)start]) continue;
        if (deg[(size_t)start] == 0) { // a frame nobody shares a landmark with: a component of its own
            done[(size_t)start] = 1;
            order.push_back(start);
            continue;
        }
        int32_t root = start, depth = -1, d2 = 0;
        for (int it = 0; it < 4; ++it) { // George-Liu: walk to a frame of (nearly) greatest eccentricity
            const int32_t far = bfs(root, comp, d2);
            if (d2 <= depth) break;
            depth = d2;
            root = far;
        }
        bfs(root, comp, d2);
        for (int32_t v : comp) { done[(size_t)v] = 1; order.push_back(v); }
    }
    std::reverse(order.begin(), order.end());
    to_int.assign((size_t)M, 0);
    for (int32_t i = 0; i < M; ++i) to_int[(size_t)order[(size_t)i]] = i;
    int64_t bw_new = 0;
    bool differs = false;
    for (int32_t j = 0; j < M; ++j) differs = differs || to_int[(size_t)j] != j;
    for (int64_t i = 0; i < N; ++i) {
        int32_t lo = M, hi = -1;
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o) {
            lo = std::min(lo, to_int[(size_t)obs_frame[o]]);
            hi = std::max(hi, to_int[(size_t)obs_frame[o]]);
        }
        if (hi >= 0) bw_new = std::max<int64_t>(bw_new, hi - lo);
    }
    if (srk_debug()) fprintf(stderr, "srk_ba frame order: bandwidth %lld frames in the caller's order, %lld after reverse Cuthill-McKee\n", (long long)bw_nat, (long long)bw_new);
    return mode > 0 ? differs : 10 * bw_new <= 7 * bw_nat;
}

static void rearm_fusion(srk_ba* h);
static int upload_scene_impl(srk_ba* h, double f0, int64_t N, const double* pts_in, int32_t M,
                             const double* cam_R_in, const double* cam_T_in, const double* K_in, int shared_k,
                             const int64_t* row_ptr, const int32_t* obs_frame, const double* obs_uv,
                             int already_normalized);
// (the upload allocates host vectors and starts sorting threads: nothing of that may leave through the C ABI as an exception)
extern "C" int srk_ba_upload_scene(srk_ba* h, double f0, int64_t N, const double* pts_in, int32_t M,
                                   const double* cam_R_in, const double* cam_T_in, const double* K_in, int shared_k,
                                   const int64_t* row_ptr, const int32_t* obs_frame, const double* obs_uv,
                                   int already_normalized)
{
    try {
        return upload_scene_impl(h, f0, N, pts_in, M, cam_R_in, cam_T_in, K_in, shared_k, row_ptr, obs_frame, obs_uv, already_normalized);
    } catch (const std::bad_alloc&) {
        if (h) h->last_error = "upload: out of host memory", h->have_scene = false;
        return SRK_E_NOMEM;
    } catch (const std::exception& e) {
        if (h) h->last_error = std::string("upload: ") + e.what(), h->have_scene = false;
        return SRK_E_DEVICE;
    } catch (...) {
        if (h) h->last_error = "upload: unknown exception", h->have_scene = false;
        return SRK_E_DEVICE;
    }
}
static int upload_scene_impl(srk_ba* h, double f0, int64_t N, const double* pts_in, int32_t M,
                             const double* cam_R_in, const double* cam_T_in, const double* K_in, int shared_k,
                             const int64_t* row_ptr, const int32_t* obs_frame, const double* obs_uv,
                             int already_normalized)
{
    auto t_stage = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) { // SRK_DEBUG=1: where the host time of an upload goes
        if (!srk_debug()) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "srk_ba upload: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_stage).count());
        t_stage = now;
    };
    int rc = validate_scene(h, f0, N, pts_in, M, cam_R_in, cam_T_in, K_in, row_ptr, obs_frame, obs_uv);
    if (rc != SRK_OK) return rc;
    rearm_fusion(h);
    stage("validate");
    HIPCHK(h, hipSetDevice(h->device));
    h->have_scene = false;
    h->seen_global = -1;
    int64_t O = row_ptr[N];
    std::vector<double> pts(pts_in, pts_in + 3 * N), camR(cam_R_in, cam_R_in + 9 * (int64_t)M),
        camT(cam_T_in, cam_T_in + 3 * (int64_t)M);
    h->normalized_on_upload = false;
    if (!already_normalized) {
        // unity_t1_comp_ind_ = 1, unity_t1_comp_value_ = 1.0 (bundle-adj-kanatani.h:131-132); :680-682
        if (!srk_ba_normalize_scene(N, pts.data(), M, camR.data(), camT.data(), 1.0, 1, &h->nrm)) {
            h->last_error = "scene cannot be normalised (cam0->cam1 translation component ~ 0)";
            return 1; // reference: ComputeInplace returns false, status string stays empty
        }
        h->normalized_on_upload = true;
    } else {
        std::memset(&h->nrm, 0, sizeof h->nrm);
        h->nrm.R0[0] = h->nrm.R0[4] = h->nrm.R0[8] = 1;
        h->nrm.world_scale = 1;
    }
    std::vector<double> Kexp(9 * (int64_t)M);
    for (int32_t j = 0; j < M; ++j) std::memcpy(&Kexp[9 * (int64_t)j], shared_k ? K_in : K_in + 9 * (int64_t)j, 72);

    // ---- internal frame order (frame_reorder above; one rank only: shards would each find another order)
    std::vector<int32_t> of_fr;
    std::vector<double> ouv_fr;
    h->frame_int.clear();
    h->frame_user.clear();
    h->obs_rank.clear();
    int32_t g0 = 0, g1 = 1;
    {
        std::vector<int32_t> to_int;
        bool renumber = false;
        h->frame_order_supplied = false;
        if (!h->frame_order_given.empty()) {
            if ((int64_t)h->frame_order_given.size() != (int64_t)M) { h->last_error = "the supplied frame order is for another number of frames"; return SRK_E_ARGS; }
            to_int = h->frame_order_given;
            for (int32_t j = 0; j < M; ++j) renumber = renumber || to_int[(size_t)j] != j;
            h->frame_order_supplied = renumber;
        } else if (!(h->allreduce || h->comm))
            renumber = frame_reorder(h->frame_order_mode, N, M, row_ptr, obs_frame, to_int);
        if (renumber) {
            h->frame_int = to_int;
            h->frame_user.assign((size_t)M, 0);
            for (int32_t j = 0; j < M; ++j) h->frame_user[(size_t)to_int[(size_t)j]] = j;
            g0 = to_int[0];
            g1 = to_int[1];
            // cameras in the internal order; every landmark's observations re-sorted by internal frame
            std::vector<double> r2(camR.size()), t2(camT.size()), k2(Kexp.size());
            for (int32_t j = 0; j < M; ++j) {
                const int64_t u = h->frame_user[(size_t)j];
                std::memcpy(&r2[9 * (size_t)j], &camR[9 * (size_t)u], 72);
                std::memcpy(&t2[3 * (size_t)j], &camT[3 * (size_t)u], 24);
                std::memcpy(&k2[9 * (size_t)j], &Kexp[9 * (size_t)u], 72);
            }
            camR.swap(r2);
            camT.swap(t2);
            Kexp.swap(k2);
            of_fr.resize((size_t)O);
            ouv_fr.resize((size_t)(2 * O));
            h->obs_rank.resize((size_t)O);
            std::vector<std::pair<int32_t, int64_t>> key;
            for (int64_t i = 0; i < N; ++i) {
                key.clear();
                for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o) key.emplace_back(to_int[(size_t)obs_frame[o]], o);
                std::sort(key.begin(), key.end());
                for (size_t a = 0; a < key.size(); ++a) {
                    const int64_t dst = row_ptr[i] + (int64_t)a, src = key[a].second;
                    of_fr[(size_t)dst] = key[a].first;
                    ouv_fr[(size_t)(2 * dst)] = obs_uv[2 * src];
                    ouv_fr[(size_t)(2 * dst + 1)] = obs_uv[2 * src + 1];
                    h->obs_rank[(size_t)src] = (int32_t)a;
                }
            }
            obs_frame = of_fr.data();
            obs_uv = ouv_fr.data();
        }
    }

    // ---- internal landmark order: sorted by frame list, so that landmarks seeing exactly the same frames are
    // contiguous (the grouped Schur kernel accumulates a run of them in registers and flushes once)
    std::vector<int64_t> order((size_t)N);
    std::iota(order.begin(), order.end(), (int64_t)0);
    auto list_less = [&](int64_t x, int64_t y) {
        int64_t ox = row_ptr[x], oy = row_ptr[y];
        int64_t nx = row_ptr[x + 1] - ox, ny = row_ptr[y + 1] - oy;
        if (nx == 0 || ny == 0) return nx < ny;
        if (obs_frame[ox] != obs_frame[oy]) return obs_frame[ox] < obs_frame[oy];
        if (nx != ny) return nx < ny;
        for (int64_t k = 1; k < nx; ++k)
            if (obs_frame[ox + k] != obs_frame[oy + k]) return obs_frame[ox + k] < obs_frame[oy + k];
        return false;
    };
    stage("normalise, frame order");
    // (large scenes: chunks sorted by a few host threads, then merged pairwise -- both stable, so the order is the one a
    // single stable_sort gives)
    const int n_thr = N >= 32768 ? (int)std::min<unsigned>(8, std::max<unsigned>(1, std::thread::hardware_concurrency())) : 1;
    if (n_thr > 1) {
        std::vector<int64_t> cut((size_t)n_thr + 1);
        for (int t = 0; t <= n_thr; ++t) cut[(size_t)t] = N * t / n_thr;
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t)
            th.emplace_back([&, t] { std::stable_sort(order.begin() + cut[(size_t)t], order.begin() + cut[(size_t)t + 1], list_less); });
        for (auto& x : th) x.join();
        for (int w = 1; w < n_thr; w *= 2) {
            th.clear();
            for (int t = 0; t + w < n_thr; t += 2 * w)
                th.emplace_back([&, t, w] {
                    std::inplace_merge(order.begin() + cut[(size_t)t], order.begin() + cut[(size_t)(t + w)],
                                       order.begin() + cut[(size_t)std::min(t + 2 * w, n_thr)], list_less);
                });
            for (auto& x : th) x.join();
        }
    } else
        std::stable_sort(order.begin(), order.end(), list_less);
    stage("sort landmarks by frame list");
    h->perm = order;
    h->row_ptr_user.assign(row_ptr, row_ptr + N + 1);
    std::vector<int64_t> rp((size_t)N + 1, 0);
    std::vector<int32_t> of((size_t)O);
    std::vector<double> ouv((size_t)(2 * O)), ppts((size_t)(3 * N));
    for (int64_t i = 0; i < N; ++i) {
        const int64_t u = order[(size_t)i];
        rp[(size_t)i + 1] = rp[(size_t)i] + (row_ptr[u + 1] - row_ptr[u]);
    }
    auto permute_range = [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) {
            const int64_t u = order[(size_t)i];
            const int64_t cnt = row_ptr[u + 1] - row_ptr[u];
            std::memcpy(&of[(size_t)rp[(size_t)i]], obs_frame + row_ptr[u], (size_t)(4 * cnt));
            std::memcpy(&ouv[(size_t)(2 * rp[(size_t)i])], obs_uv + 2 * row_ptr[u], (size_t)(16 * cnt));
            std::memcpy(&ppts[(size_t)(3 * i)], &pts[(size_t)(3 * u)], 24);
        }
    };
    if (n_thr > 1) {
        std::vector<std::thread> th;
        for (int t = 0; t < n_thr; ++t) th.emplace_back(permute_range, N * t / n_thr, N * (t + 1) / n_thr);
        for (auto& x : th) x.join();
    } else
        permute_range(0, N);
    h->row_ptr_int = rp;
    pts.swap(ppts);
    row_ptr = rp.data();
    obs_frame = of.data();
    obs_uv = ouv.data();
    stage("permute observations");
    // Runs of consecutive landmarks (internal order) whose frame lists fit a common set of <= SRK_GRP_MAXNF_HOST
    // frames -> grouped Schur kernel: the run's blocks are accumulated over that UNION of frames, a landmark that does
    // not see one of them contributes zeros there.  Identical lists (the circle-grid scenes) are the special case
    // union == list ("uniform" run: no slot table needed); ragged feature tracks, where hardly two landmarks see
    // exactly the same frames, still share a window of frames.  Landmarks with more frames -> per-landmark kernel.
    std::vector<int32_t> grp_first, grp_count, grp_nf, grp_frames, gen_list, long_cand;
    std::vector<uint8_t> obs_slot((size_t)O, 0);
    std::vector<uint32_t> pt_mask((size_t)N, 0);
    int64_t n_wide = 0, n_mid = 0;
    {
        std::vector<int32_t> uni, merged;
        for (int64_t i = 0; i < N;) {
            const int64_t nfi = rp[(size_t)i + 1] - rp[(size_t)i];
            if (nfi == 0) { ++i; continue; }
            if (nfi > SRK_GRP_MAXNF_HOST) { long_cand.push_back((int32_t)i); ++i; continue; }
            const int64_t cap = nfi > SRK_GRP_NF1_HOST ? SRK_GRP_MAXNF_HOST : SRK_GRP_NF1_HOST;
            uni.assign(of.begin() + rp[(size_t)i], of.begin() + rp[(size_t)i + 1]);
            int64_t j = i + 1;
            while (j < N && j - i < SRK_GRP_MAXPTS_HOST) {
                const int64_t nfj = rp[(size_t)j + 1] - rp[(size_t)j];
                if (nfj == 0 || nfj > cap) break;
                // (the common case first: the same frame list as the run so far -- nothing to merge)
                if (nfj == (int64_t)uni.size() && std::equal(uni.begin(), uni.end(), of.begin() + rp[(size_t)j])) { ++j; continue; }
                merged.clear();
                std::set_union(uni.begin(), uni.end(), of.begin() + rp[(size_t)j], of.begin() + rp[(size_t)j + 1],
                               std::back_inserter(merged));
                if ((int64_t)merged.size() > cap) break;
                // a wider frame set costs every landmark of the run more flops; it only pays while the run is still
                // small against its one-off flush (~ the work of a dozen landmarks)
                if (merged.size() > uni.size() && j - i >= 24) break;
                uni.swap(merged);
                ++j;
            }
            bool uniform = true;
            for (int64_t p = i; p < j; ++p) {
                uint32_t mask = 0;
                if (rp[(size_t)p + 1] - rp[(size_t)p] == (int64_t)uni.size()) { // as many frames as the union: the union itself
                    for (int64_t o = rp[(size_t)p], k = 0; o < rp[(size_t)p + 1]; ++o, ++k) obs_slot[(size_t)o] = (uint8_t)k;
                    mask = (uint32_t)((1ull << uni.size()) - 1);
                } else {
                    for (int64_t o = rp[(size_t)p]; o < rp[(size_t)p + 1]; ++o) {
                        int slot = (int)(std::lower_bound(uni.begin(), uni.end(), of[(size_t)o]) - uni.begin());
                        obs_slot[(size_t)o] = (uint8_t)slot;
                        mask |= 1u << slot;
                    }
                    uniform = false;
                }
                pt_mask[(size_t)p] = mask;
            }
            grp_first.push_back((int32_t)i);
            grp_count.push_back((int32_t)(j - i));
            grp_nf.push_back(uniform ? (int32_t)uni.size() : -(int32_t)uni.size()); // negative = ragged run
            for (int k = 0; k < SRK_GRP_MAXNF_HOST; ++k) grp_frames.push_back(k < (int)uni.size() ? uni[(size_t)k] : -1);
            if ((int64_t)uni.size() > SRK_GRP_NF1_HOST) ++n_wide;
            else if ((int64_t)uni.size() > SRK_WS_NF_HOST) ++n_mid;
            i = j;
        }
    }
    // A small scene (the dino set: 36 frames, 4983 points; the point sets of the multi-view-factorisation calls) has a few
    // dozen full-length runs: a few dozen workgroups on 256 CUs, each staging up to 128 landmarks four at a time -- the sum
    // takes as long as one workgroup's 32 rounds (config 1: 114 us for 16 k observations).  Such runs are cut into `split`
    // equal parts over the SAME frame set (slots and masks stay as they are; a cut that left a remainder to merge with the
    // next frame list made ragged and wider runs: measured, worse).  What stops the cut: every part flushes the whole tile
    // triangle of its frame set, and the parts of one round of workgroups flush together -- the fp64 atomics of a burst
    // drain at ~0.5 TB/s (100 frames x 5000 points, 20-frame tracks: 74 / 72 / 78 / 96 / 155 us with 1 / 2 / 3 / 4 / 8
    // parts; config 1, 4-frame tracks: 114 / 68 / 57 / 49 us with 1 / 2 / 3 / 4 parts, `tools/run_len_probe.py`) -- and a
    // second round of workgroups.  The model below is that shape: a double round of eight landmarks ~5.5 us, 2 KB a tile.
    {
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        auto cost = [&](int split, int64_t* n_parts) {
            double worst = 0, tiles = 0;
            int64_t parts = 0;
            for (size_t r = 0; r < grp_first.size(); ++r) {
                const int64_t nfu = std::abs(grp_nf[r]), nt = (10 * nfu + 15) / 16, len = (grp_count[r] + split - 1) / split;
                const int64_t k = std::min<int64_t>(split, (grp_count[r] + len - 1) / len);
                parts += k;
                tiles += (double)(k * nt * (nt + 1) / 2);
                worst = std::max(worst, 4.5 + 5.5 * (double)((len + 7) / 8));
            }
            if (n_parts) *n_parts = parts;
            return worst * (double)((parts + cus - 1) / cus) + tiles * 2048.0 / 0.5e6; // us
        };
        int split = 1;
        if (!grp_first.empty() && (int)grp_first.size() < cus) {
            double best = cost(1, nullptr);
            for (int sp = 2; sp <= 8; ++sp) {
                const double c = cost(sp, nullptr);
                if (c < 0.9 * best) best = c, split = sp;
            }
        }
#ifdef SRK_DEV
        if (const char* e = getenv("SRK_SCHUR_RUN_SPLIT")) split = std::max(1, std::min(16, atoi(e))); // development: fixed cut
#endif
        if (split > 1) {
            std::vector<int32_t> f2, c2, n2, fr2;
            for (size_t r = 0; r < grp_first.size(); ++r) {
                const int64_t len = (grp_count[r] + split - 1) / split;
                for (int64_t a = 0; a < grp_count[r]; a += len) {
                    f2.push_back(grp_first[r] + (int32_t)a);
                    c2.push_back((int32_t)std::min<int64_t>(len, grp_count[r] - a));
                    n2.push_back(grp_nf[r]);
                    fr2.insert(fr2.end(), grp_frames.begin() + (ptrdiff_t)(r * SRK_GRP_MAXNF_HOST), grp_frames.begin() + (ptrdiff_t)((r + 1) * SRK_GRP_MAXNF_HOST));
                    if (a > 0) {
                        if (std::abs(grp_nf[r]) > SRK_GRP_NF1_HOST) ++n_wide;
                        else if (std::abs(grp_nf[r]) > SRK_WS_NF_HOST) ++n_mid;
                    }
                }
            }
            grp_first.swap(f2); grp_count.swap(c2); grp_nf.swap(n2); grp_frames.swap(fr2);
        }
    }
    // Long tracks (> SRK_GRP_MAXNF_HOST frames; every track of the demos' all-visible scenes): runs of consecutive
    // candidates over the union of their frame lists (<= SRK_LONG_MAXNF_HOST frames), cut into blocks of 8 frames; one
    // work item per pair of blocks (k_schur_long).  A track over more frames than a run holds keeps the per-landmark kernel.
    std::vector<int32_t> lg_item, lg_np, lg_nf, lg_pts, lg_frames, lg_obs;
    std::vector<int64_t> lg_obs_off;
#ifdef SRK_DEV
    const bool no_long = getenv("SRK_SCHUR_NO_LONG") != nullptr; // development: everything through the per-landmark kernel
#else
    const bool no_long = false;
#endif
    if (no_long) {
        gen_list.insert(gen_list.end(), long_cand.begin(), long_cand.end());
        long_cand.clear();
    }
    {
        std::vector<int32_t> uni, merged;
        for (size_t ci = 0; ci < long_cand.size();) {
            const int64_t i = long_cand[ci];
            const int64_t nfi = rp[(size_t)i + 1] - rp[(size_t)i];
            if (nfi > SRK_LONG_MAXNF_HOST) { gen_list.push_back((int32_t)i); ++ci; continue; }
            uni.assign(of.begin() + rp[(size_t)i], of.begin() + rp[(size_t)i + 1]);
            size_t cj = ci + 1;
            while (cj < long_cand.size() && cj - ci < SRK_LONG_PTS_HOST) {
                const int64_t j = long_cand[cj];
                if (rp[(size_t)j + 1] - rp[(size_t)j] > SRK_LONG_MAXNF_HOST) break;
                merged.clear();
                std::set_union(uni.begin(), uni.end(), of.begin() + rp[(size_t)j], of.begin() + rp[(size_t)j + 1],
                               std::back_inserter(merged));
                if ((int64_t)merged.size() > SRK_LONG_MAXNF_HOST) break;
                // every landmark of the run pays for the whole union: let it grow freely only while the run is small
                if (merged.size() > uni.size() && cj - ci >= 16 && (int64_t)merged.size() > nfi + nfi / 4 + SRK_LONG_FB_HOST) break;
                uni.swap(merged);
                ++cj;
            }
            const int nfu = (int)uni.size();
            lg_np.push_back((int32_t)(cj - ci));
            lg_nf.push_back((int32_t)nfu);
            for (int k = 0; k < SRK_LONG_PTS_HOST; ++k) lg_pts.push_back(ci + k < cj ? long_cand[ci + (size_t)k] : 0);
            for (int k = 0; k < SRK_LONG_MAXNF_HOST; ++k) lg_frames.push_back(k < nfu ? uni[(size_t)k] : -1);
            ci = cj;
        }
    }
    // Frame blocks of 8 or of 16 frames (round 4).  A workgroup stages both blocks of its pair for every landmark of the run:
    // with 8-frame blocks that is two staged blocks for 25 MFMA tiles and the kernel spent its time staging (SQ counters on
    // 200 frames x 20 000 points, every point in every frame: 7 vector-ALU, 0.85 memory and 0.9 LDS instructions per MFMA,
    // matrix pipes 33 % busy); a pair of 16-frame blocks is two staged blocks for 100 tiles.  The larger blocks need enough
    // pairs to fill the chip: small scenes (the 36- and 60-frame demo scenes) keep the 8-frame blocks.
    {
        int64_t items16 = 0, nf_sum = 0;
        for (size_t r = 0; r < lg_np.size(); ++r) {
            const int64_t nb16 = (lg_nf[r] + 15) / 16;
            items16 += nb16 * (nb16 + 1) / 2;
            nf_sum += lg_nf[r];
        }
        // (and frame sets long enough that the padding to 16 and the coarser diagonal pairs do not eat the gain.  Counting a
        // staged block-frame as ~3 MFMA tiles -- what the counters above say -- 40 frames cost the same either way, 64 frames
        // 30 % less with the larger blocks)
        h->long_fb = (items16 >= 1024 && nf_sum >= 56 * (int64_t)lg_np.size()) ? 16 : SRK_LONG_FB_HOST;
        const int FBh = h->long_fb;
        for (size_t r = 0; r < lg_np.size(); ++r) {
            const int32_t run = (int32_t)r;
            const int nfu = lg_nf[r], nb = (nfu + FBh - 1) / FBh, nfp = nb * FBh;
            const int32_t* uni_b = lg_frames.data() + r * SRK_LONG_MAXNF_HOST;
            lg_obs_off.push_back((int64_t)lg_obs.size());
            for (int k = 0; k < lg_np[r]; ++k) {
                const int64_t p = lg_pts[r * SRK_LONG_PTS_HOST + (size_t)k];
                const size_t base = lg_obs.size();
                lg_obs.resize(base + (size_t)nfp, -1);
                for (int64_t o = rp[(size_t)p]; o < rp[(size_t)p + 1]; ++o) {
                    const int slot = (int)(std::lower_bound(uni_b, uni_b + nfu, of[(size_t)o]) - uni_b);
                    lg_obs[base + (size_t)slot] = (int32_t)o;
                }
            }
            for (int a = 0; a < nb; ++a)
                for (int b = 0; b <= a; ++b) {
                    lg_item.push_back(run); lg_item.push_back(a); lg_item.push_back(b); lg_item.push_back(0);
                }
        }
    }
    h->n_long_runs = (int64_t)lg_np.size();
    h->n_long_items = (int64_t)lg_item.size() / 4;
    h->n_groups = (int64_t)grp_first.size();
    h->n_groups_wide = n_wide;
    h->n_groups_mid = n_mid;
    h->n_mm_uniform = h->n_mm_ragged = 0;
    for (int32_t v : grp_nf) {
        if (v > 0 && v <= SRK_WS_NF_HOST) ++h->n_mm_uniform;
        if (v < 0 && -v <= SRK_WS_NF_HOST) ++h->n_mm_ragged;
    }
    h->n_generic = (int64_t)gen_list.size();
    SrkDims d{};
    d.N = N;
    d.M = M;
    d.O = O;
    d.Os = ((O + 63) / 64) * 64;
    if (d.Os == 0) d.Os = 64;
    d.Ns = ((N + 63) / 64) * 64;
    if (d.Ns == 0) d.Ns = 64;
    d.ld = ((10 * (int64_t)M + SRK_CHOL_NB - 1) / SRK_CHOL_NB) * SRK_CHOL_NB;
    d.comp = 1;
    d.g0 = g0;
    d.g1 = g1;
    d.w_f32 = h->store_f32 ? 1 : 0;
    h->d = d;
    h->f0 = f0;

    stage("Schur runs");
    // observation side tables: obs -> point, and the frame-major copy (ordered by frame, then landmark)
    std::vector<int32_t> obs_pt((size_t)O);
    std::vector<int64_t> col_ptr((size_t)M + 1, 0);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o) {
            obs_pt[(size_t)o] = (int32_t)i;
            col_ptr[(size_t)obs_frame[o] + 1]++;
        }
    h->max_frame_obs = 0;
    for (int32_t j = 0; j < M; ++j) {
        if (col_ptr[(size_t)j + 1] > h->max_frame_obs) h->max_frame_obs = col_ptr[(size_t)j + 1];
        col_ptr[(size_t)j + 1] += col_ptr[(size_t)j];
    }

    // frame range of every SRK_JF_OBS_HOST-observation workgroup of the fused Jacobian kernel
    std::vector<int32_t> wg_jmin;
    h->jac_fused = true;
    for (int64_t o0 = 0; o0 < O; o0 += SRK_JF_OBS_HOST) {
        int32_t lo = obs_frame[o0], hi = obs_frame[o0];
        for (int64_t o = o0; o < std::min<int64_t>(O, o0 + SRK_JF_OBS_HOST); ++o) {
            lo = std::min(lo, obs_frame[o]);
            hi = std::max(hi, obs_frame[o]);
        }
        if (hi - lo >= SRK_JF_SLOTS_HOST) h->jac_fused = false;
        int64_t olast = std::min<int64_t>(O, o0 + SRK_JF_OBS_HOST) - 1;
        if (obs_pt[(size_t)olast] - obs_pt[(size_t)o0] + 1 > SRK_JF_PMAX_HOST) h->jac_fused = false;
        wg_jmin.push_back(lo);
    }

    stage("side tables, frame windows");
    // tasks of the run-based Jacobian kernel: maximal runs of consecutive landmarks (internal order) with identical
    // frame lists, cut into pieces (a multiple of the landmarks per step).  The kernel holds two 4-wave workgroups per
    // CU (a wave keeps 61 frame sums and a step of look-ahead in ~250 registers); with about one task per wave slot
    // every task runs at the same time and the stores of all of them share HBM from start to end (with 1.4 rounds of
    // shorter tasks the second round ran at 2 TB/s).
    std::vector<int32_t> jr_first, jr_count, jr_jmin;
    // (the kernel addresses W with 32-bit byte offsets inside a plane and inside each half of the 30 planes)
    h->jac_runs = h->jac_mode != 0 && O < (int64_t)1 << 27;
    h->jr_min_nf = 64;
    int64_t jr_piece_target = SRK_JR_TASK_PTS_MIN_HOST;
    {
        int cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        int64_t with_obs = 0;
        for (int64_t i = 0; i < N; ++i) with_obs += rp[(size_t)i + 1] > rp[(size_t)i];
        const int64_t slots = 8 * (int64_t)cus; // waves resident at once
        jr_piece_target = std::min<int64_t>(SRK_JR_TASK_PTS_MAX_HOST, std::max<int64_t>(SRK_JR_TASK_PTS_MIN_HOST, (with_obs + slots - 1) / slots));
    }
    for (int64_t i = 0; i < N && h->jac_runs;) {
        const int64_t nf = rp[(size_t)i + 1] - rp[(size_t)i];
        if (nf == 0) { ++i; continue; }
        if (nf > 64) { h->jac_runs = false; break; }
        h->jr_min_nf = std::min<int32_t>(h->jr_min_nf, (int32_t)nf);
        int64_t j = i + 1;
        while (j < N && rp[(size_t)j + 1] - rp[(size_t)j] == nf &&
               std::equal(of.begin() + rp[(size_t)i], of.begin() + rp[(size_t)i + 1], of.begin() + rp[(size_t)j])) ++j;
        const int64_t g = 64 / nf, len = j - i;
        const int64_t most = SRK_JR_TASK_PTS_MAX_HOST / g * g; // the kernel stages a task's landmarks in LDS
        const int64_t pieces = std::max<int64_t>((len + most - 1) / most, (len + jr_piece_target / 2) / jr_piece_target);
        const int64_t piece = std::max<int64_t>(g, ((len + pieces - 1) / pieces + g - 1) / g * g);
        for (int64_t a = i; a < j; a += piece) {
            jr_first.push_back((int32_t)a);
            jr_count.push_back((int32_t)std::min<int64_t>(piece, j - a));
        }
        i = j;
    }
    if (h->jac_runs) {
        // long enough to pay: a task flushes 65 sums per lane, which an iteration of the per-observation kernel costs.
        // (Round 3, tools/jac_modes.py: at 240 observations a task -- C2 -- the run kernel takes 45 us where the fused
        // per-observation kernel takes 64; tasks of one or two ragged landmarks, ~20 observations, 118 against 69.)
        if (jr_first.empty() || (h->jac_mode != 1 && O / (int64_t)jr_first.size() < 128)) h->jac_runs = false;
        for (size_t t0 = 0; t0 < jr_first.size() && h->jac_runs; t0 += 4) {
            int32_t lo = M, hi = -1;
            for (size_t t = t0; t < std::min(jr_first.size(), t0 + 4); ++t) {
                const int64_t a = rp[(size_t)jr_first[t]], b = rp[(size_t)jr_first[t] + 1];
                lo = std::min(lo, of[(size_t)a]);
                hi = std::max(hi, of[(size_t)b - 1]);
            }
            if (hi - lo >= SRK_JF_SLOTS_HOST) h->jac_runs = false;
            jr_jmin.push_back(lo);
        }
    }
    // Ragged tracks: hardly two landmarks see exactly the same frames, so the runs above are too short to pay -- but the
    // Schur kernel's runs (consecutive landmarks over the UNION of their frame lists, <= 24 frames, masks) are not.  The same
    // kernel with a lane per (landmark, frame slot) CELL: tasks = pieces of those runs.  Needs every landmark in a run.
    std::vector<int32_t> jr_group;
    h->jac_runs_masked = false;
    // The runs the union tasks are pieces of: the Schur kernels' (every landmark is in one when no track is longer than
    // SRK_GRP_MAXNF_HOST frames), else runs of the derivative kernel's own over unions of <= 32 frames (a lane's observation is
    // found from a 32-bit mask), when no track is longer than that.
    std::vector<int32_t> jd_first, jd_count, jd_nf, jd_frames;
    std::vector<uint32_t> jd_mask;
    h->jr_own_runs = false;
    constexpr int JD_MAXNF = 32;
    if ((!h->jac_runs || h->jac_mode == 2) && h->jac_mode != 0 && O < (int64_t)1 << 27 && !long_cand.empty() && gen_list.empty()) {
        bool fits = true;
        for (int32_t p : long_cand) fits = fits && rp[(size_t)p + 1] - rp[(size_t)p] <= JD_MAXNF;
        if (fits) {
            jd_mask.assign((size_t)N, 0);
            std::vector<int32_t> uni, merged;
            for (int64_t i = 0; i < N;) {
                const int64_t nfi = rp[(size_t)i + 1] - rp[(size_t)i];
                if (nfi == 0) { ++i; continue; }
                uni.assign(of.begin() + rp[(size_t)i], of.begin() + rp[(size_t)i + 1]);
                int64_t j = i + 1;
                while (j < N && j - i < SRK_GRP_MAXPTS_HOST) {
                    const int64_t nfj = rp[(size_t)j + 1] - rp[(size_t)j];
                    if (nfj == 0) break;
                    if (nfj == (int64_t)uni.size() && std::equal(uni.begin(), uni.end(), of.begin() + rp[(size_t)j])) { ++j; continue; }
                    merged.clear();
                    std::set_union(uni.begin(), uni.end(), of.begin() + rp[(size_t)j], of.begin() + rp[(size_t)j + 1], std::back_inserter(merged));
                    if ((int64_t)merged.size() > JD_MAXNF) break;
                    if (merged.size() > uni.size() && j - i >= 24) break; // (a wider set costs every landmark of the run lanes)
                    uni.swap(merged);
                    ++j;
                }
                bool uniform = true;
                for (int64_t p = i; p < j; ++p) {
                    uint32_t mask = 0;
                    for (int64_t o = rp[(size_t)p]; o < rp[(size_t)p + 1]; ++o)
                        mask |= 1u << (int)(std::lower_bound(uni.begin(), uni.end(), of[(size_t)o]) - uni.begin());
                    uniform = uniform && rp[(size_t)p + 1] - rp[(size_t)p] == (int64_t)uni.size();
                    jd_mask[(size_t)p] = mask;
                }
                jd_first.push_back((int32_t)i);
                jd_count.push_back((int32_t)(j - i));
                jd_nf.push_back(uniform ? (int32_t)uni.size() : -(int32_t)uni.size());
                for (int k = 0; k < JD_MAXNF; ++k) jd_frames.push_back(k < (int)uni.size() ? uni[(size_t)k] : -1);
                i = j;
            }
            h->jr_own_runs = true;
        }
    }
    const std::vector<int32_t>& rn_first = h->jr_own_runs ? jd_first : grp_first;
    const std::vector<int32_t>& rn_count = h->jr_own_runs ? jd_count : grp_count;
    const std::vector<int32_t>& rn_nf = h->jr_own_runs ? jd_nf : grp_nf;
    const std::vector<int32_t>& rn_frames = h->jr_own_runs ? jd_frames : grp_frames;
    const size_t rn_stride = h->jr_own_runs ? (size_t)JD_MAXNF : (size_t)SRK_GRP_MAXNF_HOST;
    if ((!h->jac_runs || h->jac_mode == 2) && h->jac_mode != 0 && O < (int64_t)1 << 27 && (h->jr_own_runs || long_cand.empty()) && gen_list.empty() && !rn_first.empty()) {
        const bool uniform_ok = h->jac_runs; // (mode 2: the union tasks are preferred, the uniform ones stay as the fallback)
        std::vector<int32_t> u_first, u_count, u_jmin;
        u_first.swap(jr_first); u_count.swap(jr_count); u_jmin.swap(jr_jmin);
        jr_first.clear();
        jr_count.clear();
        jr_jmin.clear();
        for (size_t gi = 0; gi < rn_first.size(); ++gi) {
            const int64_t nfu = std::abs(rn_nf[gi]), g = 64 / nfu, len = rn_count[gi];
            const int64_t most = SRK_JR_TASK_PTS_MAX_HOST / g * g;
            const int64_t pieces = std::max<int64_t>((len + most - 1) / most, (len + jr_piece_target / 2) / jr_piece_target);
            const int64_t piece = std::max<int64_t>(g, ((len + pieces - 1) / pieces + g - 1) / g * g);
            for (int64_t a = 0; a < len; a += piece) {
                jr_first.push_back(rn_first[gi] + (int32_t)a);
                jr_count.push_back((int32_t)std::min<int64_t>(piece, len - a));
                jr_group.push_back((int32_t)gi);
            }
        }
        bool ok = h->jac_mode >= 1 || O / (int64_t)jr_first.size() >= 32; // (the dino stand-in: 40 a task, 31 against 60 us)
        for (size_t t0 = 0; t0 < jr_first.size() && ok; t0 += 4) {
            int32_t lo = M, hi = -1;
            for (size_t t = t0; t < std::min(jr_first.size(), t0 + 4); ++t) {
                const size_t gi = (size_t)jr_group[t];
                lo = std::min(lo, rn_frames[gi * rn_stride]);
                hi = std::max(hi, rn_frames[gi * rn_stride + (size_t)std::abs(rn_nf[gi]) - 1]);
            }
            if (hi - lo >= SRK_JF_SLOTS_HOST) ok = false;
            jr_jmin.push_back(lo);
        }
        h->jac_runs_masked = ok;
        if (!ok) { // back to the uniform tasks (if they were usable)
            jr_first.swap(u_first); jr_count.swap(u_count); jr_jmin.swap(u_jmin);
            jr_group.clear();
            h->jac_runs = uniform_ok;
            h->jr_own_runs = false;
        } else
            h->jac_runs = true;
    }
    h->jr_tasks = h->jac_runs ? (int32_t)jr_first.size() : 0;

    // ---- deterministic mode: tables of the ordered second passes.  Covered: scenes whose landmarks all take the run-based
    // derivative kernel and the MFMA Schur kernel (tracks over at most SRK_WS_NF_HOST frames, fp64 run sums).
    std::vector<int32_t> dj_ptr, dj_ent, ds_pair_ptr, ds_pair_fa, ds_pair_fb, ds_pair_ent, ds_f_ptr, ds_f_ent;
    h->det_active = false;
    if (h->deterministic && h->jac_runs && n_wide == 0 && n_mid == 0 && lg_item.empty() && gen_list.empty() && !h->schur_fp32 &&
        !grp_first.empty() && grp_first.size() < ((size_t)1 << 20) && jr_first.size() < ((size_t)1 << 25)) {
        bool all_mm = true;
        for (int32_t v : grp_nf) all_mm = all_mm && std::abs(v) <= SRK_WS_NF_HOST;
        if (all_mm) {
            // derivative tasks by frame
            dj_ptr.assign((size_t)M + 1, 0);
            auto task_frames = [&](size_t t, const int32_t*& fr) -> int {
                if (!jr_group.empty()) {
                    const size_t gi = (size_t)jr_group[t];
                    fr = grp_frames.data() + gi * SRK_GRP_MAXNF_HOST;
                    return std::abs(grp_nf[gi]);
                }
                fr = of.data() + rp[(size_t)jr_first[t]];
                return (int)(rp[(size_t)jr_first[t] + 1] - rp[(size_t)jr_first[t]]);
            };
            for (size_t t = 0; t < jr_first.size(); ++t) {
                const int32_t* fr;
                const int nf = task_frames(t, fr);
                for (int f = 0; f < nf; ++f) ++dj_ptr[(size_t)fr[f] + 1];
            }
            for (int32_t j = 0; j < M; ++j) dj_ptr[(size_t)j + 1] += dj_ptr[(size_t)j];
            dj_ent.resize((size_t)dj_ptr[(size_t)M]);
            {
                std::vector<int32_t> fill(dj_ptr.begin(), dj_ptr.end() - 1);
                for (size_t t = 0; t < jr_first.size(); ++t) {
                    const int32_t* fr;
                    const int nf = task_frames(t, fr);
                    for (int f = 0; f < nf; ++f) dj_ent[(size_t)fill[(size_t)fr[f]]++] = (int32_t)(t * 64 + (size_t)f);
                }
            }
            // Schur runs by block (fa >= fb) and by frame
            std::vector<std::pair<int64_t, int32_t>> ents;
            ds_f_ptr.assign((size_t)M + 1, 0);
            for (size_t gi = 0; gi < grp_first.size(); ++gi) {
                const int nf = std::abs(grp_nf[gi]);
                const int32_t* fr = grp_frames.data() + gi * SRK_GRP_MAXNF_HOST;
                for (int sa = 0; sa < nf; ++sa) {
                    ++ds_f_ptr[(size_t)fr[sa] + 1];
                    for (int sb = 0; sb <= sa; ++sb)
                        ents.emplace_back((int64_t)fr[sa] * M + fr[sb], (int32_t)((uint32_t)gi | (uint32_t)sa << 20 | (uint32_t)sb << 25));
                }
            }
            std::stable_sort(ents.begin(), ents.end(), [](const std::pair<int64_t, int32_t>& x, const std::pair<int64_t, int32_t>& y) { return x.first < y.first; });
            ds_pair_ent.reserve(ents.size());
            for (size_t e = 0; e < ents.size(); ++e) {
                if (e == 0 || ents[e].first != ents[e - 1].first) {
                    ds_pair_ptr.push_back((int32_t)e);
                    ds_pair_fa.push_back((int32_t)(ents[e].first / M));
                    ds_pair_fb.push_back((int32_t)(ents[e].first % M));
                }
                ds_pair_ent.push_back(ents[e].second);
            }
            ds_pair_ptr.push_back((int32_t)ents.size());
            for (int32_t j = 0; j < M; ++j) ds_f_ptr[(size_t)j + 1] += ds_f_ptr[(size_t)j];
            ds_f_ent.resize((size_t)ds_f_ptr[(size_t)M]);
            {
                std::vector<int32_t> fill(ds_f_ptr.begin(), ds_f_ptr.end() - 1);
                for (size_t gi = 0; gi < grp_first.size(); ++gi) {
                    const int nf = std::abs(grp_nf[gi]);
                    const int32_t* fr = grp_frames.data() + gi * SRK_GRP_MAXNF_HOST;
                    for (int sa = 0; sa < nf; ++sa) ds_f_ent[(size_t)fill[(size_t)fr[sa]]++] = (int32_t)((uint32_t)gi | (uint32_t)sa << 20);
                }
            }
            h->ds_n_pairs = (int32_t)ds_pair_fa.size();
            h->det_active = true;
        }
    }
    stage("deterministic-mode tables");

    // the frame-major copy of the observations (ordered by frame, then landmark): only the two-kernel derivative path reads it
    const bool need_frame_major = !h->jac_runs && !h->jac_fused;
    std::vector<int32_t> fobs_pt(need_frame_major ? (size_t)O : 0);
    std::vector<double> fobs_uv(need_frame_major ? (size_t)(2 * O) : 0);
    if (need_frame_major) {
        std::vector<int64_t> fill(col_ptr.begin(), col_ptr.end() - 1);
        for (int64_t o = 0; o < O; ++o) {
            int64_t k = fill[(size_t)obs_frame[o]]++;
            fobs_pt[(size_t)k] = obs_pt[(size_t)o];
            fobs_uv[(size_t)(2 * k)] = obs_uv[2 * o];
            fobs_uv[(size_t)(2 * k + 1)] = obs_uv[2 * o + 1];
        }
    }
    stage("derivative tasks");
#define ALLOC(buf, bytes)                              \
    do {                                               \
        int _r = dev_alloc(h, (buf), (size_t)(bytes)); \
        if (_r != SRK_OK) return _r;                   \
    } while (0)
    // several ranks, damping-parallel schedule: one slot per damping factor of a round (at most three)
    const bool multi_upload = h->allreduce || h->comm;
    // (srk_ba_set_speculation(h, 0) = strictly one attempt at a time, with several ranks as well: one slot, hence the
    // all-reduce schedule without pairs)
    const int n_slots = (multi_upload && h->dp_schedule && h->speculate && (h->world >= 2 || h->dp_force))
                            ? (h->dp_force ? SRK_SLOTS : std::min(SRK_SLOTS, h->world))
                            : (h->speculate ? 2 : 1);
    for (int w = 0; w < SRK_SLOTS + 1; ++w) {
        if (w > n_slots) continue;
        ALLOC(h->pts[w], 24 * N);
        ALLOC(h->camR[w], 72 * (int64_t)M);
        ALLOC(h->camT[w], 24 * (int64_t)M);
        ALLOC(h->cam[w], 8 * SRK_CAM_PACK * (int64_t)M);
    }
    ALLOC(h->pts0, 24 * N);
    ALLOC(h->camR0, 72 * (int64_t)M);
    ALLOC(h->camT0, 24 * (int64_t)M);
    ALLOC(h->K, 72 * (int64_t)M);
    ALLOC(h->row_ptr, 8 * (N + 1));
    ALLOC(h->obs_frame, 4 * O);
    ALLOC(h->obs_pt, 4 * O);
    ALLOC(h->obs_uv, 16 * O);
    ALLOC(h->col_ptr, 8 * ((int64_t)M + 1));
    ALLOC(h->fobs_pt, 4 * fobs_pt.size());
    ALLOC(h->fobs_uv, 8 * fobs_uv.size());
    ALLOC(h->W, (d.w_f32 ? 4 : 8) * SRK_WF_PLANES * d.Os); // the 21 rank-2 factors of every point-frame block, fp64 or (opt-in) float
    ALLOC(h->Vg, 8 * 9 * d.Ns);
    ALLOC(h->Ug, 8 * SRK_UG * (int64_t)M);
    select_attempt(h, 0);
    for (int sl = 0; sl < SRK_SLOTS; ++sl) {
        srk_ba::Attempt& a = h->att[sl];
        a.allocated = sl < n_slots;
        if (!a.allocated) continue;
        if (h->det_active) {
            ALLOC(a.det_stage, 8 * (int64_t)SRK_DET_STRIDE * (int64_t)grp_first.size());
            ALLOC(a.det_rhs, 8 * (int64_t)SRK_DET_LD * (int64_t)grp_first.size());
        }
        ALLOC(a.S, 8 * d.ld * d.ld);
        ALLOC(a.rhs, 8 * d.ld);
        ALLOC(a.wy, 8 * 2 * d.ld);
        ALLOC(a.dc, 8 * d.ld);
        ALLOC(a.acc, 8 * 3 * d.Ns + 64);
        ALLOC(a.dx, 24 * N);
        ALLOC(a.err_partial, 8 * std::max<int64_t>(1024, srk_error_partials_staged(d)));
        ALLOC(a.info, 64);
        ALLOC(a.dinv, 8 * 64 * d.ld);
        ALLOC(a.irr, 4 * (N + 2));
        HIPCHK(h, hipMemsetAsync(a.irr.p, 0, 4, h->stream));
        if (!a.sync_flags.p) { // flag words of the fused outer-step kernel: zeroed ONCE (they hold launch epochs)
            ALLOC(a.sync_flags, 4 * SRK_SYNC_WORDS);
            HIPCHK(h, hipMemset(a.sync_flags.p, 0, 4 * SRK_SYNC_WORDS));
        }
        a.sync.flags = P<unsigned>(a.sync_flags);
        a.sync.fused = h->chol_fused;
    }
    ALLOC(h->grp_first, 4 * grp_first.size());
    ALLOC(h->grp_count, 4 * grp_count.size());
    ALLOC(h->grp_nf, 4 * grp_nf.size());
    ALLOC(h->grp_frames, 4 * grp_frames.size());
    ALLOC(h->obs_slot, obs_slot.size());
    ALLOC(h->pt_mask, 4 * pt_mask.size());
    ALLOC(h->gen_list, 4 * gen_list.size());
    ALLOC(h->lg_item, 4 * lg_item.size());
    ALLOC(h->lg_np, 4 * lg_np.size());
    ALLOC(h->lg_nf, 4 * lg_nf.size());
    ALLOC(h->lg_pts, 4 * lg_pts.size());
    ALLOC(h->lg_frames, 4 * lg_frames.size());
    ALLOC(h->lg_obs_off, 8 * lg_obs_off.size());
    ALLOC(h->lg_obs, 4 * lg_obs.size());
    ALLOC(h->wg_jmin, 4 * wg_jmin.size());
    if (h->jac_runs) {
        ALLOC(h->jr_first, 4 * jr_first.size());
        ALLOC(h->jr_count, 4 * jr_count.size());
        ALLOC(h->jr_jmin, 4 * jr_jmin.size());
        if (h->jac_runs_masked) ALLOC(h->jr_group, 4 * jr_group.size());
    }
    if (h->det_active) {
        ALLOC(h->dj_ptr, 4 * dj_ptr.size());
        ALLOC(h->dj_ent, 4 * std::max<size_t>(dj_ent.size(), 1));
        ALLOC(h->dj_stage, 8 * (int64_t)SRK_UG * 64 * (int64_t)jr_first.size());
        ALLOC(h->ds_pair_ptr, 4 * ds_pair_ptr.size());
        ALLOC(h->ds_pair_fa, 4 * std::max<size_t>(ds_pair_fa.size(), 1));
        ALLOC(h->ds_pair_fb, 4 * std::max<size_t>(ds_pair_fb.size(), 1));
        ALLOC(h->ds_pair_ent, 4 * std::max<size_t>(ds_pair_ent.size(), 1));
        ALLOC(h->ds_f_ptr, 4 * ds_f_ptr.size());
        ALLOC(h->ds_f_ent, 4 * std::max<size_t>(ds_f_ent.size(), 1));
    }
    if (h->jr_own_runs) {
        ALLOC(h->jd_nf, 4 * jd_nf.size());
        ALLOC(h->jd_frames, 4 * jd_frames.size());
        ALLOC(h->jd_mask, 4 * jd_mask.size());
    }
#undef ALLOC
    hipStream_t s = h->stream;
    stage("device allocations");
#define H2D(buf, src, bytes)                                                                               \
    do {                                                                                                   \
        if ((bytes) > 0) HIPCHK(h, hipMemcpyAsync((buf).p, (src), (size_t)(bytes), hipMemcpyHostToDevice, s)); \
    } while (0)
    H2D(h->pts[0], pts.data(), 24 * N);
    H2D(h->camR[0], camR.data(), 72 * (int64_t)M);
    H2D(h->camT[0], camT.data(), 24 * (int64_t)M);
    H2D(h->pts0, pts.data(), 24 * N);
    H2D(h->camR0, camR.data(), 72 * (int64_t)M);
    H2D(h->camT0, camT.data(), 24 * (int64_t)M);
    H2D(h->K, Kexp.data(), 72 * (int64_t)M);
    H2D(h->row_ptr, row_ptr, 8 * (N + 1));
    H2D(h->obs_frame, obs_frame, 4 * O);
    H2D(h->obs_pt, obs_pt.data(), 4 * O);
    H2D(h->obs_uv, obs_uv, 16 * O);
    H2D(h->col_ptr, col_ptr.data(), 8 * ((int64_t)M + 1));
    H2D(h->fobs_pt, fobs_pt.data(), 4 * fobs_pt.size());
    H2D(h->fobs_uv, fobs_uv.data(), 8 * fobs_uv.size());
    if (h->jr_own_runs) {
        H2D(h->jd_nf, jd_nf.data(), 4 * jd_nf.size());
        H2D(h->jd_frames, jd_frames.data(), 4 * jd_frames.size());
        H2D(h->jd_mask, jd_mask.data(), 4 * jd_mask.size());
    }
    if (h->det_active) {
        H2D(h->dj_ptr, dj_ptr.data(), 4 * dj_ptr.size());
        H2D(h->dj_ent, dj_ent.data(), 4 * dj_ent.size());
        H2D(h->ds_pair_ptr, ds_pair_ptr.data(), 4 * ds_pair_ptr.size());
        H2D(h->ds_pair_fa, ds_pair_fa.data(), 4 * ds_pair_fa.size());
        H2D(h->ds_pair_fb, ds_pair_fb.data(), 4 * ds_pair_fb.size());
        H2D(h->ds_pair_ent, ds_pair_ent.data(), 4 * ds_pair_ent.size());
        H2D(h->ds_f_ptr, ds_f_ptr.data(), 4 * ds_f_ptr.size());
        H2D(h->ds_f_ent, ds_f_ent.data(), 4 * ds_f_ent.size());
    }
    H2D(h->grp_first, grp_first.data(), 4 * grp_first.size());
    H2D(h->grp_count, grp_count.data(), 4 * grp_count.size());
    H2D(h->grp_nf, grp_nf.data(), 4 * grp_nf.size());
    H2D(h->grp_frames, grp_frames.data(), 4 * grp_frames.size());
    H2D(h->obs_slot, obs_slot.data(), obs_slot.size());
    H2D(h->pt_mask, pt_mask.data(), 4 * pt_mask.size());
    H2D(h->gen_list, gen_list.data(), 4 * gen_list.size());
    H2D(h->lg_item, lg_item.data(), 4 * lg_item.size());
    H2D(h->lg_np, lg_np.data(), 4 * lg_np.size());
    H2D(h->lg_nf, lg_nf.data(), 4 * lg_nf.size());
    H2D(h->lg_pts, lg_pts.data(), 4 * lg_pts.size());
    H2D(h->lg_frames, lg_frames.data(), 4 * lg_frames.size());
    H2D(h->lg_obs_off, lg_obs_off.data(), 8 * lg_obs_off.size());
    H2D(h->lg_obs, lg_obs.data(), 4 * lg_obs.size());
    H2D(h->wg_jmin, wg_jmin.data(), 4 * wg_jmin.size());
    if (h->jac_runs) {
        H2D(h->jr_first, jr_first.data(), 4 * jr_first.size());
        H2D(h->jr_count, jr_count.data(), 4 * jr_count.size());
        H2D(h->jr_jmin, jr_jmin.data(), 4 * jr_jmin.size());
        if (h->jac_runs_masked) H2D(h->jr_group, jr_group.data(), 4 * jr_group.size());
    }
#undef H2D
    for (auto& a : h->att) {
        if (!a.allocated) continue;
        HIPCHK(h, hipMemsetAsync(a.dc.p, 0, 8 * d.ld, s));
        HIPCHK(h, hipMemsetAsync(a.dx.p, 0, 24 * N > 0 ? 24 * N : 8, s));
    }
    h->cur = 0;
    for (int sl = 0; sl < SRK_SLOTS; ++sl) h->att[sl].trial = sl + 1;
    rc = compute_cam_packs(h, 0);
    if (rc != SRK_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(s)); // host staging vectors go out of scope
    // covisibility of THIS shard; with several ranks the caller must supply the global one
    // (srk_ba_set_covisibility) -- until then the skyline is the full lower triangle
    h->min_cv.assign((size_t)M, 0);
    if (!h->allreduce && !h->comm) {
        for (int32_t j = 0; j < M; ++j) h->min_cv[(size_t)j] = j;
        for (int64_t i = 0; i < N; ++i) {
            if (row_ptr[i + 1] == row_ptr[i]) continue;
            int32_t first = obs_frame[row_ptr[i]];
            for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o)
                h->min_cv[(size_t)obs_frame[o]] = std::min(h->min_cv[(size_t)obs_frame[o]], first);
        }
    }
    stage("host-to-device copies");
    rc = build_envelope(h);
    if (rc != SRK_OK) return rc;
    stage("skyline, solver plans");
    h->have_scene = true;
    return SRK_OK;
}

// entry of an upload / optimise call: the fused outer steps again after an earlier call's hand-off timeout (see chol_fused_wanted).
// Every rank of a sharded run counts the same timeouts (the status words are exchanged), so all re-arm together.
static void rearm_fusion(srk_ba* h)
{
    if (!h->chol_fused_wanted || h->chol_fused || h->fusion_rearms_left <= 0) return;
    --h->fusion_rearms_left;
    h->chol_fused = true;
    for (auto& a : h->att) a.sync.fused = true;
}

// after a failed solve: every slot's system and zero-initialised plan buffers back to zeros (see srk_ba::poisoned)
static int clear_poison(srk_ba* h)
{
    if (!h->poisoned) return SRK_OK;
    const SrkDims& d = h->d;
    for (auto& a : h->att) {
        if (!a.allocated) continue;
        HIPCHK(h, hipStreamSynchronize(a.stream));
        HIPCHK(h, hipMemsetAsync(a.S.p, 0, (size_t)(8 * d.ld * d.ld), h->main_stream));
        HIPCHK(h, hipMemsetAsync(a.rhs.p, 0, (size_t)(8 * d.ld), h->main_stream));
        HIPCHK(h, hipMemsetAsync(a.wy.p, 0, (size_t)(16 * d.ld), h->main_stream));
        HIPCHK(h, hipMemsetAsync(a.dc.p, 0, (size_t)(8 * d.ld), h->main_stream));
        HIPCHK(h, hipMemsetAsync(a.dx.p, 0, d.N > 0 ? (size_t)(24 * d.N) : 8, h->main_stream));
        HIPCHK(h, hipMemsetAsync(a.info.p, 0, 4, h->main_stream));
        HIPCHK(h, hipMemsetAsync(a.acc.p, 0, (size_t)(8 * 3 * d.Ns + 64), h->main_stream));
        for (size_t i = 0; i < a.plan_bufs.size(); ++i)
            if (a.plan_zeroed[i]) HIPCHK(h, hipMemsetAsync(a.plan_bufs[i].p, 0, a.plan_bufs[i].bytes, h->main_stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->main_stream));
    h->poisoned = false;
    return SRK_OK;
}

// a per-frame array of `w` doubles per frame, just downloaded in the internal frame order -> the caller's order
static void frames_to_user(const srk_ba* h, double* a, int w)
{
    if (h->frame_int.empty()) return;
    const int32_t M = h->d.M;
    std::vector<double> tmp(a, a + (size_t)w * (size_t)M);
    for (int32_t j = 0; j < M; ++j) std::memcpy(a + (size_t)w * (size_t)j, &tmp[(size_t)w * (size_t)h->frame_int[(size_t)j]], (size_t)(8 * w));
}

extern "C" int srk_ba_reset_scene(srk_ba* h)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    select_attempt(h, 0);
    hipStream_t s = h->stream;
    for (int sl = 1; sl < SRK_SLOTS; ++sl) HIPCHK(h, hipStreamSynchronize(h->att[sl].stream)); // a speculative attempt may still read the current scene
    {
        int rcp = clear_poison(h);
        if (rcp != SRK_OK) return rcp;
    }
    h->cur = 0;
    for (int sl = 0; sl < SRK_SLOTS; ++sl) h->att[sl].trial = sl + 1;
    if (h->d.N > 0) HIPCHK(h, hipMemcpyAsync(h->pts[0].p, h->pts0.p, 24 * h->d.N, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->camR[0].p, h->camR0.p, 72 * (int64_t)h->d.M, hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipMemcpyAsync(h->camT[0].p, h->camT0.p, 24 * (int64_t)h->d.M, hipMemcpyDeviceToDevice, s));
    return compute_cam_packs(h, 0);
}

extern "C" int srk_ba_download_scene(srk_ba* h, double* pts, double* cam_R, double* cam_T, int revert)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int c = h->cur;
    std::vector<double> tmp((size_t)(3 * h->d.N));
    if (h->d.N > 0) HIPCHK(h, hipMemcpyAsync(tmp.data(), h->pts[c].p, 24 * h->d.N, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(cam_R, h->camR[c].p, 72 * (int64_t)h->d.M, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(cam_T, h->camT[c].p, 24 * (int64_t)h->d.M, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    frames_to_user(h, cam_R, 9);
    frames_to_user(h, cam_T, 3);
    for (int64_t i = 0; i < h->d.N; ++i) std::memcpy(pts + 3 * h->perm[(size_t)i], &tmp[(size_t)(3 * i)], 24);
    if (revert && h->normalized_on_upload) srk_ba_revert_normalization(h->d.N, pts, h->d.M, cam_R, cam_T, &h->nrm); // :706
    return SRK_OK;
}

// ------------------------------------------------------------------ phases

static int exchange(srk_ba* h, double* dev_ptr, int64_t count)
{
    if (h->comm) { // RCCL on this attempt's stream: ordered behind the kernels that filled the buffer, nothing waits on the host
        ncclComm_t comm = (h->A == &h->att[1] && h->comm2) ? h->comm2 : h->comm; // the second slot's own communicator
        ncclResult_t r = rccl().AllReduce(dev_ptr, dev_ptr, (size_t)count, ncclDouble, ncclSum, comm, h->stream);
        if (r != ncclSuccess) {
            h->last_error = std::string("ncclAllReduce: ") + rccl().GetErrorString(r);
            return SRK_E_DEVICE;
        }
        return SRK_OK;
    }
    if (!h->allreduce) return SRK_OK;
    // the hook reduces on its own (RCCL) stream: everything queued on ours must have landed first, and the hook
    // returns only after the reduced values are visible
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int rc = h->allreduce(h->allreduce_ctx, dev_ptr, count);
    if (rc != 0) {
        h->last_error = "allreduce hook failed";
        return SRK_E_DEVICE;
    }
    return SRK_OK;
}

// ---- collectives of the damping-parallel schedule (world >= 2).  A group = the same operation for the slots k = 0 .. G-1
// (root of slot k: k % world), issued by every rank in the same program order.  Natively: ONE communicator on ONE stream
// (comm_stream) -- the group waits for each slot's producer stream, runs as one ncclGroup (the rooted operations of
// different roots share the links), and each slot's stream waits for it; nothing waits on the host.  With the callback
// (gloo rehearsals, a caller's own transport) only a sum is available: a reduce is an all-reduce whose result the other
// ranks ignore, a broadcast an all-reduce of a buffer the other ranks zeroed; the callback blocks the host.
enum { SRK_COLL_ALLREDUCE = 0, SRK_COLL_REDUCE = 1, SRK_COLL_BCAST = 2 };
#define SRK_RETRY_ALLREDUCE 1000 // (internal) the damping-parallel round failed its self-check: the all-reduce schedule from here on
#ifdef SRK_DEV
static int g_dp_corrupt = 0; // test hook: the next self-check finds a mismatch at stage 1 (reduce) / 2 (broadcast)
extern "C" void srk_dbg_dp_corrupt(int stage) { g_dp_corrupt = stage; }
#endif
// {sum, sum of magnitudes} of a device buffer in a fixed order, to the host (blocking: first round of a handle only)
static int dp_checksum(srk_ba* h, hipStream_t st, const double* p, int64_t n, double out[2])
{
    int rc = dev_alloc(h, h->dp_chk, 8 * (512 + 2 + 16));
    if (rc != SRK_OK) return rc;
    srk_launch_checksum(st, p, n, P<double>(h->dp_chk), P<double>(h->dp_chk) + 512);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, P<double>(h->dp_chk) + 512, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return SRK_OK;
}
// sum over the ranks of n <= 16 host doubles through the communicator's plain all-reduce (blocking)
static int dp_allreduce_host(srk_ba* h, double* v, int n)
{
    int rc = dev_alloc(h, h->dp_chk, 8 * (512 + 2 + 16));
    if (rc != SRK_OK) return rc;
    double* stage = P<double>(h->dp_chk) + 514;
    HIPCHK(h, hipMemcpyAsync(stage, v, (size_t)(8 * n), hipMemcpyHostToDevice, h->comm_stream));
    ncclResult_t r = rccl().AllReduce(stage, stage, (size_t)n, ncclDouble, ncclSum, h->comm, h->comm_stream);
    if (r != ncclSuccess) { h->last_error = std::string("ncclAllReduce (self-check): ") + rccl().GetErrorString(r); return SRK_E_DEVICE; }
    HIPCHK(h, hipMemcpyAsync(v, stage, (size_t)(8 * n), hipMemcpyDeviceToHost, h->comm_stream));
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    return SRK_OK;
}
static int coll_group(srk_ba* h, int op, int G, double* const* ptrs, const int64_t* counts)
{
    if (h->comm) {
        for (int k = 0; k < G; ++k) {
            HIPCHK(h, hipEventRecord(h->att[k].ev_a, h->att[k].stream));
            HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->att[k].ev_a, 0));
        }
        const bool rooted = rccl().rooted && op != SRK_COLL_ALLREDUCE;
        if (!rooted && op == SRK_COLL_BCAST)
            for (int k = 0; k < G; ++k)
                if (h->rank != k % h->world) HIPCHK(h, hipMemsetAsync(ptrs[k], 0, (size_t)(8 * counts[k]), h->comm_stream));
        ncclResult_t r = ncclSuccess;
        if (rooted) r = rccl().GroupStart();
        for (int k = 0; k < G && r == ncclSuccess; ++k) {
            const int root = k % h->world;
            if (rooted && op == SRK_COLL_REDUCE)
                r = rccl().Reduce(ptrs[k], ptrs[k], (size_t)counts[k], ncclDouble, ncclSum, root, h->comm, h->comm_stream);
            else if (rooted && op == SRK_COLL_BCAST)
                r = rccl().Broadcast(ptrs[k], ptrs[k], (size_t)counts[k], ncclDouble, root, h->comm, h->comm_stream);
            else
                r = rccl().AllReduce(ptrs[k], ptrs[k], (size_t)counts[k], ncclDouble, ncclSum, h->comm, h->comm_stream);
        }
        if (rooted) {
            ncclResult_t r2 = rccl().GroupEnd();
            if (r == ncclSuccess) r = r2;
        }
        if (r != ncclSuccess) {
            h->last_error = std::string("RCCL collective: ") + rccl().GetErrorString(r);
            return SRK_E_DEVICE;
        }
        HIPCHK(h, hipEventRecord(h->ev_comm, h->comm_stream));
        for (int k = 0; k < G; ++k) HIPCHK(h, hipStreamWaitEvent(h->att[k].stream, h->ev_comm, 0));
        return SRK_OK;
    }
    if (!h->allreduce) return SRK_OK;
    for (int k = 0; k < G; ++k) {
        hipStream_t st = h->att[k].stream;
        if (op == SRK_COLL_BCAST && h->rank != k % h->world) HIPCHK(h, hipMemsetAsync(ptrs[k], 0, (size_t)(8 * counts[k]), st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (h->allreduce(h->allreduce_ctx, ptrs[k], counts[k]) != 0) {
            h->last_error = "allreduce hook failed";
            return SRK_E_DEVICE;
        }
    }
    return SRK_OK;
}

// with_status: {solver info, point-update finite flag (lives behind acc)} are packed next to the error scalar and
// summed over the ranks with it, so every rank takes the same accept / reject decision
static int phase_error(srk_ba* h, int which, double* err_host, bool with_status = false, bool no_exchange = false)
{
    const SrkDims& d = h->d;
    hipStream_t s = h->stream;
    int32_t np = srk_error_partials(d);
    srk_launch_error(s, d, P<double>(h->pts[which]), P<double>(h->cam[which]), P<int32_t>(h->obs_frame),
                     P<int32_t>(h->obs_pt), P<double>(h->obs_uv), P<double>(h->A->err_partial), np, h->A->err_dst,
                     h->jac_fused ? P<int32_t>(h->wg_jmin) : nullptr, with_status ? P<int>(h->A->info) : nullptr,
                     with_status ? reinterpret_cast<int*>(reinterpret_cast<char*>(h->A->acc.p) + 8 * 3 * d.Ns) : nullptr);
    HIPCHK(h, hipGetLastError());
    int rc = no_exchange ? SRK_OK : exchange(h, h->A->err_dst, with_status ? 3 : 1);
    if (rc != SRK_OK) return rc;
    if (err_host) {
        HIPCHK(h, hipMemcpyAsync(err_host, h->A->err_dst, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    return SRK_OK;
}

static int phase_derivatives(srk_ba* h)
{
    const SrkDims& d = h->d;
    hipStream_t s = h->stream;
    int c = h->cur;
    HIPCHK(h, hipMemsetAsync(h->Vg.p, 0, 8 * 9 * d.Ns, s));
    HIPCHK(h, hipMemsetAsync(h->Ug.p, 0, 8 * SRK_UG * (int64_t)d.M, s));
    if (h->profile_level >= 1) HIPCHK(h, hipEventRecord(h->ev[12], s));
    const SrkDetJac detj{ P<double>(h->dj_stage), P<int32_t>(h->dj_ptr), P<int32_t>(h->dj_ent) };
    if (h->jac_runs) {
        srk_launch_jac_runs(s, d, P<double>(h->pts[c]), P<double>(h->cam[c]), P<int64_t>(h->row_ptr), P<int32_t>(h->obs_frame),
                            P<double>(h->obs_uv), P<double>(h->W), P<double>(h->Vg), P<double>(h->Ug), P<int32_t>(h->jr_first),
                            P<int32_t>(h->jr_count), h->jr_tasks, P<int32_t>(h->jr_jmin),
                            h->jac_runs_masked ? P<int32_t>(h->jr_group) : nullptr,
                            h->jr_own_runs ? P<int32_t>(h->jd_nf) : P<int32_t>(h->grp_nf),
                            h->jr_own_runs ? P<int32_t>(h->jd_frames) : P<int32_t>(h->grp_frames),
                            h->jr_own_runs ? P<uint32_t>(h->jd_mask) : P<uint32_t>(h->pt_mask), h->det_active ? &detj : nullptr,
                            h->jr_own_runs ? 32 : SRK_GRP_MAXNF_HOST);
        if (h->profile_level >= 1) HIPCHK(h, hipEventRecord(h->ev[13], s));
    } else if (h->jac_fused) {
        srk_launch_jac_fused(s, d, P<double>(h->pts[c]), P<double>(h->cam[c]), P<int32_t>(h->obs_frame),
                             P<int32_t>(h->obs_pt), P<double>(h->obs_uv), P<double>(h->W), P<double>(h->Vg),
                             P<double>(h->Ug), P<int32_t>(h->wg_jmin));
        if (h->profile_level >= 1) HIPCHK(h, hipEventRecord(h->ev[13], s));
    } else {
        srk_launch_jac_points(s, d, P<double>(h->pts[c]), P<double>(h->cam[c]), P<int32_t>(h->obs_frame),
                              P<int32_t>(h->obs_pt), P<double>(h->obs_uv), P<double>(h->W), P<double>(h->Vg));
        if (h->profile_level >= 1) HIPCHK(h, hipEventRecord(h->ev[13], s));
        srk_launch_jac_frames(s, d, h->max_frame_obs, P<double>(h->pts[c]), P<double>(h->cam[c]),
                              P<int64_t>(h->col_ptr), P<int32_t>(h->fobs_pt), P<double>(h->fobs_uv), P<double>(h->Ug));
    }
    HIPCHK(h, hipGetLastError());
    // landmark shards: Ug stays this rank's partial sum; it enters the reduced camera system before that is summed
    return SRK_OK;
}

static int phase_schur(srk_ba* h, double c, bool local_only = false);
// the reduced camera system of this rank's landmarks, packed for an exchange: band + right-hand side behind it
static int schur_pack(srk_ba* h)
{
    const SrkDims& d = h->d;
    hipStream_t s = h->stream;
    int rc;
    if ((rc = dev_alloc(h, h->A->packed, (size_t)(8 * (h->band_packed + d.ld)))) != SRK_OK) return rc;
    double* tail = P<double>(h->A->packed) + h->band_packed;
    srk_launch_band_pack(s, d.ld, P<int64_t>(h->band_col), P<int64_t>(h->band_off), P<double>(h->A->S), P<double>(h->A->packed), 0);
    HIPCHK(h, hipMemcpyAsync(tail, h->A->rhs.p, (size_t)(8 * d.ld), hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipGetLastError());
    return SRK_OK;
}
static int schur_unpack(srk_ba* h)
{
    const SrkDims& d = h->d;
    hipStream_t s = h->stream;
    double* tail = P<double>(h->A->packed) + h->band_packed;
    srk_launch_band_pack(s, d.ld, P<int64_t>(h->band_col), P<int64_t>(h->band_off), P<double>(h->A->S), P<double>(h->A->packed), 1);
    HIPCHK(h, hipMemcpyAsync(h->A->rhs.p, tail, (size_t)(8 * d.ld), hipMemcpyDeviceToDevice, s));
    HIPCHK(h, hipGetLastError());
    return SRK_OK;
}
static int phase_schur(srk_ba* h, double c, bool local_only)
{
    const SrkDims& d = h->d;
    hipStream_t s = h->stream;
    srk_launch_env_zero(s, d.ld, P<int64_t>(h->env_col), P<double>(h->A->S), P<double>(h->A->rhs), P<int32_t>(h->A->irr)); // S band, rhs, hand-back counter
    const SrkDetSchur dets{ P<double>(h->A->det_stage), P<double>(h->A->det_rhs), P<int32_t>(h->ds_pair_ptr), P<int32_t>(h->ds_pair_fa),
                            P<int32_t>(h->ds_pair_fb), P<int32_t>(h->ds_pair_ent), h->ds_n_pairs, P<int32_t>(h->ds_f_ptr), P<int32_t>(h->ds_f_ent) };
    srk_launch_schur_grouped(s, d, c, P<int64_t>(h->row_ptr), P<int32_t>(h->obs_pt), P<uint8_t>(h->obs_slot),
                             P<uint32_t>(h->pt_mask), P<double>(h->W), P<double>(h->Vg), P<double>(h->A->S),
                             P<double>(h->A->rhs), P<int32_t>(h->grp_first), P<int32_t>(h->grp_count), P<int32_t>(h->grp_nf),
                             P<int32_t>(h->grp_frames), h->n_groups, h->n_groups_wide, h->n_groups_mid, h->schur_fp32 ? 1 : 0,
                             P<int32_t>(h->A->irr), h->n_mm_uniform, h->n_mm_ragged, h->det_active ? &dets : nullptr);
    srk_launch_schur_long(s, d, c, P<double>(h->W), P<double>(h->Vg), P<double>(h->A->S), P<double>(h->A->rhs),
                          P<int32_t>(h->lg_item), h->n_long_items, P<int32_t>(h->lg_np), P<int32_t>(h->lg_nf), P<int32_t>(h->lg_pts),
                          P<int32_t>(h->lg_frames), P<int64_t>(h->lg_obs_off), P<int32_t>(h->lg_obs), h->long_fb);
    srk_launch_schur(s, d, c, P<int64_t>(h->row_ptr), P<int32_t>(h->obs_frame), P<double>(h->W), P<double>(h->Vg),
                     P<double>(h->A->S), P<double>(h->A->rhs), P<int32_t>(h->gen_list), h->n_generic);
    HIPCHK(h, hipGetLastError());
    // G (frame blocks, damped) and the frame gradients are linear in this rank's landmarks as well, so they are added
    // before the exchange; the identity diagonal of fixed / padding variables comes from rank 0 alone
    srk_launch_assemble(s, d, c, P<double>(h->Ug), P<double>(h->A->S), P<double>(h->A->rhs), h->rank == 0 ? 1.0 : 0.0,
                        P<int64_t>(h->row_ptr), P<int32_t>(h->obs_frame), P<double>(h->W), P<double>(h->Vg), P<int32_t>(h->A->irr));
    HIPCHK(h, hipGetLastError());
    h->last_hessian_factor = c;
    if ((h->allreduce || h->comm) && !local_only) { // landmark shards: ONE exchange per attempt; only the band travels, the rhs rides behind it
        int rc = schur_pack(h);
        if (rc == SRK_OK) rc = exchange(h, P<double>(h->A->packed), h->band_packed + d.ld);
        if (rc == SRK_OK) rc = schur_unpack(h);
        if (rc != SRK_OK) return rc;
    }
    return SRK_OK;
}

// one solve of the reduced camera system in the current mode; prof may be NULL
static void launch_solve(srk_ba* h, SrkSolveProf* prof)
{
    const SrkDims& d = h->d;
    if (h->A->plan.P >= 2)
        srk_chol_solve_chunked(h->stream, h->A->plan, d.ld, P<double>(h->A->S), P<double>(h->A->rhs), P<double>(h->A->dc),
                               P<int64_t>(h->env_col), P<int>(h->A->info), prof, &h->A->sync);
    else
        srk_chol_solve(h->stream, d.ld, P<double>(h->A->S), P<double>(h->A->rhs), P<double>(h->A->wy), P<double>(h->A->dc),
                       P<int>(h->A->info), h->row_end_h.data(), h->col_begin_h.data(), P<double>(h->A->dinv), prof, &h->A->sync, 10 * (int64_t)d.M);
}

static int phase_solve(srk_ba* h, bool profile)
{
    const SrkDims& d = h->d;
    // inside the LM loop the status words and the point accumulators are left cleared by the kernels that consume them
    // (k_error_final, k_point_update); the step-wise entry points clear them here
    if (!h->lean_resets) HIPCHK(h, hipMemsetAsync(h->A->info.p, 0, 4, h->stream));
    h->A->solve_prof = SrkSolveProf{};
    if (profile) {
        size_t need = (size_t)(2 * (2 * (d.ld / SRK_CHOL_NB) + 64)); // every level of a nested plan included
        while (h->chol_ev.size() < need) {
            hipEvent_t e;
            HIPCHK(h, hipEventCreate(&e));
            h->chol_ev.push_back(e);
        }
        h->A->solve_prof.ev = h->chol_ev.data();
        h->A->solve_prof.cap = h->chol_ev.size();
    }
    launch_solve(h, profile ? &h->A->solve_prof : nullptr);
    HIPCHK(h, hipGetLastError());
    return SRK_OK;
}

static int read_info(srk_ba* h, int* info_host)
{
    HIPCHK(h, hipMemcpyAsync(info_host, h->A->info.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SRK_OK;
}

static int phase_backsub_apply(srk_ba* h, double c)
{
    const SrkDims& d = h->d;
    hipStream_t s = h->stream;
    int cur = h->cur, tr = h->A->trial;
    if (!h->lean_resets) HIPCHK(h, hipMemsetAsync(h->A->acc.p, 0, 8 * 3 * d.Ns + 64, s));
    srk_launch_backsub(s, d, c, P<int32_t>(h->obs_frame), P<int32_t>(h->obs_pt), P<double>(h->W), P<double>(h->Vg),
                       P<double>(h->A->dc), P<double>(h->A->acc), P<double>(h->pts[cur]), P<double>(h->pts[tr]),
                       P<double>(h->A->dx));
    HIPCHK(h, hipGetLastError());
    return SRK_OK;
}

static int phase_cam_apply(srk_ba* h)
{
    int cur = h->cur, tr = h->A->trial;
    srk_launch_cam_apply(h->stream, h->d.M, P<double>(h->camR[cur]), P<double>(h->camT[cur]), P<double>(h->A->dc),
                         P<double>(h->camR[tr]), P<double>(h->camT[tr]), P<double>(h->K), h->f0, P<double>(h->cam[tr]));
    HIPCHK(h, hipGetLastError());
    return SRK_OK;
}

extern "C" {

int srk_ba_phase_error(srk_ba* h, double* err, int64_t* seen)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    if (seen) *seen = h->d.O;
    return phase_error(h, h->cur, err);
}
int srk_ba_phase_derivatives(srk_ba* h)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = phase_derivatives(h);
    if (rc != SRK_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SRK_OK;
}
int srk_ba_phase_schur(srk_ba* h, double c)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    select_attempt(h, 0); // the staged calls always work on attempt slot 0
    h->last_slot = 0;
    int rc = clear_poison(h);
    if (rc != SRK_OK) return rc;
    rc = phase_schur(h, c);
    if (rc != SRK_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SRK_OK;
}
int srk_ba_phase_solve(srk_ba* h)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = phase_solve(h, false);
    if (rc != SRK_OK) return rc;
    int info = 0;
    rc = read_info(h, &info);
    if (rc != SRK_OK) return rc;
    if (info && srk_debug()) fprintf(stderr, "srk_ba_phase_solve: info=%d (1 = pivot, 4 = non-finite solution, 8 = hand-off timeout of the fused solve)\n", info);
    if ((info & 8) && h->A->plan.P >= 2) {
        // a hand-off of the fused outer step timed out: a scheduling event, not a numerical failure.  The chunked solve works
        // on copies (the system itself is intact): the plan's buffers back to zero, fusion off for good, once more unfused.
        ++h->sync_timeouts;
        h->chol_fused = false;
        for (auto& a : h->att) a.sync.fused = false;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < h->A->plan_bufs.size(); ++i)
            if (h->A->plan_zeroed[i]) HIPCHK(h, hipMemsetAsync(h->A->plan_bufs[i].p, 0, h->A->plan_bufs[i].bytes, h->stream));
        HIPCHK(h, hipMemsetAsync(h->A->info.p, 0, 4, h->stream));
        rc = phase_solve(h, false);
        if (rc != SRK_OK) return rc;
        rc = read_info(h, &info);
        if (rc != SRK_OK) return rc;
    } else if (info & 8) {
        // (the single-chain / dense solve factorises the system in place: nothing to repeat from; the caller builds it again)
        ++h->sync_timeouts;
        h->chol_fused = false;
        for (auto& a : h->att) a.sync.fused = false;
        h->last_error = "a hand-off of the fused solve timed out; the system was factorised in place: call srk_ba_phase_schur again "
                        "(the unfused launch sequence is selected from now on)";
    }
    if (info) h->poisoned = true;
    return info ? 1 : 0;
}
int srk_ba_phase_backsub(srk_ba* h, double c)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = phase_backsub_apply(h, c);
    if (rc != SRK_OK) return rc;
    rc = phase_cam_apply(h);
    if (rc != SRK_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SRK_OK;
}
int srk_ba_phase_accept(srk_ba* h)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    std::swap(h->cur, h->att[0].trial); // slot 0's trial scene becomes current
    return SRK_OK;
}

// ------------------------------------------------------------------ the LM loop (bundle-adj-kanatani.cpp:720-893)

int srk_ba_optimize(srk_ba* h, const double* allowed_err_change, const double* max_hessian_factor,
                    int64_t max_iterations, srk_ba_report* rep)
{
    srk_ba_report local;
    if (!rep) rep = &local;
    std::memset(rep, 0, sizeof *rep);
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const SrkDims& d = h->d;
    auto t_begin = std::chrono::steady_clock::now();
    h->iter_log.clear();
    rep->world_scale = h->nrm.world_scale;
    rearm_fusion(h);
    {
        int rcp = clear_poison(h);
        if (rcp != SRK_OK) return rcp;
    }

    auto fail_device = [&](int rc) {
        rep->status = SRK_STATUS_DEVICE_ERROR;
        rep->optimized = 0;
        return rc;
    };
    auto ev_ms = [&](int a, int b) {
        float ms = 0;
        if (h->profile_level < 1 || hipEventElapsedTime(&ms, h->ev[a], h->ev[b]) != hipSuccess) return 0.0;
        return (double)ms;
    };
#define EVREC(i) do { if (h->profile_level >= 1) HIPCHK(h, hipEventRecord(h->ev[i], s)); } while (0)

    // clear the status words and point accumulators of every slot once; from here on the kernels keep them clear
    for (auto& a : h->att)
        if (a.allocated) {
            HIPCHK(h, hipMemsetAsync(a.info.p, 0, 4, s));
            HIPCHK(h, hipMemsetAsync(a.acc.p, 0, 8 * 3 * d.Ns + 64, s));
        }
    HIPCHK(h, hipStreamSynchronize(s)); // the second slot's stream starts after this
    struct LeanGuard {
        srk_ba* h;
        explicit LeanGuard(srk_ba* hh) : h(hh) { h->lean_resets = true; }
        ~LeanGuard() { h->lean_resets = false; }
    } lean_guard(h);
    double hessian_factor = (double)0.0001f; // :723 (float literal)
    // seen_points_count over all shards (:483, :726)
    // once per uploaded scene: it does not change between optimise calls
    if (h->seen_global < 0) {
        double seen_d = (double)d.O;
        if (h->allreduce || h->comm) {
            HIPCHK(h, hipMemcpyAsync(h->A->err_dst, &seen_d, 8, hipMemcpyHostToDevice, s));
            int rc = exchange(h, h->A->err_dst, 1);
            if (rc != SRK_OK) return fail_device(rc);
            HIPCHK(h, hipMemcpyAsync(&seen_d, h->A->err_dst, 8, hipMemcpyDeviceToHost, s));
            HIPCHK(h, hipStreamSynchronize(s));
        }
        h->seen_global = (int64_t)seen_d;
    }
    const double seen_d = (double)h->seen_global;
    rep->seen = (int64_t)seen_d;

    double err_initial = 0;
    EVREC(0);
    int rc = phase_error(h, h->cur, nullptr);
    if (rc != SRK_OK) return fail_device(rc);
    EVREC(1);
    HIPCHK(h, hipMemcpyAsync(&err_initial, h->A->err_dst, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    rep->ms_error += ev_ms(0, 1);
    rep->err_initial = rep->err_final = err_initial;

    bool result_true = false;
    bool done = false;
    if (allowed_err_change && err_initial < *allowed_err_change) { // :749-753
        rep->status = SRK_STATUS_ABS_ERR_THRESHOLD;
        result_true = true;
        done = true;
    }
    double err_value = err_initial;
    bool spec_wanted = false, spec_drain = false;
    int64_t prev_attempts = 0; // attempts the previous iteration needed
    while (!done) {
        if (max_iterations > 0 && rep->iterations >= max_iterations) {
            rep->status = SRK_STATUS_MAX_ITERATIONS;
            result_true = false;
            break;
        }
        // ComputeCloseFormReprErrorDerivatives (:759)
        EVREC(0);
        rc = phase_derivatives(h);
        if (rc != SRK_OK) return fail_device(rc);
        EVREC(1);
        rep->jacobian_launches += 2;
        bool jac_timed = false;

        HIPCHK(h, hipEventRecord(h->ev_jac, s));
        const auto t_iter = std::chrono::steady_clock::now(); // SRK_DEBUG trace only

        // try_decrease_targ_fun (:764-852): the backup is the untouched `cur` buffer set, every attempt slot has a trial
        // set of its own.  With two slots the NEXT damping factor (x10) is tried speculatively beside the current one on
        // a second stream: the solve is a latency chain that leaves most of the chip idle, so the pair costs little more
        // than one attempt, and a rejected first attempt finds its successor already done.  Attempts are still judged
        // strictly in the reference's order; a speculative result that is not needed is dropped unseen.
        bool have_prev = false;
        double err_new_prev = 0, err_new = std::nan("");
        int decrease = 0; // 1 success, 2 hessian overflow, 3 converged
        int accepted_slot = 0;
        // one attempt, enqueued on slot sl's stream without waiting for it, in two parts so that a pair can put both
        // Schur sums (which fill the chip one after the other) in front of both solves
        auto enqueue_schur = [&](int sl, double c) -> int {
            select_attempt(h, sl);
            hipStream_t st = h->stream;
            int r2 = SRK_OK;
            if (sl >= 1 && hipStreamWaitEvent(st, h->ev_jac, 0) != hipSuccess) r2 = SRK_E_DEVICE;
            if (sl == 0) EVREC(2);
            if (r2 == SRK_OK) r2 = phase_schur(h, c);
            if (sl == 0) EVREC(3);
            select_attempt(h, 0);
            return r2;
        };
        auto enqueue_rest = [&](int sl, double c) -> int {
            select_attempt(h, sl);
            hipStream_t st = h->stream;
            int r2 = phase_solve(h, h->profile_level >= 2);
            if (sl == 0) EVREC(4);
            if (r2 == SRK_OK) r2 = phase_backsub_apply(h, c);
            if (sl == 0) EVREC(5);
            if (r2 == SRK_OK) r2 = phase_cam_apply(h);
            if (sl == 0) EVREC(6);
            if (r2 == SRK_OK) r2 = phase_error(h, h->A->trial, nullptr, true);
            if (sl == 0) EVREC(7);
            // one read-back per attempt into pinned host memory: {error, solver info, point-update info}
            if (r2 == SRK_OK && hipMemcpyAsync(h->A->host_back, h->A->err_dst, 24, hipMemcpyDeviceToHost, st) != hipSuccess)
                r2 = SRK_E_DEVICE;
            if (r2 == SRK_OK && hipEventRecord(h->A->done, st) != hipSuccess) r2 = SRK_E_DEVICE;
            select_attempt(h, 0);
            return r2;
        };
        // several ranks, damping-parallel schedule (instrumentation off): see dp_round below
        bool dp_mode = (h->allreduce || h->comm) && (h->world >= 2 || h->dp_force) && h->dp_schedule &&
                       h->profile_level == 0 && h->att[1].allocated;
        // wait for slot sl's attempt and judge it exactly as the reference judges the attempt with factor `hessian_factor`
        std::function<int(int, double)> redo_unfused; // (defined below: repeats one attempt after a hand-off timeout)
        auto judge_attempt = [&](int sl) -> int {
            if (hipStreamSynchronize(h->att[sl].stream) != hipSuccess) return SRK_E_DEVICE;
            const double* hb = h->att[sl].host_back;
            // (several ranks: the status words are SUMMED over the ranks, so bit 8 cannot be told from two ranks' bit 4; any
            // non-zero status of a fused solve is then taken for a possible timeout -- every rank sees the same sum and
            // repeats, a genuine failure shows again without the fused kernels)
            // (damping-parallel schedule: only the rank that solved a factor contributes its status word, bit 8 is exact, and
            // the round loop has dealt with it before anything is judged)
            const bool multi_rank = (h->allreduce || h->comm) && !dp_mode;
            if (multi_rank ? ((int)hb[1] != 0 && h->att[sl].sync.fused) : (((int)hb[1] & 8) != 0 && !dp_mode)) {
                // an in-launch hand-off of the fused solve timed out (srk_chol.hip: k_step256; its spins are bounded): the
                // numbers of this attempt are void.  From now on the unfused launch sequence (bit-identical arithmetic);
                // this attempt is repeated with the factor it stands for, which is `hessian_factor` at this point.
                int r2 = redo_unfused(sl, hessian_factor);
                if (r2 != SRK_OK) return r2;
                if (hipStreamSynchronize(h->att[sl].stream) != hipSuccess) return SRK_E_DEVICE;
            }
            struct { double err; int info; } back{ hb[0], (int)hb[1] };
            const int info2 = (int)hb[2];
            rep->attempts += 1;
            rep->schur_launches += 2;
            h->last_slot = sl;
            if (srk_debug())
                fprintf(stderr, "srk_ba[rank %d] iteration %lld attempt %lld (slot %d, +%.3f ms): hessian_factor %.3g err %.17g -> "
                                "%.17g, solver info %d, point-update info %d\n", h->rank, (long long)rep->iterations + 1,
                        (long long)rep->attempts, sl,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_iter).count(),
                        hessian_factor, err_value, back.err, back.info, info2);
            // Solve failed (:807-808, :1912-1913, :1953-1954).  The multiplicative damping keeps the diagonally scaled
            // system's smallest eigenvalue >= c, so a Cholesky pivot can only fail where diag(G) holds an exact
            // zero or a non-finite value -- exactly where the reference's Householder QR divides by a zero diagonal of R
            // and returns non-finite numbers: both sides end with "hessian overflow".
            if (back.info != 0 || info2 != 0) { h->poisoned = true; decrease = 2; return SRK_OK; }
            err_new = back.err;
            if (err_new - err_value < 0) { decrease = 1; accepted_slot = sl; return SRK_OK; } // :816-819
            // restore = drop the trial buffers (:823-826)
            if (have_prev && allowed_err_change) { // :828-838
                double change = err_new - err_new_prev;
                if (std::fabs(change) < *allowed_err_change) { decrease = 3; return SRK_OK; }
            }
            hessian_factor *= 10; // :841
            if (max_hessian_factor && hessian_factor > *max_hessian_factor) { decrease = 2; return SRK_OK; } // :843-847
            err_new_prev = err_new;
            have_prev = true;
            return SRK_OK;
        };
        redo_unfused = [&](int sl, double c) -> int {
            ++h->sync_timeouts;
            if (srk_debug()) fprintf(stderr, "srk_ba[rank %d]: hand-off timeout in the fused solve (slot %d); repeating unfused\n", h->rank, sl);
            h->chol_fused = false;
            for (auto& a : h->att) a.sync.fused = false;
            h->poisoned = true;
            int r2 = clear_poison(h); // waits for both slots' streams, re-zeroes systems, plans, status words, accumulators
            if (r2 == SRK_OK) r2 = enqueue_schur(sl, c);
            if (r2 == SRK_OK) r2 = enqueue_rest(sl, c);
            return r2;
        };
        // several ranks: every rank takes the same decisions, so the two slots' exchanges are issued in the same order
        // everywhere; the native path wants the second slot's own communicator (srk_ba_rccl_init_second), the callback
        // serialises the exchanges on the host
        const bool multi = h->allreduce || h->comm;
        const bool can_speculate = h->speculate && h->att[1].allocated && h->profile_level == 0 &&
                                   (!multi || (h->spec_multi && (h->allreduce || h->comm2)));
        unsigned spec_in_flight = 0; // slots whose speculative attempt has been enqueued and not judged
        int round = 0;
        const int64_t attempts_before = rep->attempts;
        // ---- several ranks: one round = the next G damping factors c, 10c, (100c), one attempt slot each.
        //   every rank: Schur sum of its landmarks for each factor (+ its share of the frame blocks), band packed;
        //   band k REDUCED to rank k % world (a reduce, not an all-reduce: half the traffic, and one ncclGroup for all k);
        //   rank k % world: unpack, factorise and solve factor k -- the G solves run at the same time on G GPUs;
        //   corrections of factor k BROADCAST from its rank (80 KB at 1000 frames);
        //   every rank: back-substitution, camera update and error of its shard for every factor;
        //   ONE all-reduce of the G x {error, solver status, point-update status}; every rank judges the attempts in the
        //   reference's order and takes the same decisions.  An iteration that needs <= G attempts costs about one solve.
        auto dp_round = [&](int G, const double* cf) -> int {
            double* ptrs[SRK_SLOTS];
            int64_t counts[SRK_SLOTS];
            int r2 = SRK_OK;
            for (int k = 0; k < G && r2 == SRK_OK; ++k) {
                select_attempt(h, k);
                if (k > 0 && hipStreamWaitEvent(h->stream, h->ev_jac, 0) != hipSuccess) r2 = SRK_E_DEVICE;
                if (r2 == SRK_OK) r2 = phase_schur(h, cf[k], true);
                if (r2 == SRK_OK) r2 = schur_pack(h);
                ptrs[k] = P<double>(h->A->packed);
                counts[k] = h->band_packed + d.ld;
            }
            select_attempt(h, 0);
            // first native round of this handle: every rooted collective is checked against a plain all-reduce of checksums
            const bool verify = h->comm != nullptr && !h->dp_verified;
            double expect[2 * SRK_SLOTS] = {}, bad = 0;
            auto selfcheck_failed = [&](const char* what) {
                h->dp_schedule = false;
                h->dp_selfcheck_failed = true;
                h->last_error = std::string("damping-parallel schedule: self-check of the first native round failed (") + what +
                                "); this handle runs the all-reduce schedule";
                if (srk_debug()) fprintf(stderr, "srk_ba[rank %d]: %s\n", h->rank, h->last_error.c_str());
                return SRK_RETRY_ALLREDUCE;
            };
            if (verify && r2 == SRK_OK) { // what the sum over the ranks of band k must be
                for (int k = 0; k < G && r2 == SRK_OK; ++k) r2 = dp_checksum(h, h->att[k].stream, ptrs[k], counts[k], expect + 2 * k);
                if (r2 == SRK_OK) r2 = dp_allreduce_host(h, expect, 2 * G);
            }
            if (r2 == SRK_OK) r2 = coll_group(h, SRK_COLL_REDUCE, G, ptrs, counts);
            if (verify) {
                if (r2 == SRK_E_DEVICE) return selfcheck_failed("an RCCL call of the reduce group returned an error");
                for (int k = 0; k < G && r2 == SRK_OK; ++k) {
                    if (h->rank != k % h->world) continue;
                    double got[2];
                    r2 = dp_checksum(h, h->att[k].stream, ptrs[k], counts[k], got); // (the slot's stream waits for the group)
                    if (!(std::fabs(got[0] - expect[2 * k]) <= 1e-9 * expect[2 * k + 1] + 1e-300)) bad += 1;
                }
#ifdef SRK_DEV
                if (g_dp_corrupt == 1) bad += 1, g_dp_corrupt = 0;
#endif
                if (r2 == SRK_OK) r2 = dp_allreduce_host(h, &bad, 1);
                if (r2 == SRK_OK && bad > 0) return selfcheck_failed("a reduced band does not sum to the all-reduced checksum");
            }
            for (int k = 0; k < G && r2 == SRK_OK; ++k) {
                select_attempt(h, k);
                if (h->rank == k % h->world) {
                    r2 = schur_unpack(h);
                    if (r2 == SRK_OK) r2 = phase_solve(h, false);
                }
                ptrs[k] = P<double>(h->A->dc);
                counts[k] = d.ld;
            }
            select_attempt(h, 0);
            double rootv[2 * SRK_SLOTS] = {};
            if (verify && r2 == SRK_OK) { // the checksum of the corrections on the rank that solved them, known to everybody
                for (int k = 0; k < G && r2 == SRK_OK; ++k)
                    if (h->rank == k % h->world) r2 = dp_checksum(h, h->att[k].stream, ptrs[k], counts[k], rootv + 2 * k);
                if (r2 == SRK_OK) r2 = dp_allreduce_host(h, rootv, 2 * G);
            }
            if (r2 == SRK_OK) r2 = coll_group(h, SRK_COLL_BCAST, G, ptrs, counts);
            if (verify) {
                if (r2 == SRK_E_DEVICE) return selfcheck_failed("an RCCL call of the broadcast group returned an error");
                bad = 0;
                for (int k = 0; k < G && r2 == SRK_OK; ++k) {
                    double got[2];
                    r2 = dp_checksum(h, h->att[k].stream, ptrs[k], counts[k], got);
                    // (a broadcast copies bits and the checksum has a fixed order: equal, not close; NaN corrections of a failed
                    // solve compare unequal to themselves and are let through -- the status words deal with them)
                    if (got[0] == got[0] && rootv[2 * k] == rootv[2 * k] && (got[0] != rootv[2 * k] || got[1] != rootv[2 * k + 1])) bad += 1;
                }
#ifdef SRK_DEV
                if (g_dp_corrupt == 2) bad += 1, g_dp_corrupt = 0;
#endif
                if (r2 == SRK_OK) r2 = dp_allreduce_host(h, &bad, 1);
                if (r2 == SRK_OK && bad > 0) return selfcheck_failed("broadcast corrections differ from the solving rank's");
                if (r2 == SRK_OK) h->dp_verified = true;
            }
            for (int k = 0; k < G && r2 == SRK_OK; ++k) {
                select_attempt(h, k);
                r2 = phase_backsub_apply(h, cf[k]);
                if (r2 == SRK_OK) r2 = phase_cam_apply(h);
                if (r2 == SRK_OK) r2 = phase_error(h, h->A->trial, nullptr, true, true);
            }
            select_attempt(h, 0);
            if (r2 != SRK_OK) return r2;
            // the status words of all slots in one all-reduce, then one read-back
            if (h->comm) {
                for (int k = 0; k < G; ++k) {
                    if (hipEventRecord(h->att[k].ev_b, h->att[k].stream) != hipSuccess ||
                        hipStreamWaitEvent(h->comm_stream, h->att[k].ev_b, 0) != hipSuccess) return SRK_E_DEVICE;
                }
                ncclResult_t nr = rccl().AllReduce(h->status_all.p, h->status_all.p, (size_t)(8 * G), ncclDouble, ncclSum, h->comm, h->comm_stream);
                if (nr != ncclSuccess) { h->last_error = std::string("ncclAllReduce: ") + rccl().GetErrorString(nr); return SRK_E_DEVICE; }
                if (hipMemcpyAsync(h->dp_back, h->status_all.p, (size_t)(64 * G), hipMemcpyDeviceToHost, h->comm_stream) != hipSuccess ||
                    hipStreamSynchronize(h->comm_stream) != hipSuccess) return SRK_E_DEVICE;
            } else {
                for (int k = 0; k < G; ++k)
                    if (hipStreamSynchronize(h->att[k].stream) != hipSuccess) return SRK_E_DEVICE;
                if (h->allreduce(h->allreduce_ctx, P<double>(h->status_all), 8 * G) != 0) { h->last_error = "allreduce hook failed"; return SRK_E_DEVICE; }
                if (hipMemcpy(h->dp_back, h->status_all.p, (size_t)(64 * G), hipMemcpyDeviceToHost) != hipSuccess) return SRK_E_DEVICE;
            }
            for (int k = 0; k < G; ++k)
                for (int e = 0; e < 3; ++e) h->att[k].host_back[e] = h->dp_back[8 * k + e];
            return SRK_OK;
        };
        while (dp_mode && !decrease) {
            double cf[SRK_SLOTS];
            int G = 0;
            for (double cc = hessian_factor; G < SRK_SLOTS && h->att[G].allocated; cc *= 10) {
                if (G > 0 && max_hessian_factor && cc > *max_hessian_factor) break; // the reference stops before such an attempt (:843-847)
                cf[G++] = cc;
            }
            rc = dp_round(G, cf);
            if (rc == SRK_RETRY_ALLREDUCE) { // the first native round failed its self-check (every rank saw the same verdict):
                // nothing of it was judged; the systems are rebuilt and this iteration's attempts run through the all-reduce schedule
                dp_mode = false;
                h->poisoned = true;
                rc = clear_poison(h);
                if (rc != SRK_OK) return fail_device(rc);
                break;
            }
            if (rc != SRK_OK) return fail_device(rc);
            bool timeout = false;
            for (int k = 0; k < G; ++k) timeout = timeout || (((int)h->att[k].host_back[1] & 8) != 0);
            if (timeout && h->chol_fused) { // a hand-off of the fused solve timed out on the rank that solved: the round again, unfused, everywhere
                ++h->sync_timeouts;
                h->chol_fused = false;
                for (auto& a : h->att) a.sync.fused = false;
                h->poisoned = true;
                rc = clear_poison(h);
                if (rc != SRK_OK) return fail_device(rc);
                continue;
            }
            for (int k = 0; k < G && !decrease; ++k) {
                rc = judge_attempt(k);
                if (rc != SRK_OK) return fail_device(rc);
            }
        }
        while (!decrease) {
            // speculate once this optimise call has seen a rejection (or from its second iteration on): the first
            // iteration of a fresh scene is usually accepted at once -- and after a rejected pair the third attempt
            // usually is the last one: it runs alone unless the previous iteration needed four or more (then the damping
            // factor has a long way to climb and pairs pay again)
            const bool pair_pays = round == 0 ? (spec_wanted || rep->iterations >= 1) : (round >= 2 || prev_attempts >= 4);
            // (Round 3, measured and dropped: TRIPLES when the previous iteration needed three attempts -- late in a run on
            // the circle-grid scenes 16 of 20 iterations reject c / 10 and c and accept 10 c.  Three solves side by side are
            // slower than a pair plus a lone attempt, 299-302 against 331 it/s: one solve's fused outer steps hold ~200
            // workgroups of 66 KB LDS, two solves fill the chip's LDS, the third waits for slots.  Holding the solves back
            // until the last Schur sum is done: 325 against 331 it/s with pairs, 283-299 with triples.)
            const int want = (can_speculate && pair_pays) ? 2 : 1;
            int n_now = 1; // attempts enqueued this round: the factors hessian_factor * 10^k that the cap allows
            for (double cc = hessian_factor * 10; n_now < want && !(max_hessian_factor && cc > *max_hessian_factor); cc *= 10) ++n_now;
            const bool speculate_now = n_now >= 2;
            ++round;
            // one rank: all Schur sums first (a later one would otherwise wait behind ~0.6 ms of launch calls), then the
            // solves.  Several ranks: a Schur phase ends in a blocking exchange, so slot 0's solve is enqueued before it
            // and runs under slot 1's Schur sum and exchange.
            double cfk[SRK_SLOTS];
            for (int k = 0; k < SRK_SLOTS; ++k) cfk[k] = k == 0 ? hessian_factor : cfk[k - 1] * 10;
            rc = enqueue_schur(0, cfk[0]);
            if (h->allreduce || h->comm) {
                if (rc == SRK_OK) rc = enqueue_rest(0, cfk[0]);
                if (rc == SRK_OK && speculate_now) rc = enqueue_schur(1, cfk[1]);
            } else {
                for (int k = 1; k < n_now && rc == SRK_OK; ++k) rc = enqueue_schur(k, cfk[k]);
                if (rc == SRK_OK) rc = enqueue_rest(0, cfk[0]);
            }
            for (int k = 1; k < n_now && rc == SRK_OK; ++k) rc = enqueue_rest(k, cfk[k]);
            if (rc != SRK_OK) return fail_device(rc);
            for (int k = 1; k < n_now; ++k) spec_in_flight |= 1u << k;
            rc = judge_attempt(0);
            if (rc != SRK_OK) return fail_device(rc);
            if (!jac_timed) {
                rep->ms_jacobian += ev_ms(0, 1);
                rep->ms_jacobian_kernel += ev_ms(12, 13);
                jac_timed = true;
            }
            rep->ms_schur += ev_ms(2, 3);
            rep->ms_solve += ev_ms(3, 4);
            rep->ms_backsub += ev_ms(4, 5);
            rep->ms_apply += ev_ms(5, 6);
            rep->ms_error += ev_ms(6, 7);
            if (h->profile_level >= 2) {
                for (size_t kb = 0; kb < h->att[0].solve_prof.n; ++kb) {
                    float ms = 0;
                    if (hipEventElapsedTime(&ms, h->chol_ev[2 * kb], h->chol_ev[2 * kb + 1]) == hipSuccess)
                        rep->ms_solve_syrk += ms;
                }
                rep->solve_mfma_flops += h->att[0].solve_prof.flops;
            }
            if (!decrease) spec_wanted = true; // a rejection: from now on pairs pay
            for (int k = 1; k < n_now && !decrease; ++k) { // the successors were computed meanwhile, with exactly these factors
                rc = judge_attempt(k);
                if (rc != SRK_OK) return fail_device(rc);
                spec_in_flight &= ~(1u << k);
            }
        }
        prev_attempts = rep->attempts - attempts_before;
        if (spec_in_flight) {
            // speculative attempts nobody needs are still running: later work on the main stream (the next derivatives
            // overwrite what they read) must come after them; nothing on the host waits
            for (int k = 1; k < SRK_SLOTS; ++k)
                if (spec_in_flight & (1u << k)) HIPCHK(h, hipStreamWaitEvent(s, h->att[k].done, 0));
            spec_in_flight = 0;
            spec_drain = true;
        }
        if (decrease != 1) { // :857-873
            rep->status = decrease == 2 ? SRK_STATUS_HESSIAN_OVERFLOW : SRK_STATUS_ERR_CONVERGED;
            result_true = false;
            break;
        }
        { // accept: the winning slot's trial scene becomes current, the old current set becomes that slot's trial set
            const int newcur = h->att[accepted_slot].trial;
            h->att[accepted_slot].trial = h->cur;
            h->cur = newcur;
        }
        rep->iterations += 1;
        if (h->iter_log.size() < (size_t)1 << 20)
            h->iter_log.push_back({ (int32_t)prev_attempts, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(),
                                    err_new, hessian_factor });
        double change = err_new - err_value;
        rep->err_final = err_new;
        if (allowed_err_change && std::fabs(change) < *allowed_err_change) { // :880-884
            rep->status = SRK_STATUS_SMALL_ERR_CHANGE;
            result_true = true;
            break;
        }
        err_value = err_new;
        hessian_factor /= 10; // :889
    }
    if (spec_drain) {
        for (int k = 1; k < SRK_SLOTS; ++k) {
            if (!h->att[k].allocated) continue;
            HIPCHK(h, hipStreamSynchronize(h->att[k].stream)); // leave no speculative work behind
            if (h->att[k].host_back[1] != 0.0 || h->att[k].host_back[2] != 0.0) h->poisoned = true; // a dropped attempt that failed
        }
    }
    select_attempt(h, 0);
    rep->hessian_factor = hessian_factor;
    rep->optimized = result_true ? 1 : 0;
    rep->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return result_true ? 0 : 1;
}

int srk_ba_compute_inplace(srk_ba* h, double f0, int64_t N, double* pts, int32_t M, double* cam_R, double* cam_T,
                           const double* K, int shared_k, const int64_t* row_ptr, const int32_t* obs_frame,
                           const double* obs_uv, const double* allowed_err_change, const double* max_hessian_factor,
                           int64_t max_iterations, srk_ba_report* rep)
{
    srk_ba_report local;
    if (!rep) rep = &local;
    std::memset(rep, 0, sizeof *rep);
    if (!h) return SRK_E_ARGS;
    int rc = srk_ba_upload_scene(h, f0, N, pts, M, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv, 0);
    if (rc == 1) { // normalisation failed: reference returns false with an empty status string (:681-682)
        rep->status = SRK_STATUS_NONE;
        return 1;
    }
    if (rc != SRK_OK) return rc;
    int result = srk_ba_optimize(h, allowed_err_change, max_hessian_factor, max_iterations, rep);
    if (result < 0) return result;
    rc = srk_ba_download_scene(h, pts, cam_R, cam_T, 1);
    if (rc != SRK_OK) return rc;
    return result;
}

// Standalone scoring: ReprojError callers that only score a scene (multi-view-factorization.cpp:373,409-413) do not
// need the LM state -- no gauge normalisation, landmark sort, grouping or solver plan; the uploaded BA scene (if any)
// stays untouched.  z_tol < 0 keeps every observation (BundleAdjustmentKanatani::ReprojError, :589-600).
static int score_scene(srk_ba* h, double f0, int64_t N, const double* pts, int32_t M, const double* cam_R,
                       const double* cam_T, const double* K, int shared_k, const int64_t* row_ptr,
                       const int32_t* obs_frame, const double* obs_uv, int32_t min_frames, double z_tol, double* err,
                       int64_t* count)
{
    int rc = validate_scene(h, f0, N, pts, M, cam_R, cam_T, K, row_ptr, obs_frame, obs_uv, min_frames);
    if (rc != SRK_OK) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int64_t O = row_ptr[N];
    std::vector<int32_t> obs_pt((size_t)O);
    for (int64_t i = 0; i < N; ++i)
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o) obs_pt[(size_t)o] = (int32_t)i;
    std::vector<double> Kexp(9 * (size_t)M);
    for (int32_t j = 0; j < M; ++j) std::memcpy(&Kexp[9 * (size_t)j], shared_k ? K : K + 9 * (int64_t)j, 72);
    SrkDims d{};
    d.O = O;
    const int32_t np = srk_error_partials(d);
    struct { DevBuf* b; const void* src; size_t bytes; } up[] = {
        { &h->sc_pts, pts, (size_t)(24 * N) },          { &h->sc_R, cam_R, (size_t)(72 * (int64_t)M) },
        { &h->sc_T, cam_T, (size_t)(24 * (int64_t)M) }, { &h->sc_K, Kexp.data(), (size_t)(72 * (int64_t)M) },
        { &h->sc_frame, obs_frame, (size_t)(4 * O) },   { &h->sc_pt, obs_pt.data(), (size_t)(4 * O) },
        { &h->sc_uv, obs_uv, (size_t)(16 * O) },
    };
    for (auto& u : up) {
        if ((rc = dev_alloc(h, *u.b, u.bytes)) != SRK_OK) return rc;
        if (u.bytes) HIPCHK(h, hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, s));
    }
    if ((rc = dev_alloc(h, h->sc_cam, (size_t)(8 * SRK_CAM_PACK * (int64_t)M))) != SRK_OK) return rc;
    if ((rc = dev_alloc(h, h->sc_partial, (size_t)(16 * np))) != SRK_OK) return rc;
    if ((rc = dev_alloc(h, h->sc_out, 16)) != SRK_OK) return rc;
    srk_launch_cam_pack(s, M, P<double>(h->sc_R), P<double>(h->sc_T), P<double>(h->sc_K), f0, P<double>(h->sc_cam));
    srk_launch_error_score(s, O, P<double>(h->sc_pts), P<double>(h->sc_cam), P<int32_t>(h->sc_frame), P<int32_t>(h->sc_pt),
                           P<double>(h->sc_uv), z_tol, P<double>(h->sc_partial), np, P<double>(h->sc_out));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->A->host_back, h->sc_out.p, 16, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    *err = h->A->host_back[0];
    *count = (int64_t)h->A->host_back[1];
    return SRK_OK;
}

double srk_ba_reproj_error(srk_ba* h, double f0, int64_t N, const double* pts, int32_t M, const double* cam_R,
                           const double* cam_T, const double* K, int shared_k, const int64_t* row_ptr,
                           const int32_t* obs_frame, const double* obs_uv, int64_t* seen)
{
    if (!h) return std::nan("");
    double e = std::nan("");
    int64_t cnt = 0;
    if (score_scene(h, f0, N, pts, M, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv, 2, -1.0, &e, &cnt) != SRK_OK)
        return std::nan("");
    if (seen) *seen = cnt;
    return e;
}

int srk_ba_reproj_error_mvf(srk_ba* h, double f0, int64_t N, const double* pts, int32_t M, const double* cam_R,
                            const double* cam_T, const double* K, int shared_k, const int64_t* row_ptr,
                            const int32_t* obs_frame, const double* obs_uv, double z_tol, double* reproj_err,
                            int64_t* summands)
{
    if (!h || !reproj_err || z_tol < 0) return SRK_E_ARGS;
    double e = 0;
    int64_t cnt = 0;
    int rc = score_scene(h, f0, N, pts, M, cam_R, cam_T, K, shared_k, row_ptr, obs_frame, obs_uv, 1, z_tol, &e, &cnt);
    if (rc != SRK_OK) return rc;
    if (summands) *summands = cnt;
    if (cnt == 0) return 0; // no points, or every point at infinity: the reference returns false (:470-471)
    *reproj_err = e;
    return 1;
}

// ------------------------------------------------------------------ f32 boundary (suriko_scalar_type_string = f32)
// A reference built with Scalar = float (rt-config.h:41-48, suriko-engine/CMakeLists.txt:14-15,76-82) hands over float
// arrays.  They are widened here, the fp64 pipeline runs unchanged, the result is rounded back: the arithmetic is
// strictly more accurate than the reference's own f32 build (an f32 device pipeline is not implemented).
extern "C" int srk_ba_compute_inplace_f32(srk_ba* h, float f0, int64_t N, float* pts, int32_t M, float* cam_R, float* cam_T,
                                          const float* K, int shared_k, const int64_t* row_ptr, const int32_t* obs_frame,
                                          const float* obs_uv, const float* allowed_err_change,
                                          const float* max_hessian_factor, int64_t max_iterations, srk_ba_report* out)
{
    if (!h) return SRK_E_ARGS;
    if (N < 0 || M < 1 || !row_ptr || !cam_R || !cam_T || !K || (N > 0 && !pts)) {
        h->last_error = "null scene array";
        return SRK_E_ARGS;
    }
    const int64_t O = row_ptr[N];
    if (O > 0 && !obs_uv) { h->last_error = "null observation arrays"; return SRK_E_ARGS; }
    auto widen = [](const float* p, size_t n) { return std::vector<double>(p, p + n); };
    std::vector<double> dp = widen(pts, (size_t)(3 * N)), dR = widen(cam_R, 9 * (size_t)M), dT = widen(cam_T, 3 * (size_t)M),
                        dK = widen(K, shared_k ? 9 : 9 * (size_t)M), duv = widen(obs_uv, (size_t)(2 * O));
    double a = allowed_err_change ? (double)*allowed_err_change : 0, m = max_hessian_factor ? (double)*max_hessian_factor : 0;
    int rc = srk_ba_compute_inplace(h, (double)f0, N, dp.data(), M, dR.data(), dT.data(), dK.data(), shared_k, row_ptr,
                                    obs_frame, duv.data(), allowed_err_change ? &a : nullptr,
                                    max_hessian_factor ? &m : nullptr, max_iterations, out);
    if (rc < 0) return rc;
    for (size_t i = 0; i < dp.size(); ++i) pts[i] = (float)dp[i];
    for (size_t i = 0; i < dR.size(); ++i) cam_R[i] = (float)dR[i];
    for (size_t i = 0; i < dT.size(); ++i) cam_T[i] = (float)dT[i];
    return rc;
}

// ------------------------------------------------------------------ multi-view-factorization steps
namespace {
// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major, destroyed): V's columns = eigenvectors
void jacobi_eig(int n, double* A, double* V, double* ev)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; ++i) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-34 * diag || off == 0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (apq == 0) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(1 + theta * theta));
                double c = 1 / std::sqrt(1 + t * t), sn = c * t;
                for (int k = 0; k < n; ++k) { // columns p, q
                    double x = A[k * n + p], y = A[k * n + q];
                    A[k * n + p] = c * x - sn * y;
                    A[k * n + q] = sn * x + c * y;
                }
                for (int k = 0; k < n; ++k) { // rows p, q
                    double x = A[p * n + k], y = A[q * n + k];
                    A[p * n + k] = c * x - sn * y;
                    A[q * n + k] = sn * x + c * y;
                }
                for (int k = 0; k < n; ++k) {
                    double x = V[k * n + p], y = V[k * n + q];
                    V[k * n + p] = c * x - sn * y;
                    V[k * n + q] = sn * x + c * y;
                }
            }
    }
    for (int i = 0; i < n; ++i) ev[i] = A[i * n + i];
}

// ProjectOntoSO3 (multi-view-factorization.cpp:79-104; MASKS 8.41, 8.42): R = sign(det(U V^T)) U V^T from the SVD of
// the noisy R (via the eigen-decomposition of R^T R), T scaled by sign / cbrt(det S).  false when det S ~ 0.
bool project_onto_so3(const double Rn[9], const double Tn[3], double R[9], double T[3])
{
    double G[9], V[9], ev[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) G[3 * i + j] = Rn[i] * Rn[j] + Rn[3 + i] * Rn[3 + j] + Rn[6 + i] * Rn[6 + j];
    jacobi_eig(3, G, V, ev);
    double sv[3];
    for (int k = 0; k < 3; ++k) sv[k] = std::sqrt(std::max(ev[k], 0.0));
    double det_S = sv[0] * sv[1] * sv[2];
    if (srk::is_close(0.0, det_S)) return false; // :88-89
    double U[9]; // U = R V S^-1
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) U[3 * i + k] = (Rn[3 * i] * V[k] + Rn[3 * i + 1] * V[3 + k] + Rn[3 * i + 2] * V[6 + k]) / sv[k];
    double ng[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) ng[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    double det = ng[0] * (ng[4] * ng[8] - ng[5] * ng[7]) - ng[1] * (ng[3] * ng[8] - ng[5] * ng[6]) +
                 ng[2] * (ng[3] * ng[7] - ng[4] * ng[6]);
    int sign = det >= 0 ? 1 : -1; // approx-alg.h:41
    for (int i = 0; i < 9; ++i) R[i] = sign * ng[i];
    double sc = sign / std::cbrt(det_S);
    for (int i = 0; i < 3; ++i) T[i] = sc * Tn[i];
    return true;
}
} // namespace

extern "C" int srk_mvf_project_onto_so3(const double* R_noisy, const double* T_noisy, double* R_out, double* T_out)
{
    if (!R_noisy || !T_noisy || !R_out || !T_out) return SRK_E_ARGS;
    return project_onto_so3(R_noisy, T_noisy, R_out, T_out) ? 1 : 0;
}

extern "C" int srk_mvf_estimate_depths(srk_ba* h, int64_t n_tracks, const int64_t* row_ptr, const int32_t* frame,
                                       const double* x_meter, int32_t n_frames, const double* cam_R, const double* cam_T,
                                       double* depth_out)
{
    if (!h) return SRK_E_ARGS;
    if (n_tracks < 0 || n_frames < 1 || !row_ptr || !cam_R || !cam_T || (n_tracks > 0 && !depth_out) || row_ptr[0] != 0) {
        h->last_error = "srk_mvf_estimate_depths: bad arguments";
        return SRK_E_ARGS;
    }
    const int64_t O = row_ptr[n_tracks];
    if (O > 0 && (!frame || !x_meter)) { h->last_error = "srk_mvf_estimate_depths: null observation arrays"; return SRK_E_ARGS; }
    for (int64_t i = 0; i < n_tracks; ++i) {
        if (row_ptr[i + 1] < row_ptr[i]) { h->last_error = "row_ptr not monotone"; return SRK_E_ARGS; }
        for (int64_t o = row_ptr[i]; o < row_ptr[i + 1]; ++o)
            if (frame[o] < 0 || frame[o] >= n_frames) { h->last_error = "frame out of range"; return SRK_E_ARGS; }
    }
    if (n_tracks == 0) return SRK_OK;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int rc;
    struct { DevBuf* b; const void* src; size_t bytes; } up[] = {
        { &h->sc_pts, x_meter, (size_t)(24 * O) },       { &h->sc_R, cam_R, (size_t)(72 * (int64_t)n_frames) },
        { &h->sc_T, cam_T, (size_t)(24 * (int64_t)n_frames) }, { &h->sc_frame, frame, (size_t)(4 * O) },
        { &h->sc_uv, row_ptr, (size_t)(8 * (n_tracks + 1)) },
    };
    for (auto& u : up) {
        if ((rc = dev_alloc(h, *u.b, u.bytes)) != SRK_OK) return rc;
        if (u.bytes) HIPCHK(h, hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, s));
    }
    if ((rc = dev_alloc(h, h->sc_partial, (size_t)(8 * n_tracks))) != SRK_OK) return rc;
    srk_launch_mvf_depth(s, n_tracks, P<int64_t>(h->sc_uv), P<int32_t>(h->sc_frame), P<double>(h->sc_pts), P<double>(h->sc_R),
                         P<double>(h->sc_T), P<double>(h->sc_partial));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(depth_out, h->sc_partial.p, (size_t)(8 * n_tracks), hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    return SRK_OK;
}

extern "C" int srk_mvf_relative_motion(srk_ba* h, int64_t n_points, const double* x_anchor, const double* x_target,
                                       const double* depth_anchor, double* R_out, double* T_out)
{
    if (!h) return SRK_E_ARGS;
    // every point contributes two independent equations ([x2]x has rank 2) towards the 11 needed for a unique
    // null vector of the 12 unknowns: fewer than 6 points leave the answer arbitrary (the reference does not check)
    if (n_points < 6 || !x_anchor || !x_target || !depth_anchor || !R_out || !T_out) {
        h->last_error = "srk_mvf_relative_motion: need at least 6 common points and non-null arrays";
        return SRK_E_ARGS;
    }
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int rc;
    const int64_t nblk = (n_points + 255) / 256;
    struct { DevBuf* b; const void* src; size_t bytes; } up[] = {
        { &h->sc_pts, x_anchor, (size_t)(24 * n_points) }, { &h->sc_uv, x_target, (size_t)(24 * n_points) },
        { &h->sc_R, depth_anchor, (size_t)(8 * n_points) },
    };
    for (auto& u : up) {
        if ((rc = dev_alloc(h, *u.b, u.bytes)) != SRK_OK) return rc;
        HIPCHK(h, hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, s));
    }
    if ((rc = dev_alloc(h, h->sc_partial, (size_t)(8 * 78 * nblk))) != SRK_OK) return rc;
    srk_launch_mvf_gram(s, n_points, P<double>(h->sc_pts), P<double>(h->sc_uv), P<double>(h->sc_R), P<double>(h->sc_partial));
    HIPCHK(h, hipGetLastError());
    std::vector<double> part((size_t)(78 * nblk));
    HIPCHK(h, hipMemcpyAsync(part.data(), h->sc_partial.p, part.size() * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    double G[144], V[144], ev[12];
    int e = 0;
    for (int a = 0; a < 12; ++a)
        for (int b = a; b < 12; ++b, ++e) {
            double sum = 0;
            for (int64_t k = 0; k < nblk; ++k) sum += part[(size_t)(78 * k + e)]; // fixed order
            G[a * 12 + b] = G[b * 12 + a] = sum;
        }
    jacobi_eig(12, G, V, ev);
    int jmin = 0;
    for (int j = 1; j < 12; ++j)
        if (ev[j] < ev[jmin]) jmin = j;
    double Rn[9], Tn[3];
    for (int col = 0; col < 3; ++col) // vec(R) is column-major in the reference (:174)
        for (int row = 0; row < 3; ++row) Rn[3 * row + col] = V[(3 * col + row) * 12 + jmin];
    for (int i = 0; i < 3; ++i) Tn[i] = V[(9 + i) * 12 + jmin];
    return project_onto_so3(Rn, Tn, R_out, T_out) ? 1 : 0;
}

// ------------------------------------------------------------------ downloads for the parity tests

int64_t srk_ba_buffer_size(srk_ba* h, int which)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    const SrkDims& d = h->d;
    switch (which) {
    case SRK_BUF_GRAD: return 3 * d.N + 10 * (int64_t)d.M;
    case SRK_BUF_POINT_BLOCKS: return 9 * d.N;
    case SRK_BUF_FRAME_BLOCKS: return 100 * (int64_t)d.M;
    case SRK_BUF_POINT_FRAME: return 30 * d.O;
    case SRK_BUF_RCS: return 100 * (int64_t)d.M * d.M;
    case SRK_BUF_RCS_RHS: return 10 * (int64_t)d.M;
    case SRK_BUF_CORRECTIONS: return 3 * d.N + 10 * (int64_t)d.M;
    case SRK_BUF_POINTS: return 3 * d.N;
    case SRK_BUF_CAM_R: return 9 * (int64_t)d.M;
    case SRK_BUF_CAM_T: return 3 * (int64_t)d.M;
    default: return SRK_E_ARGS;
    }
}

int srk_ba_download(srk_ba* h, int which, double* dst, int64_t count)
{
    if (!h || !h->have_scene || !dst) return SRK_E_STATE;
    if (count != srk_ba_buffer_size(h, which)) { h->last_error = "download: wrong count"; return SRK_E_ARGS; }
    HIPCHK(h, hipSetDevice(h->device));
    const SrkDims& d = h->d;
    hipStream_t s = h->stream;
    HIPCHK(h, hipStreamSynchronize(s));
    for (int sl = 1; sl < SRK_SLOTS; ++sl) HIPCHK(h, hipStreamSynchronize(h->att[sl].stream));
    // reduced camera system, rhs and corrections: those of the last attempt the LM loop judged (or of the staged calls)
    struct SlotGuard {
        srk_ba* h;
        ~SlotGuard() { h->A = &h->att[0]; }
    } guard{ h };
    h->A = &h->att[h->last_slot];
    auto d2h = [&](void* dstp, const void* src, size_t bytes) -> int {
        if (bytes == 0) return SRK_OK;
        HIPCHK(h, hipMemcpy(dstp, src, bytes, hipMemcpyDeviceToHost));
        return SRK_OK;
    };
    int rc = SRK_OK;
    switch (which) {
    case SRK_BUF_GRAD:
    case SRK_BUF_POINT_BLOCKS: {
        std::vector<double> vg((size_t)(9 * d.Ns));
        if ((rc = d2h(vg.data(), h->Vg.p, vg.size() * 8)) != SRK_OK) return rc;
        if (which == SRK_BUF_POINT_BLOCKS) {
            static const int map[9] = { 0, 1, 2, 1, 3, 4, 2, 4, 5 };
            for (int64_t i = 0; i < d.N; ++i)
                for (int e = 0; e < 9; ++e) dst[9 * h->perm[(size_t)i] + e] = vg[(size_t)(map[e] * d.Ns + i)];
            return SRK_OK;
        }
        for (int64_t i = 0; i < d.N; ++i)
            for (int e = 0; e < 3; ++e) dst[3 * h->perm[(size_t)i] + e] = vg[(size_t)((6 + e) * d.Ns + i)];
        std::vector<double> ug((size_t)(SRK_UG * (int64_t)d.M));
        if ((rc = d2h(ug.data(), h->Ug.p, ug.size() * 8)) != SRK_OK) return rc;
        for (int32_t j = 0; j < d.M; ++j)
            for (int e = 0; e < 10; ++e) dst[3 * d.N + 10 * (int64_t)j + e] = ug[(size_t)(SRK_UG * (int64_t)j + 55 + e)];
        frames_to_user(h, dst + 3 * d.N, 10);
        return SRK_OK;
    }
    case SRK_BUF_FRAME_BLOCKS: {
        std::vector<double> ug((size_t)(SRK_UG * (int64_t)d.M));
        if ((rc = d2h(ug.data(), h->Ug.p, ug.size() * 8)) != SRK_OK) return rc;
        for (int32_t j = 0; j < d.M; ++j)
            for (int v1 = 0; v1 < 10; ++v1)
                for (int v2 = 0; v2 < 10; ++v2) {
                    int a = v1 < v2 ? v1 : v2, b = v1 < v2 ? v2 : v1;
                    dst[100 * (int64_t)j + 10 * v1 + v2] = ug[(size_t)(SRK_UG * (int64_t)j + a * 10 - a * (a - 1) / 2 + (b - a))];
                }
        frames_to_user(h, dst, 100);
        return SRK_OK;
    }
    case SRK_BUF_POINT_FRAME: {
        std::vector<double> w((size_t)(30 * d.Os));
        {
            // the library keeps the rank-2 factors (srk_dev.hpp SRK_WF_*; as floats in the f32 storage mode): the products are formed here
            std::vector<double> f((size_t)(SRK_WF_PLANES * d.Os));
            if (d.w_f32) {
                std::vector<float> ff(f.size());
                if ((rc = d2h(ff.data(), h->W.p, ff.size() * 4)) != SRK_OK) return rc;
                for (size_t i = 0; i < ff.size(); ++i) f[i] = (double)ff[i];
            } else if ((rc = d2h(f.data(), h->W.p, f.size() * 8)) != SRK_OK) return rc;
            auto F = [&](int plane, int64_t o) { return plane >= 0 ? f[(size_t)(plane * d.Os + o)] : 0.0; };
            for (int64_t o = 0; o < d.O; ++o)
                for (int pv = 0; pv < 3; ++pv)
                    for (int fv = 0; fv < 10; ++fv) {
                        const int pa = fv >= 4 ? SRK_WF_AF4 + fv - 4 : (fv == 0 ? SRK_WF_AF0 : (fv == 2 ? SRK_WF_G : -1));
                        const int pb = fv >= 4 ? SRK_WF_BF4 + fv - 4 : (fv == 1 ? SRK_WF_BF1 : (fv == 3 ? SRK_WF_G : -1));
                        w[(size_t)((10 * pv + fv) * d.Os + o)] = F(SRK_WF_AP + pv, o) * F(pa, o) + F(SRK_WF_BP + pv, o) * F(pb, o);
                    }
        }
        for (int64_t i = 0; i < d.N; ++i) {
            int64_t oi = h->row_ptr_int[(size_t)i], ou = h->row_ptr_user[(size_t)h->perm[(size_t)i]];
            int64_t cnt = h->row_ptr_int[(size_t)i + 1] - oi;
            for (int64_t a = 0; a < cnt; ++a) { // the caller's observation ou + a: internal place obs_rank inside its landmark
                const int64_t ai = h->obs_rank.empty() ? a : h->obs_rank[(size_t)(ou + a)];
                for (int k = 0; k < 30; ++k) dst[30 * (ou + a) + k] = w[(size_t)(k * d.Os + oi + ai)];
            }
        }
        return SRK_OK;
    }
    case SRK_BUF_RCS: {
        int64_t n = 10 * (int64_t)d.M;
        std::vector<double> row((size_t)d.ld);
        auto uvar = [&](int64_t v) { return h->frame_user.empty() ? v : 10 * (int64_t)h->frame_user[(size_t)(v / 10)] + v % 10; };
        for (int64_t r = 0; r < n; ++r) {
            if ((rc = d2h(row.data(), P<double>(h->A->S) + r * d.ld, (size_t)(8 * n))) != SRK_OK) return rc;
            const int64_t ur = uvar(r);
            for (int64_t c = 0; c <= r; ++c) {
                const int64_t uc = uvar(c);
                dst[ur * n + uc] = row[(size_t)c];
                dst[uc * n + ur] = row[(size_t)c]; // lower triangle is authoritative
            }
        }
        return SRK_OK;
    }
    case SRK_BUF_RCS_RHS:
        if ((rc = d2h(dst, h->A->rhs.p, (size_t)(80 * (int64_t)d.M))) != SRK_OK) return rc;
        frames_to_user(h, dst, 10);
        return SRK_OK;
    case SRK_BUF_CORRECTIONS: {
        std::vector<double> tmp((size_t)(3 * d.N));
        if ((rc = d2h(tmp.data(), h->A->dx.p, (size_t)(24 * d.N))) != SRK_OK) return rc;
        for (int64_t i = 0; i < d.N; ++i) std::memcpy(dst + 3 * h->perm[(size_t)i], &tmp[(size_t)(3 * i)], 24);
        if ((rc = d2h(dst + 3 * d.N, h->A->dc.p, (size_t)(80 * (int64_t)d.M))) != SRK_OK) return rc;
        frames_to_user(h, dst + 3 * d.N, 10);
        return SRK_OK;
    }
    case SRK_BUF_POINTS: {
        std::vector<double> tmp((size_t)(3 * d.N));
        if ((rc = d2h(tmp.data(), h->pts[h->cur].p, (size_t)(24 * d.N))) != SRK_OK) return rc;
        for (int64_t i = 0; i < d.N; ++i) std::memcpy(dst + 3 * h->perm[(size_t)i], &tmp[(size_t)(3 * i)], 24);
        return SRK_OK;
    }
    case SRK_BUF_CAM_R:
        if ((rc = d2h(dst, h->camR[h->cur].p, (size_t)(72 * (int64_t)d.M))) != SRK_OK) return rc;
        frames_to_user(h, dst, 9);
        return SRK_OK;
    case SRK_BUF_CAM_T:
        if ((rc = d2h(dst, h->camT[h->cur].p, (size_t)(24 * (int64_t)d.M))) != SRK_OK) return rc;
        frames_to_user(h, dst, 3);
        return SRK_OK;
    default: return SRK_E_ARGS;
    }
}

int srk_ba_download_rcs_rows(srk_ba* h, const int64_t* rows, int64_t n_rows, double* dst)
{
    if (!h || !h->have_scene || !rows || !dst || n_rows < 0) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int sl = 1; sl < SRK_SLOTS; ++sl) HIPCHK(h, hipStreamSynchronize(h->att[sl].stream));
    const SrkDims& d = h->d;
    const int64_t n = 10 * (int64_t)d.M;
    const double* S = P<double>(h->att[h->last_slot].S);
    std::vector<double> rowi, coli;
    for (int64_t k = 0; k < n_rows; ++k) {
        const int64_t r = rows[k];
        if (r < 0 || r >= n) { h->last_error = "download_rcs_rows: row out of range"; return SRK_E_ARGS; }
        std::memset(dst + k * n, 0, (size_t)(8 * n));
        if (h->frame_int.empty()) {
            HIPCHK(h, hipMemcpy(dst + k * n, S + r * d.ld, (size_t)(8 * (r + 1)), hipMemcpyDeviceToHost));
            continue;
        }
        // reordered frames: the caller's row r is internal row ri; its entries left of the diagonal in the CALLER's order
        // lie in internal row ri (internal columns <= ri) and in internal column ri (rows > ri)
        const int64_t ri = 10 * (int64_t)h->frame_int[(size_t)(r / 10)] + r % 10;
        rowi.assign((size_t)n, 0.0);
        coli.assign((size_t)n, 0.0);
        HIPCHK(h, hipMemcpy(rowi.data(), S + ri * d.ld, (size_t)(8 * (ri + 1)), hipMemcpyDeviceToHost));
        if (ri + 1 < n)
            HIPCHK(h, hipMemcpy2D(coli.data() + ri + 1, 8, S + (ri + 1) * d.ld + ri, (size_t)(8 * d.ld), 8, (size_t)(n - ri - 1), hipMemcpyDeviceToHost));
        for (int64_t ci = 0; ci < n; ++ci) {
            const int64_t c = 10 * (int64_t)h->frame_user[(size_t)(ci / 10)] + ci % 10;
            if (c <= r) dst[k * n + c] = ci <= ri ? rowi[(size_t)ci] : coli[(size_t)ci];
        }
    }
    return SRK_OK;
}

// ------------------------------------------------------------------ dense SPD solve on its own

int srk_ba_dense_spd_solve(srk_ba* h, int64_t n, const double* A, const double* b, double* x, double* ms_factor)
{
    if (!h || n <= 0 || !A || !b || !x) return SRK_E_ARGS;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int64_t ld = ((n + SRK_CHOL_NB - 1) / SRK_CHOL_NB) * SRK_CHOL_NB;
    DevBuf dA, dw, dy, dx, dinfo, ddinv;
    int rc;
    if ((rc = dev_alloc(h, ddinv, (size_t)(8 * 64 * ld))) != SRK_OK) return rc;
    if ((rc = dev_alloc(h, dA, (size_t)(8 * ld * ld))) != SRK_OK) return rc;
    if ((rc = dev_alloc(h, dw, (size_t)(8 * ld))) != SRK_OK) return rc;
    if ((rc = dev_alloc(h, dy, (size_t)(8 * ld))) != SRK_OK) return rc;
    if ((rc = dev_alloc(h, dx, (size_t)(8 * ld))) != SRK_OK) return rc;
    if ((rc = dev_alloc(h, dinfo, 64)) != SRK_OK) return rc;
    std::vector<double> Ap((size_t)(ld * ld), 0.0), bp((size_t)ld, 0.0);
    for (int64_t r = 0; r < ld; ++r) {
        if (r < n) std::memcpy(&Ap[(size_t)(r * ld)], A + r * n, (size_t)(8 * n));
        else Ap[(size_t)(r * ld + r)] = 1.0;
    }
    std::memcpy(bp.data(), b, (size_t)(8 * n));
    HIPCHK(h, hipMemcpyAsync(dA.p, Ap.data(), Ap.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(dw.p, bp.data(), bp.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemsetAsync(dinfo.p, 0, 4, s));
    HIPCHK(h, hipEventRecord(h->ev[14], s));
    DevBuf dflags;
    if ((rc = dev_alloc(h, dflags, 4 * SRK_SYNC_WORDS)) != SRK_OK) return rc;
    HIPCHK(h, hipMemsetAsync(dflags.p, 0, 4 * SRK_SYNC_WORDS, s));
    SrkCholSync sync;
    sync.flags = P<unsigned>(dflags);
    sync.fused = h->chol_fused;
    srk_chol_solve(s, ld, P<double>(dA), P<double>(dw), P<double>(dy), P<double>(dx), P<int>(dinfo), nullptr, nullptr,
                   P<double>(ddinv), nullptr, &sync, n);
    HIPCHK(h, hipEventRecord(h->ev[15], s));
    HIPCHK(h, hipGetLastError());
    int info = 0;
    std::vector<double> xs(bp.size());
    HIPCHK(h, hipMemcpyAsync(xs.data(), dx.p, xs.size() * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemcpyAsync(&info, dinfo.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    if (info & 8) { // a hand-off timed out (a scheduling event): the inputs again from the host copies, unfused, once
        ++h->sync_timeouts;
        sync.fused = false;
        HIPCHK(h, hipMemcpyAsync(dA.p, Ap.data(), Ap.size() * 8, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(dw.p, bp.data(), bp.size() * 8, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemsetAsync(dinfo.p, 0, 4, s));
        srk_chol_solve(s, ld, P<double>(dA), P<double>(dw), P<double>(dy), P<double>(dx), P<int>(dinfo), nullptr, nullptr,
                       P<double>(ddinv), nullptr, &sync, n);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(xs.data(), dx.p, xs.size() * 8, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(&info, dinfo.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    bp.swap(xs);
    if (ms_factor) {
        float ms = 0;
        hipEventElapsedTime(&ms, h->ev[14], h->ev[15]);
        *ms_factor = ms;
    }
    std::memcpy(x, bp.data(), (size_t)(8 * n));
    dev_free(dA); dev_free(dw); dev_free(dy); dev_free(dx); dev_free(dinfo); dev_free(ddinv); dev_free(dflags);
    return info ? 1 : 0;
}

// global covisibility for sharded runs: min_cv[j] = smallest frame index that shares a landmark with frame j,
// taken over ALL ranks' landmarks.  Call after srk_ba_upload_scene.  NULL = dense (full lower triangle).
int srk_ba_set_covisibility(srk_ba* h, const int32_t* min_cv)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    // a covisibility in the caller's numbering says nothing about the internal one: with a SUPPLIED frame order
    // (srk_ba_set_frame_order) min_cv is taken in that numbering, which the caller knows; an automatic one is one rank's own
    if (min_cv && !h->frame_int.empty() && !h->frame_order_supplied) {
        h->last_error = "set_covisibility: the frames of this scene were renumbered (srk_ba_set_frame_reordering 0 keeps the caller's order)";
        return SRK_E_STATE;
    }
    if (min_cv) {
        for (int32_t j = 0; j < h->d.M; ++j) {
            if (min_cv[j] < 0 || min_cv[j] > j) { h->last_error = "min_cv[j] must be in [0, j]"; return SRK_E_ARGS; }
            h->min_cv[(size_t)j] = min_cv[j];
        }
    } else {
        std::fill(h->min_cv.begin(), h->min_cv.end(), 0);
    }
    return build_envelope(h);
}

// 0 = treat the reduced camera system as dense (full lower triangle), 1 = exploit its skyline (default)
int srk_ba_set_rcs_mode(srk_ba* h, int use_envelope)
{
    if (!h) return SRK_E_ARGS;
    h->use_envelope = use_envelope != 0;
    h->use_chunks = use_envelope != 1; // 0 dense, 1 skyline in one chain, 2 (default) skyline cut into chunks
    if (h->have_scene) return build_envelope(h);
    return SRK_OK;
}

int srk_ba_rcs_chunks(srk_ba* h) { return (h && h->have_scene) ? h->A->plan.P : 0; }

// fraction of the lower triangle inside the skyline (1.0 = dense)
double srk_ba_rcs_fill(srk_ba* h)
{
    if (!h || !h->have_scene) return -1.0;
    double full = 0.5 * (double)h->d.ld * (double)h->d.ld;
    return (double)h->env_packed / full;
}

// flops executed by the MFMA trailing updates of one solve with the current skyline (2 flops per FMA)
double srk_ba_solve_mfma_flops(srk_ba* h)
{
    if (!h || !h->have_scene) return -1.0;
    SrkSolveProf dry;
    dry.dry = true; // walks the launch sequence of the current mode without launching anything
    launch_solve(h, &dry);
    return dry.flops;
}

// 1 (default) = an outer step of the blocked Cholesky is ONE launch whose workgroups hand tiles to each other (k_step256),
// 0 = the k_panel / k_upd64 launch sequence.  Bit-identical results; takes effect at once.
int srk_ba_set_solver_fusion(srk_ba* h, int on)
{
    if (!h || (on != 0 && on != 1)) return SRK_E_ARGS;
    h->chol_fused = h->chol_fused_wanted = on != 0;
    h->fusion_rearms_left = SRK_FUSION_RETRIES; // an explicit request renews the budget
    for (auto& a : h->att) a.sync.fused = h->chol_fused;
    return SRK_OK;
}
int64_t srk_ba_solver_sync_timeouts(srk_ba* h) { return h ? h->sync_timeouts : -1; }
int64_t srk_ba_iteration_log(srk_ba* h, int64_t cap, int32_t* attempts, double* ms_since_start, double* err, double* hessian_factor)
{
    if (!h) return -1;
    const int64_t n = (int64_t)h->iter_log.size();
    for (int64_t k = 0; k < n && k < cap; ++k) {
        if (attempts) attempts[k] = h->iter_log[k].attempts;
        if (ms_since_start) ms_since_start[k] = h->iter_log[k].ms;
        if (err) err[k] = h->iter_log[k].err;
        if (hessian_factor) hessian_factor[k] = h->iter_log[k].factor;
    }
    return n;
}
int srk_ba_solver_fusion(srk_ba* h) { return h ? (h->chol_fused ? 1 : 0) : -1; }

int srk_ba_set_deterministic(srk_ba* h, int on)
{
    if (!h) return SRK_E_ARGS;
    h->deterministic = on != 0;
    return SRK_OK;
}
int srk_ba_deterministic(srk_ba* h) { return (h && h->have_scene && h->det_active) ? 1 : 0; }
int srk_ba_set_multi_schedule(srk_ba* h, int mode)
{
    if (!h || mode < 0 || mode > 2) return SRK_E_ARGS;
    h->dp_schedule = mode != 0;
    h->dp_force = mode == 2;
    h->dp_selfcheck_failed = false;
    return SRK_OK;
}
int srk_ba_multi_schedule(srk_ba* h)
{
    if (!h) return -1;
    if (h->dp_selfcheck_failed) return 3;
    if (!h->dp_schedule) return 0;
    return h->dp_verified ? 2 : 1;
}
int srk_ba_set_speculation(srk_ba* h, int on)
{
    if (!h || (on != 0 && on != 1)) return SRK_E_ARGS;
    h->speculate = on != 0; // takes effect at the next upload (the second attempt slot is allocated there)
    return SRK_OK;
}

// internal frame order (frame_reorder): -1 = automatic (renumber when the caller's order is far from banded), 0 = never,
// 1 = whenever reverse Cuthill-McKee gives another order than the caller's; takes effect at the next upload
int srk_ba_set_frame_reordering(srk_ba* h, int mode)
{
    if (!h || mode < -1 || mode > 1) return SRK_E_ARGS;
    h->frame_order_mode = mode;
    return SRK_OK;
}
// the numbering to use at the next upload instead of the automatic one (NULL: automatic again); with landmark shards every
// rank must be given the same one, found on the WHOLE scene (srk_frame_order)
int srk_ba_set_frame_order(srk_ba* h, const int32_t* to_internal, int32_t n_frames)
{
    if (!h || (to_internal && n_frames < 1)) return SRK_E_ARGS;
    if (!to_internal) { h->frame_order_given.clear(); return SRK_OK; }
    std::vector<char> hit((size_t)n_frames, 0);
    for (int32_t j = 0; j < n_frames; ++j) {
        if (to_internal[j] < 0 || to_internal[j] >= n_frames || hit[(size_t)to_internal[j]]) { h->last_error = "set_frame_order: not a permutation"; return SRK_E_ARGS; }
        hit[(size_t)to_internal[j]] = 1;
    }
    h->frame_order_given.assign(to_internal, to_internal + n_frames);
    return SRK_OK;
}
// 1 = the uploaded scene's frames are stored in another order (to_internal[caller's frame] filled when not NULL), 0 = the caller's order
int srk_ba_frame_order(srk_ba* h, int32_t* to_internal)
{
    if (!h || !h->have_scene) return SRK_E_STATE;
    for (int32_t j = 0; to_internal && j < h->d.M; ++j) to_internal[j] = h->frame_int.empty() ? j : h->frame_int[(size_t)j];
    return h->frame_int.empty() ? 0 : 1;
}
// the ordering alone (host only, no device needed): what srk_ba_upload_scene would decide for these tracks
int srk_frame_order(int mode, int64_t N, int32_t M, const int64_t* row_ptr, const int32_t* obs_frame, int32_t* to_internal)
{
    if (N < 0 || M < 1 || !row_ptr || (row_ptr[N] > 0 && !obs_frame) || !to_internal) return SRK_E_ARGS;
    std::vector<int32_t> to_int;
    const bool on = frame_reorder(mode, N, M, row_ptr, obs_frame, to_int);
    for (int32_t j = 0; j < M; ++j) to_internal[j] = on ? to_int[(size_t)j] : j;
    return on ? 1 : 0;
}

// -1 = automatic (run-based kernel when the runs of identical frame lists are long enough), 0 = per-observation kernels
// only, 1 = run-based (uniform runs) whenever the scene allows it, 2 = run-based over frame unions (the ragged-track form)
// whenever the scene allows it; takes effect at the next upload.  For A/B runs and for the parity tests of both kernels on the same scene.
int srk_ba_set_jacobian_mode(srk_ba* h, int mode)
{
    if (!h || mode < -1 || mode > 2) return SRK_E_ARGS;
    h->jac_mode = mode;
    return SRK_OK;
}
int srk_ba_jacobian_kernel(srk_ba* h) { return (h && h->have_scene) ? (h->jac_runs ? (h->jac_runs_masked ? 3 : 2) : (h->jac_fused ? 1 : 0)) : -1; }

// 0 = everything stored in fp64 (default, the reference's Scalar = double); 1 = the point-frame blocks W -- 240 of the
// 260 bytes per observation the derivative kernel writes and the Schur and back-substitution kernels read -- are stored
// as float and widened on load; every sum, the reduced camera system and the solve stay fp64.  Next upload.
int srk_ba_set_storage_precision(srk_ba* h, int f32)
{
    if (!h || (f32 != 0 && f32 != 1)) return SRK_E_ARGS;
    h->store_f32 = f32 != 0;
    return SRK_OK;
}

int srk_ba_set_schur_precision(srk_ba* h, int fp32)
{
    if (!h || (fp32 != 0 && fp32 != 1)) return SRK_E_ARGS;
    h->schur_fp32 = fp32 != 0;
    return SRK_OK;
}

// knob for bench.py: event pairs around every MFMA trailing-update launch (report.ms_solve_syrk)
int srk_ba_set_profile(srk_ba* h, int level)
{
    if (!h || level < 0 || level > 2) return SRK_E_ARGS;
    h->profile_level = level;
    return SRK_OK;
}

} // extern "C"
