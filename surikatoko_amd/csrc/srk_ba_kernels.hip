// srk_ba_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the Kanatani BA hot path.
//
// Reference being accelerated (whigg/surikatoko, cpp_impl/suriko-engine/src/bundle-adj-kanatani.cpp):
//   reprojection error                :410-490   -> k_error / k_error_final
//   closed-form derivatives           :1140-1549 -> k_jac_points (point-major) + k_jac_frames (frame-major)
//   reduced camera system (Schur)     :1771-1908 -> k_schur_mm (fp64 MFMA) / k_schur_grouped / k_schur + k_assemble
//   point back-substitution + apply   :1919-1960, :1997-2017 -> k_backsub_obs + k_point_update
//   camera apply (Rodrigues)          :2021-2062, :59-92     -> k_cam_apply
// All arithmetic is fp64 (reference Scalar = double).  Except for the Schur sum (compute bound: a matrix product per
// run of landmarks) these kernels are HBM-bound streaming passes over the observation arrays: every global access is lane-contiguous (SoA blocks), per-landmark sums are wavefront
// segmented reductions, per-frame sums are register accumulators + wavefront reductions + one atomic per block.
#include "srk_dev.hpp"
#include <type_traits>

#define WAVE 64

__device__ __forceinline__ bool srk_is_fixed_var(int64_t var, const SrkDims& d)
{
    // bundle-adj-kanatani.cpp:539-563: frame-local 4..9 of frame 0 and 14+comp (frame 1's 4+comp) are removed by the gauge;
    // d.g0, d.g1: where the caller's frames 0 and 1 sit in the internal frame order (0 and 1 unless the frames were reordered)
    const int64_t b0 = 10 * (int64_t)d.g0;
    return (var >= b0 + 4 && var <= b0 + 9) || (var == 10 * (int64_t)d.g1 + 4 + d.comp);
}

// ------------------------------------------------------------------ camera pack
// one camera's pack from its inverse pose (r, t) and intrinsics k; also called by k_cam_apply on the pose it has just made
__device__ __forceinline__ void cam_pack_one(const double* r, const double* t, const double* k, double f0, double* p)
{
    for (int i = 0; i < 9; ++i) { p[i] = r[i]; p[12 + i] = k[i]; }
    for (int i = 0; i < 3; ++i) p[9 + i] = t[i];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) p[21 + 3 * a + b] = k[3 * a] * r[b] + k[3 * a + 1] * r[3 + b] + k[3 * a + 2] * r[6 + b];
    // direct pose: Rd = R^T, Td = -(Rd T)   (obs-geom.cpp:117-122)
    double td[3];
    for (int c = 0; c < 3; ++c) td[c] = -(r[c] * t[0] + r[3 + c] * t[1] + r[6 + c] * t[2]);
    double fx = k[0], fy = k[4], u0 = k[2], v0 = k[5];
    for (int c = 0; c < 3; ++c) {
        p[30 + c] = td[c];
        // Rd[c][0] = R[0][c] etc.
        p[33 + c] = fx * r[c] + u0 * r[6 + c];
        p[36 + c] = fy * r[3 + c] + v0 * r[6 + c];
        p[39 + c] = f0 * r[6 + c];
    }
    p[42] = 1 / fx;
    p[43] = u0 / (f0 * fx);
    p[44] = 1 / fy;
    p[45] = v0 / (f0 * fy);
    p[46] = 1 / f0;
    p[47] = f0;
}
__global__ void k_cam_pack(int32_t M, const double* __restrict__ R, const double* __restrict__ T,
                           const double* __restrict__ K, double f0, double* __restrict__ pack)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    cam_pack_one(R + 9 * (int64_t)j, T + 3 * (int64_t)j, K + 9 * (int64_t)j, f0, pack + (int64_t)SRK_CAM_PACK * j);
}

void srk_launch_cam_pack(hipStream_t s, int32_t M, const double* R, const double* T, const double* K, double f0,
                         double* pack)
{
    hipLaunchKernelGGL(k_cam_pack, dim3((M + 63) / 64), dim3(64), 0, s, M, R, T, K, f0, pack);
}

// ------------------------------------------------------------------ storage of the point-frame blocks
// Every entry of an observation's 3 x 10 block is  W[pv][fv] = Ap[pv] Af[fv] + Bp[pv] Bf[fv]  (formula 9 with the A's and
// B's scaled by sqrt(2) / r^2): the block is a rank-2 product.  With fp64 storage (the default) the library keeps the FACTORS
// -- 3 + 3 + 8 + 7 = 21 doubles (Af[1] = Af[3] = Bf[0] = Bf[2] = 0 and Af[2] = Bf[3]) instead of 30 products, 168 instead
// of 240 bytes an observation for the derivative pass to write and the Schur and back-substitution passes to read -- as SoA
// planes F[k Os + o] (SRK_WF_* in srk_dev.hpp), and every consumer forms the entries it needs with two multiply-adds.
// The opt-in f32 storage mode keeps the same 21 factors as floats (84 bytes; arithmetic stays fp64: loads widen).
// plane of Af[fv] / Bf[fv], or -1 where the factor is structurally zero
__device__ __forceinline__ int srk_wf_af_plane(int fv) { return fv >= 4 ? SRK_WF_AF4 + fv - 4 : (fv == 0 ? SRK_WF_AF0 : (fv == 2 ? SRK_WF_G : -1)); }
__device__ __forceinline__ int srk_wf_bf_plane(int fv) { return fv >= 4 ? SRK_WF_BF4 + fv - 4 : (fv == 1 ? SRK_WF_BF1 : (fv == 3 ? SRK_WF_G : -1)); }
// entry k = 10 pv + fv of observation o, whatever the storage
template <typename WT> __device__ __forceinline__ double w_entry(const WT* __restrict__ W, int64_t Os, int64_t o, int k)
{
    const int pv = k / 10, fv = k - 10 * pv;
    const int pa = srk_wf_af_plane(fv), pb = srk_wf_bf_plane(fv);
    const double af = pa >= 0 ? (double)W[(int64_t)pa * Os + o] : 0.0, bf = pb >= 0 ? (double)W[(int64_t)pb * Os + o] : 0.0;
    return (double)W[(int64_t)(SRK_WF_AP + pv) * Os + o] * af + (double)W[(int64_t)(SRK_WF_BP + pv) * Os + o] * bf;
}
// the block of observation o from factors scaled so that W = Ap Af + Bp Bf
template <typename WT>
__device__ __forceinline__ void w_store(WT* __restrict__ W, int64_t Os, int64_t o, const double (&Ap)[3], const double (&Bp)[3],
                                        const double (&Af)[10], const double (&Bf)[10])
{
    WT* wp = W + o;
#pragma unroll
    for (int v = 0; v < 3; ++v) { wp[(int64_t)(SRK_WF_AP + v) * Os] = (WT)Ap[v]; wp[(int64_t)(SRK_WF_BP + v) * Os] = (WT)Bp[v]; }
    wp[(int64_t)SRK_WF_AF0 * Os] = (WT)Af[0];
    wp[(int64_t)SRK_WF_G * Os] = (WT)Af[2];
    wp[(int64_t)SRK_WF_BF1 * Os] = (WT)Bf[1];
#pragma unroll
    for (int v = 4; v < 10; ++v) { wp[(int64_t)(SRK_WF_AF4 + v - 4) * Os] = (WT)Af[v]; wp[(int64_t)(SRK_WF_BF4 + v - 4) * Os] = (WT)Bf[v]; }
}

// ------------------------------------------------------------------ per-observation geometry
struct ObsGeom {
    double p, q, r;
    double ex, ey;    // p/r - u/f0 , q/r - v/f0
    double s1, s2;    // 2/r^2 , 2/r^4
};

__device__ __forceinline__ void obs_pqr(const double* __restrict__ c, double X0, double X1, double X2, double u,
                                        double v, ObsGeom& g)
{
    double xc0 = c[0] * X0 + c[1] * X1 + c[2] * X2 + c[9];
    double xc1 = c[3] * X0 + c[4] * X1 + c[5] * X2 + c[10];
    double xc2 = c[6] * X0 + c[7] * X1 + c[8] * X2 + c[11];
    g.p = c[12] * xc0 + c[13] * xc1 + c[14] * xc2;
    g.q = c[15] * xc0 + c[16] * xc1 + c[17] * xc2;
    g.r = c[18] * xc0 + c[19] * xc1 + c[20] * xc2;
    // one IEEE reciprocal; p/r, q/r, 2/r^2, 2/r^4 and u/f0 become multiplies (<= 1-2 ulp from the divided forms)
    double ir = 1.0 / g.r;
    double if0 = c[46];
    g.ex = g.p * ir - u * if0;
    g.ey = g.q * ir - v * if0;
    double ir2 = ir * ir;
    g.s1 = 2 * ir2;
    g.s2 = g.s1 * ir2;
}

// A_v = r p'_v - p r'_v , B_v = r q'_v - q r'_v for the three landmark variables (:1450-1455)
__device__ __forceinline__ void point_ab(const double* __restrict__ c, const ObsGeom& g, double A[3], double B[3])
{
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        double gp = c[21 + v], gq = c[24 + v], gr = c[27 + v];
        A[v] = g.r * gp - g.p * gr;
        B[v] = g.r * gq - g.q * gr;
    }
}

// the ten frame variables [fx fy u0 v0 Tx Ty Tz Wx Wy Wz] (:1457-1525)
__device__ __forceinline__ void frame_ab(const double* __restrict__ c, const ObsGeom& g, double X0, double X1,
                                         double X2, double A[10], double B[10])
{
    double gp_fx = c[42] * g.p - c[43] * g.r;
    double gq_fy = c[44] * g.q - c[45] * g.r;
    double g_uv = c[46] * g.r;
    A[0] = g.r * gp_fx; B[0] = 0;
    A[1] = 0;           B[1] = g.r * gq_fy;
    A[2] = g.r * g_uv;  B[2] = 0;
    A[3] = 0;           B[3] = g.r * g_uv;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double gp = -c[33 + k], gq = -c[36 + k], gr = -c[39 + k];
        A[4 + k] = g.r * gp - g.p * gr;
        B[4 + k] = g.r * gq - g.q * gr;
    }
    double t0 = X0 - c[30], t1 = X1 - c[31], t2 = X2 - c[32];
    // cross(rot, t)
    double cp[3], cq[3], cr[3];
    cp[0] = c[34] * t2 - c[35] * t1; cp[1] = c[35] * t0 - c[33] * t2; cp[2] = c[33] * t1 - c[34] * t0;
    cq[0] = c[37] * t2 - c[38] * t1; cq[1] = c[38] * t0 - c[36] * t2; cq[2] = c[36] * t1 - c[37] * t0;
    cr[0] = c[40] * t2 - c[41] * t1; cr[1] = c[41] * t0 - c[39] * t2; cr[2] = c[39] * t1 - c[40] * t0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A[7 + k] = g.r * cp[k] - g.p * cr[k];
        B[7 + k] = g.r * cq[k] - g.q * cr[k];
    }
}

// ------------------------------------------------------------------ K2a: point-major Jacobian pass
// One thread per observation.  Writes the 3x10 point-frame block (SoA, lane-contiguous 8-byte stores) and
// reduces the point block V (6 unique) + point gradient (3) over the landmark's observations with a
// wavefront segmented reduction; one atomic per (wave, landmark) segment.
template <typename WT> // storage type of the point-frame blocks W: double, or float (srk_ba_set_storage_precision)
__global__ __launch_bounds__(256) void k_jac_points(SrkDims d, const double* __restrict__ pts,
                                                    const double* __restrict__ cam,
                                                    const int32_t* __restrict__ obs_frame,
                                                    const int32_t* __restrict__ obs_pt,
                                                    const double* __restrict__ obs_uv, WT* __restrict__ W,
                                                    double* __restrict__ Vg)
{
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int lane = threadIdx.x & (WAVE - 1);
    bool valid = o < d.O;
    int32_t pt = -1;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0;
    if (valid) {
        pt = obs_pt[o];
        int32_t j = obs_frame[o];
        double2 uv = reinterpret_cast<const double2*>(obs_uv)[o];
        const double* X = pts + 3 * (int64_t)pt;
        double X0 = X[0], X1 = X[1], X2 = X[2];
        const double* c = cam + (int64_t)SRK_CAM_PACK * j;
        ObsGeom g;
        obs_pqr(c, X0, X1, X2, uv.x, uv.y, g);
        double Ap[3], Bp[3], Af[10], Bf[10];
        point_ab(c, g, Ap, Bp);
        frame_ab(c, g, X0, X1, X2, Af, Bf);
        {
            const double sc = 0.7071067811865476 * g.s1; // sqrt(2 / r^4): both sides of every product carry it once
            double Aps[3], Bps[3], Afs[10], Bfs[10];
#pragma unroll
            for (int v = 0; v < 3; ++v) { Aps[v] = Ap[v] * sc; Bps[v] = Bp[v] * sc; }
#pragma unroll
            for (int v = 0; v < 10; ++v) { Afs[v] = Af[v] * sc; Bfs[v] = Bf[v] * sc; }
            w_store<WT>(W, d.Os, o, Aps, Bps, Afs, Bfs);
        }
        acc[0] = (Ap[0] * Ap[0] + Bp[0] * Bp[0]) * g.s2;
        acc[1] = (Ap[0] * Ap[1] + Bp[0] * Bp[1]) * g.s2;
        acc[2] = (Ap[0] * Ap[2] + Bp[0] * Bp[2]) * g.s2;
        acc[3] = (Ap[1] * Ap[1] + Bp[1] * Bp[1]) * g.s2;
        acc[4] = (Ap[1] * Ap[2] + Bp[1] * Bp[2]) * g.s2;
        acc[5] = (Ap[2] * Ap[2] + Bp[2] * Bp[2]) * g.s2;
        acc[6] = (g.ex * Ap[0] + g.ey * Bp[0]) * g.s1;
        acc[7] = (g.ex * Ap[1] + g.ey * Bp[1]) * g.s1;
        acc[8] = (g.ex * Ap[2] + g.ey * Bp[2]) * g.s1;
    }
    // segmented (by landmark) suffix sums inside the wave: observations of one landmark are contiguous
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        int32_t okey = __shfl_down(pt, off, WAVE);
        bool take = (lane + off < WAVE) && (okey == pt);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            double other = __shfl_down(acc[k], off, WAVE);
            if (take) acc[k] += other;
        }
    }
    int32_t prev = __shfl_up(pt, 1, WAVE);
    bool head = (lane == 0) || (prev != pt);
    if (head && pt >= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(&Vg[(int64_t)k * d.Ns + pt], acc[k]);
    }
}

// ------------------------------------------------------------------ K2 fused: one pass over the observations
// Same per-observation work as k_jac_points, plus the frame blocks: because landmarks are stored sorted by frame
// list, the SRK_JF_OBS consecutive observations of a workgroup touch a narrow range of frames; their 55 + 10 frame
// entries are summed with LDS atomics into per-frame slots (direct-mapped on frame - jmin) and leave the chip once
// per workgroup.  The host only selects this kernel when every workgroup's frame range fits SRK_JF_SLOTS.
#define SRK_JF_CHUNKS 4
#define SRK_JF_OBS (256 * SRK_JF_CHUNKS)
#define SRK_JF_SLOTS 48

#define SRK_JF_PMAX 448 // landmarks per workgroup held in LDS (host falls back to the two-kernel path beyond)

__device__ __forceinline__ double jf_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

template <typename WT>
__global__ __launch_bounds__(256) void k_jac_fused(SrkDims d, const double* __restrict__ pts,
                                                   const double* __restrict__ cam,
                                                   const int32_t* __restrict__ obs_frame,
                                                   const int32_t* __restrict__ obs_pt,
                                                   const double* __restrict__ obs_uv, WT* __restrict__ W,
                                                   double* __restrict__ Vg, double* __restrict__ Ug,
                                                   const int32_t* __restrict__ wg_jmin)
{
    __shared__ double sU[SRK_JF_SLOTS][SRK_UG + 1];                              // frame blocks + frame gradients
    __shared__ __attribute__((aligned(16))) double sCam[SRK_JF_SLOTS][SRK_CAM_PACK]; // camera packs of the frame range
    __shared__ double sV[9][SRK_JF_PMAX];                                        // point blocks + point gradients
    __shared__ int sTouched[SRK_JF_SLOTS];
    const int lane = threadIdx.x & (WAVE - 1);
    const int jmin = wg_jmin[blockIdx.x];
    const int64_t o_first = (int64_t)blockIdx.x * SRK_JF_OBS;
    const int64_t o_last = (o_first + SRK_JF_OBS < d.O ? o_first + SRK_JF_OBS : d.O) - 1;
    const int32_t pmin = obs_pt[o_first], pmax = obs_pt[o_last];
    const int npts = pmax - pmin + 1;
    for (int t = threadIdx.x; t < SRK_JF_SLOTS * (SRK_UG + 1); t += 256) (&sU[0][0])[t] = 0.0;
    for (int t = threadIdx.x; t < 9 * SRK_JF_PMAX; t += 256) (&sV[0][0])[t] = 0.0;
    if (threadIdx.x < SRK_JF_SLOTS) sTouched[threadIdx.x] = 0;
    {
        int nfr = d.M - jmin < SRK_JF_SLOTS ? d.M - jmin : SRK_JF_SLOTS;
        const double* src = cam + (int64_t)SRK_CAM_PACK * jmin;
        for (int t = threadIdx.x; t < nfr * SRK_CAM_PACK; t += 256) (&sCam[0][0])[t] = src[t];
    }
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < SRK_JF_CHUNKS; ++ch) {
        const int64_t o = o_first + ch * 256 + threadIdx.x;
        const bool valid = o < d.O;
        int32_t pt = -1;
        double acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0;
        if (valid) {
            pt = obs_pt[o];
            const int js = obs_frame[o] - jmin;
            double2 uv = reinterpret_cast<const double2*>(obs_uv)[o];
            const double* X = pts + 3 * (int64_t)pt;
            const double X0 = X[0], X1 = X[1], X2 = X[2];
            const double* c = sCam[js];
            // (p, q, r) = K (R X + T)   (:469-470)
            const double xc0 = c[0] * X0 + c[1] * X1 + c[2] * X2 + c[9];
            const double xc1 = c[3] * X0 + c[4] * X1 + c[5] * X2 + c[10];
            const double xc2 = c[6] * X0 + c[7] * X1 + c[8] * X2 + c[11];
            const double p = c[12] * xc0 + c[13] * xc1 + c[14] * xc2;
            const double q = c[15] * xc0 + c[16] * xc1 + c[17] * xc2;
            const double r = c[18] * xc0 + c[19] * xc1 + c[20] * xc2;
            const double ir = jf_rcp(r), ir2 = ir * ir;
            const double s1 = 2 * ir2, s2 = s1 * ir2; // 2 / r^2 , 2 / r^4
            const double ex1 = (p * ir - uv.x * c[46]) * s1, ey1 = (q * ir - uv.y * c[46]) * s1;
            // A_v = r p'_v - p r'_v , B_v = r q'_v - q r'_v
            double Ap[3], Bp[3], Af[10], Bf[10];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                Ap[v] = r * c[21 + v] - p * c[27 + v];
                Bp[v] = r * c[24 + v] - q * c[27 + v];
            }
            const double g_uv = c[46] * r * r;
            Af[0] = r * (c[42] * p - c[43] * r); Bf[0] = 0;
            Af[1] = 0;                           Bf[1] = r * (c[44] * q - c[45] * r);
            Af[2] = g_uv;                        Bf[2] = 0;
            Af[3] = 0;                           Bf[3] = g_uv;
            // translation and rotation columns share a1 = r rot1 - p rot3, b1 = r rot2 - q rot3 :
            //   d/dT = -(a1, b1)   (:1503-1505)      d/dW = (a1 x t, b1 x t), t = X - T_direct   (:1511-1520)
            double a1[3], b1[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a1[k] = r * c[33 + k] - p * c[39 + k];
                b1[k] = r * c[36 + k] - q * c[39 + k];
                Af[4 + k] = -a1[k];
                Bf[4 + k] = -b1[k];
            }
            const double t0 = X0 - c[30], t1 = X1 - c[31], t2 = X2 - c[32];
            Af[7] = a1[1] * t2 - a1[2] * t1; Af[8] = a1[2] * t0 - a1[0] * t2; Af[9] = a1[0] * t1 - a1[1] * t0;
            Bf[7] = b1[1] * t2 - b1[2] * t1; Bf[8] = b1[2] * t0 - b1[0] * t2; Bf[9] = b1[0] * t1 - b1[1] * t0;
            // fold 2/r^4 into one side of every product once
            double Aps[3], Bps[3], Afs[10], Bfs[10];
#pragma unroll
            for (int v = 0; v < 3; ++v) { Aps[v] = Ap[v] * s2; Bps[v] = Bp[v] * s2; }
#pragma unroll
            for (int v = 0; v < 10; ++v) { Afs[v] = Af[v] * s2; Bfs[v] = Bf[v] * s2; }
            {
                const double sc = sqrt(s2); // both sides of every product carry sqrt(2 / r^4) once
                double Apq[3], Bpq[3], Afq[10], Bfq[10];
#pragma unroll
                for (int v = 0; v < 3; ++v) { Apq[v] = Ap[v] * sc; Bpq[v] = Bp[v] * sc; }
#pragma unroll
                for (int v = 0; v < 10; ++v) { Afq[v] = Af[v] * sc; Bfq[v] = Bf[v] * sc; }
                w_store<WT>(W, d.Os, o, Apq, Bpq, Afq, Bfq);
            }
            acc[0] = Aps[0] * Ap[0] + Bps[0] * Bp[0];
            acc[1] = Aps[0] * Ap[1] + Bps[0] * Bp[1];
            acc[2] = Aps[0] * Ap[2] + Bps[0] * Bp[2];
            acc[3] = Aps[1] * Ap[1] + Bps[1] * Bp[1];
            acc[4] = Aps[1] * Ap[2] + Bps[1] * Bp[2];
            acc[5] = Aps[2] * Ap[2] + Bps[2] * Bp[2];
            acc[6] = ex1 * Ap[0] + ey1 * Bp[0];
            acc[7] = ex1 * Ap[1] + ey1 * Bp[1];
            acc[8] = ex1 * Ap[2] + ey1 * Bp[2];
            // frame block (55 unique) + frame gradient (10) -> LDS slot of this frame (few lanes share a frame)
            double* su = sU[js];
            sTouched[js] = 1;
            int idx = 0;
#pragma unroll
            for (int v1 = 0; v1 < 10; ++v1)
#pragma unroll
                for (int v2 = v1; v2 < 10; ++v2) {
                    atomicAdd(&su[idx], Afs[v1] * Af[v2] + Bfs[v1] * Bf[v2]);
                    ++idx;
                }
#pragma unroll
            for (int v = 0; v < 10; ++v) atomicAdd(&su[55 + v], ex1 * Af[v] + ey1 * Bf[v]);
        }
        // point block + gradient: wavefront segmented reduction over the landmark's (contiguous) observations, then
        // one LDS add per (wave, landmark) -- 20 lanes adding to one LDS word would serialise (measured slower)
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            int32_t okey = __shfl_down(pt, off, WAVE);
            bool take = (lane + off < WAVE) && (okey == pt);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                double other = __shfl_down(acc[k], off, WAVE);
                if (take) acc[k] += other;
            }
        }
        int32_t prev = __shfl_up(pt, 1, WAVE);
        if (((lane == 0) || (prev != pt)) && pt >= 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) atomicAdd(&sV[k][pt - pmin], acc[k]);
        }
    }
    __syncthreads();
    // landmarks strictly inside the workgroup's range are complete: plain coalesced stores; the first and the last
    // may continue in the neighbouring workgroups: atomics
    for (int t = threadIdx.x; t < 9 * npts; t += 256) {
        int k = t / npts, ps = t - k * npts;
        double v = sV[k][ps];
        double* dst = &Vg[(int64_t)k * d.Ns + pmin + ps];
        if (ps == 0 || ps == npts - 1) atomicAdd(dst, v);
        else *dst = v;
    }
    for (int t = threadIdx.x; t < SRK_JF_SLOTS * SRK_UG; t += 256) {
        int slot = t / SRK_UG, k = t - slot * SRK_UG;
        if (sTouched[slot]) atomicAdd(&Ug[(int64_t)(jmin + slot) * SRK_UG + k], sU[slot][k]);
    }
}

void srk_launch_jac_fused(hipStream_t s, const SrkDims& d, const double* pts, const double* cam,
                          const int32_t* obs_frame, const int32_t* obs_pt, const double* obs_uv, double* W,
                          double* Vg, double* Ug, const int32_t* wg_jmin)
{
    if (d.O == 0) return;
    int64_t blocks = (d.O + SRK_JF_OBS - 1) / SRK_JF_OBS;
    if (d.w_f32)
        hipLaunchKernelGGL(k_jac_fused<float>, dim3((unsigned)blocks), dim3(256), 0, s, d, pts, cam, obs_frame, obs_pt, obs_uv,
                           reinterpret_cast<float*>(W), Vg, Ug, wg_jmin);
    else
        hipLaunchKernelGGL(k_jac_fused<double>, dim3((unsigned)blocks), dim3(256), 0, s, d, pts, cam, obs_frame, obs_pt, obs_uv,
                           W, Vg, Ug, wg_jmin);
}

// ------------------------------------------------------------------ K2 by runs: a lane keeps ONE frame for a whole task
// Landmarks are stored sorted by frame list, so landmarks with IDENTICAL lists form runs.  A task = a piece of one
// run (nf frames each; the host sizes the pieces so that about one task per resident wave exists); one wave per task.  Lane l works on frame
// f = l % nf of landmark m = l / nf of the current iteration (g = 64 / nf landmarks per iteration, g nf <= 64 active
// lanes), so across the task's iterations a lane stays on ONE frame:
//   * the 55 + 10 frame-block / frame-gradient sums live in the lane's registers and go to the workgroup's LDS slot of
//     that frame once per task (k_jac_fused: 65 LDS atomics per OBSERVATION);
//   * a landmark's nf observations all sit in one iteration of one wave: its point block and gradient are summed in
//     a fixed order from a per-wave LDS scratch and stored plainly (no atomics, no 64-lane segmented scan);
//   * the W stores stay lane-contiguous (observations o_base + it g nf + lane).
// The host uses this kernel when the tasks are long enough to pay (circle-grid scenes: ~100 landmarks per run).
// Measured on C3 (2 M observations, 20 frames a landmark): 119 us a launch against 178 us for k_jac_fused; without the
// W stores it takes 57 us, the stores alone 87 us -- what is left is overlap between the two at two waves per SIMD
// (61 frame sums + 26 derivative values = 230 registers).  Tried and dropped (round 2, each measured): W stored from a
// second, 64-aligned walk over the observations (whole 512-byte rows instead of 60-lane rows that start off a line
// boundary; the derivatives formed twice): 123 us; the same as straight-line code with buffer stores (masked lanes
// dropped by the range check) so that the compiler counts the stores behind a load instead of waiting with vmcnt(0):
// 177 us at two waves per SIMD, 194 us at one -- like the ablation without the frame sums (161 us), every variant that
// lets the waves issue their stores FASTER ran slower, the 60 k concurrent 512-byte write streams (2048 waves x 30
// planes) evidently need the pacing; non-temporal stores: +40 us; the four waves of a workgroup interleaved over one
// task (neighbouring pieces written together): +5 us.
#define SRK_JR_RSTRIDE 66 // doubles between the 9 planes of the per-wave reduction scratch (64 lanes + bank skew)

// MASKED (round 3): the task is a piece of a run over the UNION of its landmarks' frame lists (the runs of the Schur kernel:
// grp_*, pt_mask) -- ragged feature tracks, where hardly two landmarks see exactly the same frames.  Lane (m, f) owns the CELL
// (landmark m of the step, frame slot f of the union): the cell's observation is the landmark's first one plus the number of
// mask bits below the slot; a cell the landmark does not see computes nothing and adds zeros to the landmark's sums.  A lane
// still stays on one frame for the whole task.  The landmarks' first observations and masks sit in a per-wave LDS table.
template <typename WT, bool MASKED, bool DET = false>
__global__ __launch_bounds__(256, 2) void k_jac_runs(SrkDims d, const double* __restrict__ pts,
                                                     const double* __restrict__ cam,
                                                     const int64_t* __restrict__ row_ptr,
                                                     const int32_t* __restrict__ obs_frame,
                                                     const double* __restrict__ obs_uv, WT* __restrict__ W,
                                                     double* __restrict__ Vg, double* __restrict__ Ug,
                                                     const int32_t* __restrict__ task_first,
                                                     const int32_t* __restrict__ task_count, int32_t n_tasks,
                                                     const int32_t* __restrict__ wg_jmin,
                                                     const int32_t* __restrict__ task_group, const int32_t* __restrict__ grp_nf,
                                                     const int32_t* __restrict__ grp_frames, const uint32_t* __restrict__ pt_mask,
                                                     double* __restrict__ det_stage /* DET: [task][64][SRK_UG] */, int frames_stride)
{
    __shared__ int32_t sTOff[MASKED ? 4 : 1][MASKED ? SRK_JR_TASK_PTS_MAX_HOST : 1];
    __shared__ uint32_t sTMask[MASKED ? 4 : 1][MASKED ? SRK_JR_TASK_PTS_MAX_HOST : 1];
    __shared__ double sU[SRK_JF_SLOTS][SRK_UG + 1];                                  // frame blocks + frame gradients
    __shared__ __attribute__((aligned(16))) double sCam[SRK_JF_SLOTS][SRK_CAM_PACK]; // camera packs of the frame window
    __shared__ double sR[4][9 * SRK_JR_RSTRIDE];                                     // per-wave landmark reduction
    __shared__ int sTouched[SRK_JF_SLOTS];
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    const int jmin = wg_jmin[blockIdx.x];
    for (int t = threadIdx.x; t < SRK_JF_SLOTS * (SRK_UG + 1); t += 256) (&sU[0][0])[t] = 0.0;
    if (threadIdx.x < SRK_JF_SLOTS) sTouched[threadIdx.x] = 0;
    {
        int nfr = d.M - jmin < SRK_JF_SLOTS ? d.M - jmin : SRK_JF_SLOTS;
        const double* src = cam + (int64_t)SRK_CAM_PACK * jmin;
        for (int t = threadIdx.x; t < nfr * SRK_CAM_PACK; t += 256) (&sCam[0][0])[t] = src[t];
    }
    __syncthreads();
    {
    const int task = blockIdx.x * 4 + wv;
    const int step_first = 0, step_stride = 1;
    if (task < n_tasks) {
        const int32_t first_pt = task_first[task], n_pts = task_count[task];
        const int64_t o_base = row_ptr[first_pt];
        int nf = (int)(row_ptr[first_pt + 1] - o_base);
        const int32_t* fr = nullptr; // (MASKED) the run's frame set
        if constexpr (MASKED) {
            const int gi = task_group[task];
            nf = grp_nf[gi] < 0 ? -grp_nf[gi] : grp_nf[gi];
            fr = grp_frames + (int64_t)gi * frames_stride;
            for (int l = lane; l < n_pts; l += WAVE) {
                sTOff[wv][l] = (int32_t)(row_ptr[first_pt + l] - o_base);
                sTMask[wv][l] = pt_mask[first_pt + l];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (one wave: its LDS operations execute in order)
        }
        const int g = WAVE / nf, active = g * nf;
        const int m = lane / nf, f = lane - m * nf;
        const bool on = lane < active;
        const int js = on ? (MASKED ? fr[f] : obs_frame[o_base + f]) - jmin : 0;
        // the observation of cell (landmark il of the task, this lane's slot): MASKED -> from the table, else il nf + f
        auto cell = [&](int il, bool& seen) -> int64_t {
            if constexpr (MASKED) {
                const uint32_t mk = sTMask[wv][il];
                seen = (mk >> f) & 1u;
                return o_base + sTOff[wv][il] + __builtin_popcount(mk & ((1u << f) - 1u));
            } else {
                seen = true;
                return o_base + (int64_t)il * nf + f;
            }
        };
        const double* c = sCam[js];
        double* sr = sR[wv];
        double acc[SRK_UG];
#pragma unroll
        for (int k = 0; k < SRK_UG; ++k) acc[k] = 0;
        // software pipeline: the next iteration's observation and landmark are loaded while this one is computed
        double2 uv_n = make_double2(0, 0);
        double Xn0 = 0, Xn1 = 0, Xn2 = 0;
        bool seen_n = false;
        int64_t o_n = 0;
        if (on && step_first * g + m < n_pts) {
            o_n = cell(step_first * g + m, seen_n);
            if (seen_n) uv_n = reinterpret_cast<const double2*>(obs_uv)[o_n];
            const double* X = pts + 3 * (int64_t)(first_pt + step_first * g + m);
            Xn0 = X[0]; Xn1 = X[1]; Xn2 = X[2];
        }
#pragma unroll 1
        for (int i0 = step_first * g; i0 < n_pts; i0 += step_stride * g) {
            const int il = i0 + m;
            const bool valid = on && il < n_pts && seen_n;
            const int64_t o = o_n;
            const double2 uv = uv_n;
            const double X0 = Xn0, X1 = Xn1, X2 = Xn2;
            seen_n = false;
            if (on && il + step_stride * g < n_pts) {
                o_n = cell(il + step_stride * g, seen_n);
                if (seen_n) uv_n = reinterpret_cast<const double2*>(obs_uv)[o_n];
                const double* X = pts + 3 * (int64_t)(first_pt + il + step_stride * g);
                Xn0 = X[0]; Xn1 = X[1]; Xn2 = X[2];
            }
            double v9[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) v9[k] = 0;
            if (valid) {
                // (p, q, r) = K (R X + T)   (:469-470)
                const double xc0 = c[0] * X0 + c[1] * X1 + c[2] * X2 + c[9];
                const double xc1 = c[3] * X0 + c[4] * X1 + c[5] * X2 + c[10];
                const double xc2 = c[6] * X0 + c[7] * X1 + c[8] * X2 + c[11];
                const double p = c[12] * xc0 + c[13] * xc1 + c[14] * xc2;
                const double q = c[15] * xc0 + c[16] * xc1 + c[17] * xc2;
                const double r = c[18] * xc0 + c[19] * xc1 + c[20] * xc2;
                const double ir = jf_rcp(r), ir2 = ir * ir;
                // Every entry is (2 / r^4) (A A' + B B') (formula 9) or (2 / r^2) (ex A + ey B) (formula 8): with the
                // A's and B's scaled by sqrt(2) / r^2 once, the blocks are plain products and the gradient terms carry
                // sqrt(2) ex, sqrt(2) ey (no second, pre-scaled copy of the 26 values: they would not fit the registers
                // beside the 65 frame sums).
                const double sc = 1.4142135623730951 * ir2;
                const double exs = (p * ir - uv.x * c[46]) * 1.4142135623730951, eys = (q * ir - uv.y * c[46]) * 1.4142135623730951;
                const double rs = r * sc, ps = p * sc, qs = q * sc;
                double Ap[3], Bp[3], Af[10], Bf[10];
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    Ap[v] = rs * c[21 + v] - ps * c[27 + v];
                    Bp[v] = rs * c[24 + v] - qs * c[27 + v];
                }
                const double g_uv = c[46] * r * rs;
                Af[0] = rs * (c[42] * p - c[43] * r); Bf[0] = 0;
                Af[1] = 0;                            Bf[1] = rs * (c[44] * q - c[45] * r);
                Af[2] = g_uv;                         Bf[2] = 0;
                Af[3] = 0;                            Bf[3] = g_uv;
                // translation and rotation columns share a1 = r rot1 - p rot3, b1 = r rot2 - q rot3 :
                //   d/dT = -(a1, b1)   (:1503-1505)      d/dW = (a1 x t, b1 x t), t = X - T_direct   (:1511-1520)
                double a1[3], b1[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    a1[k] = rs * c[33 + k] - ps * c[39 + k];
                    b1[k] = rs * c[36 + k] - qs * c[39 + k];
                    Af[4 + k] = -a1[k];
                    Bf[4 + k] = -b1[k];
                }
                const double t0 = X0 - c[30], t1 = X1 - c[31], t2 = X2 - c[32];
                Af[7] = a1[1] * t2 - a1[2] * t1; Af[8] = a1[2] * t0 - a1[0] * t2; Af[9] = a1[0] * t1 - a1[1] * t0;
                Bf[7] = b1[1] * t2 - b1[2] * t1; Bf[8] = b1[2] * t0 - b1[0] * t2; Bf[9] = b1[0] * t1 - b1[1] * t0;
                w_store<WT>(W, d.Os, o, Ap, Bp, Af, Bf); // the 21 factors (as floats in the f32 storage mode)
                v9[0] = Ap[0] * Ap[0] + Bp[0] * Bp[0];
                v9[1] = Ap[0] * Ap[1] + Bp[0] * Bp[1];
                v9[2] = Ap[0] * Ap[2] + Bp[0] * Bp[2];
                v9[3] = Ap[1] * Ap[1] + Bp[1] * Bp[1];
                v9[4] = Ap[1] * Ap[2] + Bp[1] * Bp[2];
                v9[5] = Ap[2] * Ap[2] + Bp[2] * Bp[2];
                v9[6] = exs * Ap[0] + eys * Bp[0];
                v9[7] = exs * Ap[1] + eys * Bp[1];
                v9[8] = exs * Ap[2] + eys * Bp[2];
                int idx = 0;
#pragma unroll
                for (int v1 = 0; v1 < 10; ++v1)
#pragma unroll
                    for (int v2 = v1; v2 < 10; ++v2) {
                        acc[idx] = fma(Af[v1], Af[v2], fma(Bf[v1], Bf[v2], acc[idx]));
                        ++idx;
                    }
#pragma unroll
                for (int v = 0; v < 10; ++v) acc[55 + v] = fma(exs, Af[v], fma(eys, Bf[v], acc[55 + v]));
            }
            // point block + gradient of the g landmarks of this iteration: every lane parks its nine terms in the wave's
            // scratch, lane (k, mm) adds landmark mm's nf terms of entry k in frame order and stores the sum.  LDS
            // operations of one wave execute in order: no barrier and no wait, only the compiler must keep the order.
#pragma unroll
            for (int k = 0; k < 9; ++k) sr[k * SRK_JR_RSTRIDE + lane] = v9[k];
            asm volatile("" ::: "memory");
            for (int rr = lane; rr < 9 * g; rr += WAVE) {
                const int k = rr / g, mm = rr - k * g;
                if (i0 + mm < n_pts) {
                    const double* src = sr + k * SRK_JR_RSTRIDE + mm * nf;
                    // four independent partial sums (fixed order): the LDS reads of a group are in flight together
                    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                    int ff = 0;
                    for (; ff + 4 <= nf; ff += 4) {
                        s0 += src[ff];
                        s1 += src[ff + 1];
                        s2 += src[ff + 2];
                        s3 += src[ff + 3];
                    }
                    for (; ff < nf; ++ff) s0 += src[ff];
                    Vg[(int64_t)k * d.Ns + first_pt + i0 + mm] = (s0 + s1) + (s2 + s3);
                }
            }
            asm volatile("" ::: "memory");
        }
        if (DET) {
            // deterministic mode (srk_ba_set_deterministic): no atomics.  The g lanes of a frame (one per landmark of a step)
            // are summed in lane order, the first one stores the task's 65 sums of that frame; k_jac_det_gather adds the
            // tasks' sums per frame in task order.
            for (int m2 = 1; m2 < g; ++m2) {
#pragma unroll
                for (int k = 0; k < SRK_UG; ++k) {
                    const double o = __shfl(acc[k], f + m2 * nf, WAVE);
                    if (m == 0) acc[k] += o; // (the other lanes keep their own sums: they are read in the steps behind)
                }
            }
            if (on && m == 0) {
                double* dst = det_stage + ((int64_t)task * WAVE + f) * SRK_UG;
#pragma unroll
                for (int k = 0; k < SRK_UG; ++k) dst[k] = acc[k];
            }
        } else if (on) {
            double* su = sU[js];
            sTouched[js] = 1;
#pragma unroll
            for (int k = 0; k < SRK_UG; ++k) atomicAdd(&su[k], acc[k]);
        }
    }
    }
    if (DET) return;
    __syncthreads();
    for (int t = threadIdx.x; t < SRK_JF_SLOTS * SRK_UG; t += 256) {
        int slot = t / SRK_UG, k = t - slot * SRK_UG;
        if (sTouched[slot]) atomicAdd(&Ug[(int64_t)(jmin + slot) * SRK_UG + k], sU[slot][k]);
    }
}

// deterministic mode: frame j's block and gradient = the tasks' sums of that frame, added in task order
__global__ __launch_bounds__(128) void k_jac_det_gather(int32_t M, const int32_t* __restrict__ ptr, const int32_t* __restrict__ ent,
                                                        const double* __restrict__ stage, double* __restrict__ Ug)
{
    const int j = blockIdx.x, k = threadIdx.x;
    if (j >= M || k >= SRK_UG) return;
    double sum = 0;
    for (int e = ptr[j]; e < ptr[j + 1]; ++e) sum += stage[(int64_t)ent[e] * SRK_UG + k];
    Ug[(int64_t)j * SRK_UG + k] = sum;
}

void srk_launch_jac_runs(hipStream_t s, const SrkDims& d, const double* pts, const double* cam, const int64_t* row_ptr,
                         const int32_t* obs_frame, const double* obs_uv, double* W, double* Vg, double* Ug,
                         const int32_t* task_first, const int32_t* task_count, int32_t n_tasks, const int32_t* wg_jmin,
                         const int32_t* task_group, const int32_t* grp_nf, const int32_t* grp_frames, const uint32_t* pt_mask,
                         const SrkDetJac* det, int frames_stride)
{
    if (n_tasks <= 0) return;
    const dim3 grid((unsigned)((n_tasks + 3) / 4));
#define SRK_JR_ARGS(WP) d, pts, cam, row_ptr, obs_frame, obs_uv, WP, Vg, Ug, task_first, task_count, n_tasks, wg_jmin, task_group, grp_nf, grp_frames, pt_mask, det ? det->stage : nullptr, frames_stride
#define SRK_JR_LAUNCH(MASKED)                                                                                                         \
    do {                                                                                                                              \
        if (det) {                                                                                                                    \
            if (d.w_f32) hipLaunchKernelGGL((k_jac_runs<float, MASKED, true>), grid, dim3(256), 0, s, SRK_JR_ARGS(reinterpret_cast<float*>(W))); \
            else hipLaunchKernelGGL((k_jac_runs<double, MASKED, true>), grid, dim3(256), 0, s, SRK_JR_ARGS(W));                          \
        } else {                                                                                                                      \
            if (d.w_f32) hipLaunchKernelGGL((k_jac_runs<float, MASKED, false>), grid, dim3(256), 0, s, SRK_JR_ARGS(reinterpret_cast<float*>(W))); \
            else hipLaunchKernelGGL((k_jac_runs<double, MASKED, false>), grid, dim3(256), 0, s, SRK_JR_ARGS(W));                         \
        }                                                                                                                             \
    } while (0)
    if (task_group) SRK_JR_LAUNCH(true); // tasks over unions of frame lists (ragged tracks)
    else SRK_JR_LAUNCH(false);
#undef SRK_JR_LAUNCH
#undef SRK_JR_ARGS
    if (det) hipLaunchKernelGGL(k_jac_det_gather, dim3((unsigned)d.M), dim3(128), 0, s, d.M, det->ptr, det->ent, det->stage, Ug);
}

void srk_launch_jac_points(hipStream_t s, const SrkDims& d, const double* pts, const double* cam,
                           const int32_t* obs_frame, const int32_t* obs_pt, const double* obs_uv, double* W,
                           double* Vg)
{
    if (d.O == 0) return;
    int64_t blocks = (d.O + 255) / 256;
    if (d.w_f32)
        hipLaunchKernelGGL(k_jac_points<float>, dim3((unsigned)blocks), dim3(256), 0, s, d, pts, cam, obs_frame, obs_pt, obs_uv,
                           reinterpret_cast<float*>(W), Vg);
    else
        hipLaunchKernelGGL(k_jac_points<double>, dim3((unsigned)blocks), dim3(256), 0, s, d, pts, cam, obs_frame, obs_pt, obs_uv,
                           W, Vg);
}

// ------------------------------------------------------------------ K2b: frame-major Jacobian pass
// blockIdx.y = frame, blockIdx.x = chunk of SRK_FCHUNK observations of that frame.  Each thread keeps the 55
// unique entries of the 10x10 frame block + the 10 gradient entries in registers, then wave-reduces.
#define SRK_FCHUNK 1024

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

__global__ __launch_bounds__(256) void k_jac_frames(SrkDims d, const double* __restrict__ pts,
                                                    const double* __restrict__ cam,
                                                    const int64_t* __restrict__ col_ptr,
                                                    const int32_t* __restrict__ fobs_pt,
                                                    const double* __restrict__ fobs_uv, double* __restrict__ Ug)
{
    __shared__ double red[4][SRK_UG];
    int j = blockIdx.y;
    int64_t begin = col_ptr[j] + (int64_t)blockIdx.x * SRK_FCHUNK;
    int64_t end = col_ptr[j + 1];
    if (begin >= end) return;
    if (end > begin + SRK_FCHUNK) end = begin + SRK_FCHUNK;
    const double* c = cam + (int64_t)SRK_CAM_PACK * j;
    double acc[SRK_UG];
#pragma unroll
    for (int k = 0; k < SRK_UG; ++k) acc[k] = 0;
    for (int64_t k = begin + threadIdx.x; k < end; k += 256) {
        int32_t pt = fobs_pt[k];
        double2 uv = reinterpret_cast<const double2*>(fobs_uv)[k];
        const double* X = pts + 3 * (int64_t)pt;
        double X0 = X[0], X1 = X[1], X2 = X[2];
        ObsGeom g;
        obs_pqr(c, X0, X1, X2, uv.x, uv.y, g);
        double Af[10], Bf[10];
        frame_ab(c, g, X0, X1, X2, Af, Bf);
        int idx = 0;
#pragma unroll
        for (int v1 = 0; v1 < 10; ++v1)
#pragma unroll
            for (int v2 = v1; v2 < 10; ++v2) {
                acc[idx] += (Af[v1] * Af[v2] + Bf[v1] * Bf[v2]) * g.s2;
                ++idx;
            }
#pragma unroll
        for (int v = 0; v < 10; ++v) acc[55 + v] += (g.ex * Af[v] + g.ey * Bf[v]) * g.s1;
    }
    int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < SRK_UG; ++k) {
        double v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < SRK_UG) {
        double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(&Ug[(int64_t)j * SRK_UG + threadIdx.x], v);
    }
}

void srk_launch_jac_frames(hipStream_t s, const SrkDims& d, int64_t max_frame_obs, const double* pts,
                           const double* cam, const int64_t* col_ptr, const int32_t* fobs_pt, const double* fobs_uv,
                           double* Ug)
{
    if (d.O == 0 || max_frame_obs == 0) return;
    int64_t chunks = (max_frame_obs + SRK_FCHUNK - 1) / SRK_FCHUNK;
    hipLaunchKernelGGL(k_jac_frames, dim3((unsigned)chunks, (unsigned)d.M), dim3(256), 0, s, d, pts, cam, col_ptr,
                       fobs_pt, fobs_uv, Ug);
}

// expand the packed per-frame accumulators to the oracle's [M][10][10] + [M][10] layout (tests / downloads)
__global__ void k_expand_ug(int32_t M, const double* __restrict__ Ug, double* __restrict__ U, double* __restrict__ gf)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * 110) return;
    int j = t / 110, e = t % 110;
    const double* u = Ug + (int64_t)j * SRK_UG;
    if (e < 100) {
        int v1 = e / 10, v2 = e % 10;
        int a = v1 < v2 ? v1 : v2, b = v1 < v2 ? v2 : v1;
        int idx = a * 10 - a * (a - 1) / 2 + (b - a);
        U[(int64_t)j * 100 + e] = u[idx];
    } else {
        gf[(int64_t)j * 10 + (e - 100)] = u[55 + (e - 100)];
    }
}
void srk_launch_expand_ug(hipStream_t s, int32_t M, const double* Ug, double* U_full, double* g_full)
{
    int n = M * 110;
    hipLaunchKernelGGL(k_expand_ug, dim3((n + 255) / 256), dim3(256), 0, s, M, Ug, U_full, g_full);
}

// ------------------------------------------------------------------ 3x3 damped point block inverse
// E = V with diagonal * (1 + c) (:1825-1834); Eigen computeInverseAndDetWithCheck: invertible iff |det| > 1e-12
__device__ __forceinline__ bool point_block_inverse(const double* __restrict__ Vg, int64_t Ns, int64_t pt, double c,
                                                    double Einv[9], double g[3])
{
    double e00 = Vg[0 * Ns + pt] * (1 + c), e01 = Vg[1 * Ns + pt], e02 = Vg[2 * Ns + pt];
    double e11 = Vg[3 * Ns + pt] * (1 + c), e12 = Vg[4 * Ns + pt], e22 = Vg[5 * Ns + pt] * (1 + c);
    g[0] = Vg[6 * Ns + pt]; g[1] = Vg[7 * Ns + pt]; g[2] = Vg[8 * Ns + pt];
    double c00 = e11 * e22 - e12 * e12;
    double c01 = e12 * e02 - e01 * e22;
    double c02 = e01 * e12 - e11 * e02;
    double det = e00 * c00 + e01 * c01 + e02 * c02;
    if (!(fabs(det) > 1e-12)) return false;
    double id = 1 / det;
    Einv[0] = c00 * id;
    Einv[1] = (e02 * e12 - e01 * e22) * id;
    Einv[2] = (e01 * e12 - e02 * e11) * id;
    Einv[3] = c01 * id;
    Einv[4] = (e00 * e22 - e02 * e02) * id;
    Einv[5] = (e02 * e01 - e00 * e12) * id;
    Einv[6] = c02 * id;
    Einv[7] = (e01 * e02 - e00 * e12) * id;
    Einv[8] = (e00 * e11 - e01 * e01) * id;
    return true;
}

// The same block as a Cholesky factor E = L L^T, for the SYRK form of the Schur sum (k_schur_mm on uniform runs):
//     W^T E^-1 W = Z^T Z,  Z = L^-1 W,      W^T E^-1 g = Z^T (L^-1 g).
// Lc = the lower triangle of L^-1 by rows { I00, I10, I11, I20, I21, I22 } (so that every row of Z = L^-1 W is three
// multiply-adds on its own), h = L^-1 g.  Returns 0: |det| <= 1e-12, the landmark is skipped exactly as
// with the inverse (:1877-1881); 1: factor formed; 2: the block passes the determinant check but a pivot is not positive
// (E = V + c diag(V) with V a sum of outer products is positive definite in exact arithmetic, so this needs a damping
// factor below the rounding level of V) -- the caller hands such a landmark to the per-landmark inverse path, so the sum
// stays the reference's W^T E^-1 W whatever E is.
__device__ __forceinline__ int point_block_cholesky(const double* __restrict__ Vg, int64_t Ns, int64_t pt, double c,
                                                    double Lc[6], double h[3])
{
    const double e00 = Vg[0 * Ns + pt] * (1 + c), e01 = Vg[1 * Ns + pt], e02 = Vg[2 * Ns + pt];
    const double e11 = Vg[3 * Ns + pt] * (1 + c), e12 = Vg[4 * Ns + pt], e22 = Vg[5 * Ns + pt] * (1 + c);
    const double g0 = Vg[6 * Ns + pt], g1 = Vg[7 * Ns + pt], g2 = Vg[8 * Ns + pt];
    const double c00 = e11 * e22 - e12 * e12;
    const double c01 = e12 * e02 - e01 * e22;
    const double c02 = e01 * e12 - e11 * e02;
    const double det = e00 * c00 + e01 * c01 + e02 * c02; // the same expression as point_block_inverse: same skip decision
    if (!(fabs(det) > 1e-12)) return 0;
    const double l00 = sqrt(e00), i0 = 1 / l00;
    const double l10 = e01 * i0, l20 = e02 * i0;
    const double d1 = e11 - l10 * l10;
    const double l11 = sqrt(d1), i1 = 1 / l11;
    const double l21 = (e12 - l20 * l10) * i1;
    const double d2 = e22 - l20 * l20 - l21 * l21;
    const double l22 = sqrt(d2), i2 = 1 / l22;
    if (!(e00 > 0 && d1 > 0 && d2 > 0) || !isfinite(i0) || !isfinite(i1) || !isfinite(i2)) return 2;
    const double I10 = -l10 * i0 * i1, I21 = -l21 * i1 * i2;
    const double I20 = -(l20 * i0 + l21 * I10) * i2;
    Lc[0] = i0; Lc[1] = I10; Lc[2] = i1; Lc[3] = I20; Lc[4] = I21; Lc[5] = i2;
    h[0] = i0 * g0;
    h[1] = I10 * g0 + i1 * g1;
    h[2] = I20 * g0 + I21 * g1 + i2 * g2;
    return 1;
}

// ------------------------------------------------------------------ K3: Schur accumulation (per landmark)
// One workgroup per landmark: 3x3 elimination block inverted in registers, W_i staged in LDS, Y = E^-1 W_i in LDS,
// then the n_i(n_i+1)/2 lower 10x10 outer-product blocks are subtracted from S with fp64 atomics.
#define SRK_SCH 32 // observations per LDS chunk

// one landmark by the whole 256-thread workgroup (k_schur; and the tail workgroups of k_assemble for the landmarks the
// SYRK form of k_schur_mm hands back)
struct SchurLmScratch {
    double sWa[SRK_SCH][30];
    double sYb[SRK_SCH][30];
    int32_t sFa[SRK_SCH], sFb[SRK_SCH];
};
template <typename WT>
__device__ __forceinline__ void schur_one_landmark(const SrkDims& d, double c, const int64_t* __restrict__ row_ptr,
                                                   const int32_t* __restrict__ obs_frame, const WT* __restrict__ W,
                                                   const double* __restrict__ Vg, double* __restrict__ S,
                                                   double* __restrict__ rhs, int64_t pt, SchurLmScratch& sm)
{
    auto& sWa = sm.sWa;
    auto& sYb = sm.sYb;
    auto& sFa = sm.sFa;
    auto& sFb = sm.sFb;
    double Einv[9], g[3];
    bool ok = point_block_inverse(Vg, d.Ns, pt, c, Einv, g); // block-uniform
    if (!ok) return;                                          // :1877-1881 skip the landmark
    double Eg[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) Eg[m] = Einv[3 * m] * g[0] + Einv[3 * m + 1] * g[1] + Einv[3 * m + 2] * g[2];
    int64_t o0 = row_ptr[pt];
    int n = (int)(row_ptr[pt + 1] - o0);
    int nchunks = (n + SRK_SCH - 1) / SRK_SCH;
    for (int ca = 0; ca < nchunks; ++ca) {
        int a0 = ca * SRK_SCH;
        int na = n - a0 < SRK_SCH ? n - a0 : SRK_SCH;
        __syncthreads();
        for (int t = threadIdx.x; t < na * 30; t += 256) {
            int k = t / na, a = t - k * na;
            sWa[a][k] = w_entry<WT>(W, d.Os, o0 + a0 + a, k);
        }
        if (threadIdx.x < na) sFa[threadIdx.x] = obs_frame[o0 + a0 + threadIdx.x];
        __syncthreads();
        // rhs += F^T E^-1 g   (:1895-1897)
        for (int t = threadIdx.x; t < na * 10; t += 256) {
            int a = t / 10, r = t - a * 10;
            int64_t row = 10 * (int64_t)sFa[a] + r;
            if (!srk_is_fixed_var(row, d)) {
                double v = sWa[a][r] * Eg[0] + sWa[a][10 + r] * Eg[1] + sWa[a][20 + r] * Eg[2];
                atomicAdd(&rhs[row], v);
            }
        }
        for (int cb = 0; cb <= ca; ++cb) {
            int b0 = cb * SRK_SCH;
            int nb = n - b0 < SRK_SCH ? n - b0 : SRK_SCH;
            __syncthreads();
            for (int t = threadIdx.x; t < nb * 10; t += 256) {
                int fv = t / nb, b = t - fv * nb;
                int64_t ob = o0 + b0 + b;
                double w0 = w_entry<WT>(W, d.Os, ob, fv), w1 = w_entry<WT>(W, d.Os, ob, 10 + fv), w2 = w_entry<WT>(W, d.Os, ob, 20 + fv);
                // Y = E^-1 W  (3 x 10)
                sYb[b][fv] = Einv[0] * w0 + Einv[1] * w1 + Einv[2] * w2;
                sYb[b][10 + fv] = Einv[3] * w0 + Einv[4] * w1 + Einv[5] * w2;
                sYb[b][20 + fv] = Einv[6] * w0 + Einv[7] * w1 + Einv[8] * w2;
            }
            if (threadIdx.x < nb) sFb[threadIdx.x] = obs_frame[o0 + b0 + threadIdx.x];
            __syncthreads();
            int total = na * nb * 100;
            for (int t = threadIdx.x; t < total; t += 256) {
                int e = t % 100, pr = t / 100;
                int b = pr % nb, a = pr / nb;
                if (ca == cb && b > a) continue; // lower block triangle only (frames ascend inside a landmark)
                int r = e / 10, cc = e - r * 10;
                int64_t row = 10 * (int64_t)sFa[a] + r, col = 10 * (int64_t)sFb[b] + cc;
                if (srk_is_fixed_var(row, d) || srk_is_fixed_var(col, d)) continue;
                double v = sWa[a][r] * sYb[b][cc] + sWa[a][10 + r] * sYb[b][10 + cc] + sWa[a][20 + r] * sYb[b][20 + cc];
                atomicAdd(&S[row * d.ld + col], -v); // S = G - sum F^T E^-1 F  (:1891-1892)
            }
        }
    }
    __syncthreads(); // the scratch is free for the next landmark
}

template <typename WT>
__global__ __launch_bounds__(256) void k_schur(SrkDims d, double c, const int64_t* __restrict__ row_ptr,
                                               const int32_t* __restrict__ obs_frame, const WT* __restrict__ W,
                                               const double* __restrict__ Vg, double* __restrict__ S,
                                               double* __restrict__ rhs, const int32_t* __restrict__ pt_list,
                                               int64_t n_list)
{
    __shared__ SchurLmScratch sm;
    for (int64_t li = blockIdx.x; li < n_list; li += gridDim.x)
        schur_one_landmark<WT>(d, c, row_ptr, obs_frame, W, Vg, S, rhs, pt_list ? pt_list[li] : li, sm);
}

void srk_launch_schur(hipStream_t s, const SrkDims& d, double c, const int64_t* row_ptr, const int32_t* obs_frame,
                      const double* W, const double* Vg, double* S, double* rhs, const int32_t* pt_list,
                      int64_t n_list)
{
    if (n_list <= 0) return;
    int64_t blocks = n_list < 65536 ? n_list : 65536;
    if (d.w_f32)
        hipLaunchKernelGGL(k_schur<float>, dim3((unsigned)blocks), dim3(256), 0, s, d, c, row_ptr, obs_frame,
                           reinterpret_cast<const float*>(W), Vg, S, rhs, pt_list, n_list);
    else
        hipLaunchKernelGGL(k_schur<double>, dim3((unsigned)blocks), dim3(256), 0, s, d, c, row_ptr, obs_frame, W, Vg, S, rhs,
                           pt_list, n_list);
}

// ------------------------------------------------------------------ K3g: Schur accumulation, grouped landmarks
// Landmarks are stored sorted by their frame list, so landmarks that see exactly the same frames are contiguous.
// One workgroup takes a run of such landmarks (<= SRK_GRP_MAXPTS) and keeps the whole lower block triangle of their
// common nf x nf frame-pair blocks in REGISTERS (5x10 half blocks, SLOTS per thread); every landmark's 3x3
// elimination block is inverted once up front, W_i and Y_i = E_i^-1 W_i are staged in LDS four landmarks at a time
// (the next four are already in flight in registers while these are multiplied), and the sums leave the chip once per
// run as fp64 atomics -- ~100x fewer atomic bytes than one flush per landmark.
// The accumulation is LDS-bandwidth bound (every operand of the rank-3 updates is an LDS read): a 5x10 register tile
// needs 15 operands per 50 FMAs, against 12 per 20 for the 2x10 strips this kernel started with.
#define SRK_GRP_THREADS 512
#define SRK_GRP_MAXNF 24     // nf (nf + 1) half blocks <= SRK_GRP_THREADS * 2
#define SRK_GRP_NF1 21       // nf (nf + 1) <= SRK_GRP_THREADS: one half block per thread
#define SRK_GRP_MAXPTS 128
#ifndef SRK_GRP_PB
#define SRK_GRP_PB 4         // landmarks staged per barrier round
#endif
#define SRK_GRP_WS 36        // LDS doubles per (landmark, frame) of W: 3 x (5 | pad | 5 | pad), halves 16-byte aligned
#define SRK_GRP_PRE ((30 * SRK_GRP_PB * SRK_GRP_MAXNF + SRK_GRP_THREADS - 1) / SRK_GRP_THREADS) // prefetch registers
static_assert(SRK_GRP_MAXNF == SRK_GRP_MAXNF_HOST && SRK_GRP_NF1 == SRK_GRP_NF1_HOST && SRK_GRP_MAXPTS == SRK_GRP_MAXPTS_HOST,
              "host / kernel grouping constants differ");
static_assert(SRK_GRP_PRE * SRK_GRP_THREADS >= 30 * SRK_GRP_PB * SRK_GRP_MAXNF, "staging map too small");
static_assert(SRK_GRP_NF1 * (SRK_GRP_NF1 + 1) <= SRK_GRP_THREADS && SRK_GRP_MAXNF * (SRK_GRP_MAXNF + 1) <= 2 * SRK_GRP_THREADS,
              "half blocks do not fit the thread slots");

// Arithmetic type of the accumulation.  double: the reference's arithmetic.  float (opt-in, srk_ba_set_schur_precision):
// W and Y are rounded to fp32 when they are staged, the rank-3 updates run as packed fp32 FMAs (twice the fp64 rate,
// half the LDS bytes and accumulator registers) over the <= 128 landmarks of a run, and the run's sums are added to
// the fp64 system -- a mixed-precision reduced camera system.  LDS strides keep every 5-wide half
// row and every 10-wide row on a 16-byte boundary.
template <typename T> struct SchurLayout;
template <> struct SchurLayout<double> { static constexpr int WH = 6, WM = 12, WS = 36, YM = 10, YS = 30; };
template <> struct SchurLayout<float> { static constexpr int WH = 8, WM = 16, WS = 48, YM = 12, YS = 36; };
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// acc[5][10] += w[0..4] (x) y[0..9] for one point coordinate m
__device__ __forceinline__ void schur_tile_update(double (&acc)[5][10], const double* wp, const double* yv)
{
    const double2 w01 = *reinterpret_cast<const double2*>(wp);
    const double2 w23 = *reinterpret_cast<const double2*>(wp + 2);
    const double wr[5] = { w01.x, w01.y, w23.x, w23.y, wp[4] };
    const double2* yp = reinterpret_cast<const double2*>(yv);
#pragma unroll
    for (int h = 0; h < 5; ++h) {
        const double2 y2 = yp[h];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            acc[i][2 * h] = fma(wr[i], y2.x, acc[i][2 * h]);
            acc[i][2 * h + 1] = fma(wr[i], y2.y, acc[i][2 * h + 1]);
        }
    }
}
// fp32: the ten columns as five packed pairs (v_pk_fma_f32)
__device__ __forceinline__ void schur_tile_update(float (&acc)[5][10], const float* wp, const float* yv)
{
    const float4_t w0123 = *reinterpret_cast<const float4_t*>(wp);
    const float wr[5] = { w0123.x, w0123.y, w0123.z, w0123.w, wp[4] };
    const float4_t y0123 = *reinterpret_cast<const float4_t*>(yv), y4567 = *reinterpret_cast<const float4_t*>(yv + 4);
    const float2_t y89 = *reinterpret_cast<const float2_t*>(yv + 8);
    const float2_t y2[5] = { { y0123.x, y0123.y }, { y0123.z, y0123.w }, { y4567.x, y4567.y }, { y4567.z, y4567.w }, y89 };
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float2_t w2 = { wr[i], wr[i] };
#pragma unroll
        for (int h = 0; h < 5; ++h) {
            float2_t a2 = { acc[i][2 * h], acc[i][2 * h + 1] };
            a2 = __builtin_elementwise_fma(w2, y2[h], a2);
            acc[i][2 * h] = a2.x;
            acc[i][2 * h + 1] = a2.y;
        }
    }
}

template <int SLOTS, typename T, typename WT>
__global__ __launch_bounds__(SRK_GRP_THREADS) void k_schur_grouped(
    SrkDims d, double c, const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ obs_pt,
    const uint8_t* __restrict__ obs_slot, const uint32_t* __restrict__ pt_mask, const WT* __restrict__ W,
    const double* __restrict__ Vg, double* __restrict__ S, double* __restrict__ rhs,
    const int32_t* __restrict__ grp_first, const int32_t* __restrict__ grp_count, const int32_t* __restrict__ grp_nf,
    const int32_t* __restrict__ grp_frames, int nf_skip /* runs with at most this many frames belong to k_schur_mm / k_schur_ws */)
{
    // one LDS arena: W | Y staging during the accumulation, then the staging buffer of the coalesced flush
    using L = SchurLayout<T>;
    constexpr int W_LM = SRK_GRP_MAXNF * L::WS, Y_LM = SRK_GRP_MAXNF * L::YS; // elements per staged landmark
    constexpr int CAP = SRK_GRP_PB * (W_LM + Y_LM);
    __shared__ __attribute__((aligned(16))) T sBuf[CAP];
    __shared__ __attribute__((aligned(16))) double sE[SRK_GRP_MAXPTS][12];
    __shared__ int32_t sF[SRK_GRP_MAXNF];
    __shared__ uint32_t sM[SRK_GRP_PB]; // ragged runs: the frame slots each staged landmark sees
    T* const sW = sBuf;
    T* const sY = sBuf + SRK_GRP_PB * W_LM;
    const int tid = threadIdx.x;
    const int64_t p0 = grp_first[blockIdx.x];
    const int np = grp_count[blockIdx.x];
    // the run's frame set (union of its landmarks' frame lists); ragged = some landmark misses some of these frames
    const int nfu = grp_nf[blockIdx.x];
    const bool ragged = nfu < 0;
    const int nf = ragged ? -nfu : nfu;
    if ((SLOTS == 1) != (nf <= SRK_GRP_NF1) || nf <= nf_skip) return; // another kernel takes this run
    if (tid < nf) sF[tid] = grp_frames[(int64_t)blockIdx.x * SRK_GRP_MAXNF + tid];
    if (tid < np) { // 3x3 damped block inverses; a singular block contributes nothing (:1877-1881)
        double Einv[9], g[3];
        bool ok = point_block_inverse(Vg, d.Ns, p0 + tid, c, Einv, g);
#pragma unroll
        for (int k = 0; k < 9; ++k) sE[tid][k] = ok ? Einv[k] : 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m)
            sE[tid][9 + m] = ok ? Einv[3 * m] * g[0] + Einv[3 * m + 1] * g[1] + Einv[3 * m + 2] * g[2] : 0.0;
    }
    // half block u -> (block pair (a, b), b <= a ; rows 5 hf .. 5 hf + 4)
    const int n_half = nf * (nf + 1);
    int offW[SLOTS], offY[SLOTS], sa[SLOTS], sb[SLOTS], sh[SLOTS];
    bool act[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        int u = tid + s * SRK_GRP_THREADS;
        act[s] = u < n_half;
        int pi = act[s] ? u >> 1 : 0;
        int hf = act[s] ? u & 1 : 0;
        int a = (int)((sqrtf(8.0f * (float)pi + 1.0f) - 1.0f) * 0.5f);
        while ((a + 1) * (a + 2) / 2 <= pi) ++a;
        while (a * (a + 1) / 2 > pi) --a;
        int b = pi - a * (a + 1) / 2;
        sa[s] = a; sb[s] = b; sh[s] = hf;
        offW[s] = a * L::WS + L::WH * hf;
        offY[s] = b * L::YS;
    }
    T acc[SLOTS][5][10];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int cc = 0; cc < 10; ++cc) acc[s][i][cc] = 0;
    // staging map (the same for every round): element idx = tid + j THREADS -> W row k = idx / (PB nf), position q in
    // the round's PB nf consecutive observations (coalesced over q); LDS slot of (landmark q / nf, frame q % nf, k).
    // Everything per-thread is worked out here, once: the round loop below only adds the round's offset.
    const int qn = SRK_GRP_PB * nf;
    int gk[SRK_GRP_PRE]; // entry (k = 10 pv + fv) of the staged element; its observation is oa + qpos
    int loff[SRK_GRP_PRE], koff[SRK_GRP_PRE], qpos[SRK_GRP_PRE];
#pragma unroll
    for (int j = 0; j < SRK_GRP_PRE; ++j) {
        int idx = tid + j * SRK_GRP_THREADS;
        int k = idx / qn, q = idx - k * qn;
        bool in = k < 30;
        int pl = q / nf, a = q - pl * nf;
        int m = k / 10, r = k - 10 * m;
        qpos[j] = in ? q : (1 << 30);
        koff[j] = L::WM * m + r + (r >= 5 ? L::WH - 5 : 0);
        loff[j] = in ? pl * W_LM + a * L::WS + koff[j] : 0; // uniform runs: landmark q / nf, slot q % nf
        gk[j] = in ? k : 0;
    }
    // Y stage map: item t = tid + i THREADS -> (staging slot pl, frame a, frame variable fv).  The same thread also
    // accumulates rhs: W^T (E^-1 g) of ITS staging slot; the PB slots of one (a, fv) are summed by the flush atomics.
    constexpr int YI = (SRK_GRP_PB * SRK_GRP_MAXNF * 10 + SRK_GRP_THREADS - 1) / SRK_GRP_THREADS;
    int ypl[YI], ywo[YI], yyo[YI], yrow[YI], ya[YI];
    double racc[YI];
#pragma unroll
    for (int i = 0; i < YI; ++i) {
        int t = tid + i * SRK_GRP_THREADS;
        int pl = t / (nf * 10), e = t - pl * nf * 10;
        int a = e / 10, fv = e - a * 10;
        ypl[i] = pl < SRK_GRP_PB ? pl : (1 << 30);
        ywo[i] = pl * W_LM + a * L::WS + fv + (fv >= 5 ? L::WH - 5 : 0);
        yyo[i] = pl * Y_LM + a * L::YS + fv;
        yrow[i] = e;
        ya[i] = a;
        racc[i] = 0;
    }
    // The observations of the round's landmarks are contiguous: [row_ptr[p0 + pb], row_ptr[p0 + pb + nb]).  Ragged runs
    // also fetch each observation's (staged landmark, frame slot) and the landmarks' slot masks.
    double pre[SRK_GRP_PRE];
    int pdst[SRK_GRP_PRE];
    uint32_t pmask = 0;
    auto prefetch = [&](int pb) {
        const int nbn = np - pb < SRK_GRP_PB ? np - pb : SRK_GRP_PB;
        const int64_t oa = row_ptr[p0 + pb];
        const int nq = (int)(row_ptr[p0 + pb + nbn] - oa);
#pragma unroll
        for (int j = 0; j < SRK_GRP_PRE; ++j) pre[j] = qpos[j] < nq ? w_entry<WT>(W, d.Os, oa + qpos[j], gk[j]) : 0.0;
        if (ragged) {
#pragma unroll
            for (int j = 0; j < SRK_GRP_PRE; ++j)
                if (qpos[j] < nq)
                    pdst[j] = (obs_pt[oa + qpos[j]] - (int)(p0 + pb)) * W_LM + (int)obs_slot[oa + qpos[j]] * L::WS + koff[j];
            if (tid < nbn) pmask = pt_mask[p0 + pb + tid];
        }
        return nq;
    };
    int nq_next = prefetch(0);
    for (int pb = 0; pb < np; pb += SRK_GRP_PB) {
        const int nb = np - pb < SRK_GRP_PB ? np - pb : SRK_GRP_PB;
        const int nq = nq_next;
        __syncthreads(); // the previous round's W / Y are consumed (first round: sE, sF are visible)
        if (!ragged) {
#pragma unroll
            for (int j = 0; j < SRK_GRP_PRE; ++j)
                if (qpos[j] < nq) sW[loff[j]] = (T)pre[j];
        } else {
#pragma unroll
            for (int j = 0; j < SRK_GRP_PRE; ++j)
                if (qpos[j] < nq) sW[pdst[j]] = (T)pre[j];
            if (tid < nb) sM[tid] = pmask;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            if (ypl[i] >= nb) continue;
            T* wp = sW + ywo[i];
            T* yp = sY + yyo[i];
            if (ragged && !((sM[ypl[i]] >> ya[i]) & 1u)) { // this landmark does not see this frame: zero blocks
                wp[0] = wp[L::WM] = wp[2 * L::WM] = (T)0;
                yp[0] = yp[L::YM] = yp[2 * L::YM] = (T)0;
                continue;
            }
            const double2* E2 = reinterpret_cast<const double2*>(sE[pb + ypl[i]]); // rows are 96 B: 16-byte aligned
            const double2 e01 = E2[0], e23 = E2[1], e45 = E2[2], e67 = E2[3], e89 = E2[4], eab = E2[5];
            const double w0 = (double)wp[0], w1 = (double)wp[L::WM], w2 = (double)wp[2 * L::WM];
            yp[0] = (T)(e01.x * w0 + e01.y * w1 + e23.x * w2);
            yp[L::YM] = (T)(e23.y * w0 + e45.x * w1 + e45.y * w2);
            yp[2 * L::YM] = (T)(e67.x * w0 + e67.y * w1 + e89.x * w2);
            racc[i] += w0 * e89.y + w1 * eab.x + w2 * eab.y;
        }
        if (pb + SRK_GRP_PB < np) nq_next = prefetch(pb + SRK_GRP_PB); // in flight while this round is multiplied
        __syncthreads();
        for (int pl = 0; pl < nb; ++pl) {
            const T* w = sW + pl * W_LM;
            const T* yv = sY + pl * Y_LM;
#ifdef SRK_SCH_NOACC
            if (d.N >= 0) continue;
#endif
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                if (!act[s]) continue;
#pragma unroll
                for (int m = 0; m < 3; ++m) schur_tile_update(acc[s], w + offW[s] + L::WM * m, yv + offY[s] + L::YM * m);
            }
        }
    }
    // flush: S -= sum F^T E^-1 F (lower block triangle), rhs += sum F^T E^-1 g, gauge rows/columns dropped.
    // The register tiles are transposed through LDS a few block rows at a time so that consecutive lanes add to
    // consecutive columns of one row of S (block row a = 10 rows of 10 (a + 1) doubles, contiguous when the frames
    // are) -- a wave-wide atomic then covers whole cache lines instead of 64 different rows.
    for (int a0 = 0; a0 < nf;) {
        int a1 = a0, used = 0;
        while (a1 < nf && used + 100 * (a1 + 1) <= CAP) { used += 100 * (a1 + 1); ++a1; }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            if (!act[s] || sa[s] < a0 || sa[s] >= a1) continue;
            int off = 50 * (sa[s] * (sa[s] + 1) - a0 * (a0 + 1)); // sum_{a'=a0}^{a-1} 100 (a' + 1)
            int w = 10 * (sa[s] + 1);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int cc = 0; cc < 10; ++cc) sBuf[off + (5 * sh[s] + i) * w + 10 * sb[s] + cc] = acc[s][i][cc];
        }
        __syncthreads();
        // one wave per row of S: rows rho = 10 (a - a0) + r of the staged block rows go round the waves, the lanes walk
        // the row's 10 (a + 1) columns (consecutive addresses where the frames are consecutive); the only divisions
        // left are by the constant 10
        {
            const int lane = tid & 63, wv = tid >> 6;
            for (int rho = wv; rho < 10 * (a1 - a0); rho += SRK_GRP_THREADS / 64) {
                const int a = a0 + rho / 10, r = rho - 10 * (a - a0);
                const int w = 10 * (a + 1);
                const int64_t row = 10 * (int64_t)sF[a] + r;
                if (srk_is_fixed_var(row, d)) continue;
                const T* src = sBuf + 50 * (a * (a + 1) - a0 * (a0 + 1)) + r * w;
                double* dst = S + row * d.ld;
                for (int cw = lane; cw < w; cw += 64) {
                    const int b = cw / 10, cc = cw - b * 10;
                    const int64_t col = 10 * (int64_t)sF[b] + cc;
                    if (srk_is_fixed_var(col, d)) continue;
#ifdef SRK_SCH_NOFLUSH
                    if (d.N >= 0) continue;
#endif
                    atomicAdd(&dst[col], -(double)src[cw]);
                }
            }
        }
        a0 = a1;
    }
#pragma unroll
    for (int i = 0; i < YI; ++i) {
        if (ypl[i] >= SRK_GRP_PB || ypl[i] >= np) continue; // a slot that never held a landmark
        int a = yrow[i] / 10, r = yrow[i] - a * 10;
        int64_t row = 10 * (int64_t)sF[a] + r;
        if (!srk_is_fixed_var(row, d)) atomicAdd(&rhs[row], racc[i]);
    }
}

// ------------------------------------------------------------------ K3w: the same, with a dedicated loader wave
// k_schur_grouped spends about as long staging (global -> LDS, Y = E^-1 W, three barriers a round) as multiplying,
// and with 214 accumulator-heavy VGPRs only one workgroup fits a CU, so nothing overlaps the two.  A run over at most
// SRK_WS_NF frames has nf (nf + 1) <= 420 half blocks = the lanes of waves 0..6: wave 7 does nothing but stage.  Here
// it stages round r + 1 into the second half of a double-buffered LDS arena (and already has round r + 2's global
// loads in flight) while waves 0..6 multiply round r -- one barrier per round, the multiply never waits for memory.
#define SRK_WS_NF 20
template <typename T, typename WT>
__global__ __launch_bounds__(SRK_GRP_THREADS) void k_schur_ws(
    SrkDims d, double c, const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ obs_pt,
    const uint8_t* __restrict__ obs_slot, const uint32_t* __restrict__ pt_mask, const WT* __restrict__ W,
    const double* __restrict__ Vg, double* __restrict__ S, double* __restrict__ rhs,
    const int32_t* __restrict__ grp_first, const int32_t* __restrict__ grp_count, const int32_t* __restrict__ grp_nf,
    const int32_t* __restrict__ grp_frames)
{
    using L = SchurLayout<T>;
    constexpr int PB = SRK_GRP_PB;
    constexpr int W_LM = SRK_WS_NF * L::WS, Y_LM = SRK_WS_NF * L::YS; // elements per staged landmark
    constexpr int BUF = PB * (W_LM + Y_LM);                           // one staging buffer
    constexpr int CAP = 2 * BUF;                                      // the flush uses both
    __shared__ __attribute__((aligned(16))) T sBuf[CAP];
    __shared__ __attribute__((aligned(16))) double sE[SRK_GRP_MAXPTS][12];
    __shared__ double sRhs[SRK_WS_NF * 10];
    __shared__ int32_t sF[SRK_WS_NF];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t p0 = grp_first[blockIdx.x];
    const int np = grp_count[blockIdx.x];
    const int nfu = grp_nf[blockIdx.x];
    const bool ragged = nfu < 0;
    const int nf = ragged ? -nfu : nfu;
    if (nf > SRK_WS_NF) return; // k_schur_grouped takes the wider runs
    if (tid < nf) sF[tid] = grp_frames[(int64_t)blockIdx.x * SRK_GRP_MAXNF + tid];
    if (tid < nf * 10) sRhs[tid] = 0.0;
    if (tid < np) { // 3x3 damped block inverses; a singular block contributes nothing (:1877-1881)
        double Einv[9], g[3];
        bool ok = point_block_inverse(Vg, d.Ns, p0 + tid, c, Einv, g);
#pragma unroll
        for (int k = 0; k < 9; ++k) sE[tid][k] = ok ? Einv[k] : 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m)
            sE[tid][9 + m] = ok ? Einv[3 * m] * g[0] + Einv[3 * m + 1] * g[1] + Einv[3 * m + 2] * g[2] : 0.0;
    }
    const int R = (np + PB - 1) / PB;
    const int nf10 = nf * 10;
    // one pass of the flush streams the staged block rows a0 .. a1 - 1 out of LDS: one wave per row of S, the lanes walk
    // the row's 10 (a + 1) columns
    auto flush_stream = [&](int a0, int a1) {
        for (int rho = wv; rho < 10 * (a1 - a0); rho += SRK_GRP_THREADS / 64) {
            const int a = a0 + rho / 10, r = rho - 10 * (a - a0);
            const int w = 10 * (a + 1);
            const int64_t row = 10 * (int64_t)sF[a] + r;
            if (srk_is_fixed_var(row, d)) continue;
            const T* src = sBuf + 50 * (a * (a + 1) - a0 * (a0 + 1)) + r * w;
            double* dst = S + row * d.ld;
            for (int cw = lane; cw < w; cw += 64) {
                const int b = cw / 10, cc = cw - b * 10;
                const int64_t col = 10 * (int64_t)sF[b] + cc;
                if (srk_is_fixed_var(col, d)) continue;
#ifdef SRK_SCH_NOFLUSH
                if (d.N >= 0) continue;
#endif
                atomicAdd(&dst[col], -(double)src[cw]);
            }
        }
    };
    auto flush_span = [&](int a0, int& a1) { // block rows that fit the arena from a0 on
        int used = 0;
        a1 = a0;
        while (a1 < nf && used + 100 * (a1 + 1) <= CAP) { used += 100 * (a1 + 1); ++a1; }
    };
    __syncthreads(); // sE, sF, sRhs are visible
    // The two roles are two separate code paths with the same barrier sequence (R + 1 for the rounds, two per flush
    // pass), so that the accumulators live only in the multiplier path and the prefetch registers only in the loader's.
    if (wv == 7) {
        // ---- loader.  A round's observations are contiguous, at most PB nf <= 80: the lane covers q = lane and
        // q = lane + 64; for each it moves the 30 W values (k-major in memory: one coalesced load per k).
        const unsigned magic_nf = (65536u + nf - 1) / nf; // q < 128
        double pre[30][2];
        int dq[2]; // LDS offset of (staged landmark, frame slot) for the two q of this lane
        int nq_pre = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) { // uniform runs: landmark q / nf, slot q % nf
            const int q = lane + 64 * h;
            const int pl = (int)((q * magic_nf) >> 16);
            dq[h] = pl * W_LM + (q - pl * nf) * L::WS;
        }
        auto load_round = [&](int r) { // global loads of round r into `pre` (left in flight)
            const int pb = r * PB;
            const int nbn = np - pb < PB ? np - pb : PB;
            const int64_t oa = row_ptr[p0 + pb];
            nq_pre = (int)(row_ptr[p0 + pb + nbn] - oa);
#pragma unroll
            for (int k = 0; k < 30; ++k) {
                pre[k][0] = lane < nq_pre ? w_entry<WT>(W, d.Os, oa + lane, k) : 0.0;
                pre[k][1] = lane + 64 < nq_pre ? w_entry<WT>(W, d.Os, oa + lane + 64, k) : 0.0;
            }
            if (ragged) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = lane + 64 * h;
                    if (q < nq_pre) dq[h] = (obs_pt[oa + q] - (int)(p0 + pb)) * W_LM + (int)obs_slot[oa + q] * L::WS;
                }
            }
        };
        auto stage_round = [&](T* bw) { // `pre` -> W in LDS (Y = E^-1 W is the multipliers' first step of the round)
#pragma unroll
            for (int k = 0; k < 30; ++k) {
                const int m = k / 10, rr = k - 10 * m;
                const int ko = L::WM * m + rr + (rr >= 5 ? L::WH - 5 : 0);
                if (lane < nq_pre) bw[dq[0] + ko] = (T)pre[k][0];
                if (lane + 64 < nq_pre) bw[dq[1] + ko] = (T)pre[k][1];
            }
        };
        load_round(0);
        stage_round(sBuf);
        if (R > 1) load_round(1);
        __syncthreads();
        for (int r = 0; r < R; ++r) {
            __syncthreads(); // the multipliers' Y of round r: pass at once, they must not wait for the staging below
            if (r + 1 < R) {
#ifndef SRK_WS_NOSTAGE
                stage_round(sBuf + ((r + 1) & 1) * BUF);
#endif
#ifndef SRK_WS_NOLOAD
                if (r + 2 < R) load_round(r + 2);
#endif
            }
            __syncthreads(); // their products of round r
        }
        for (int a0 = 0; a0 < nf;) {
            int a1;
            flush_span(a0, a1);
            __syncthreads();
            __syncthreads();
            flush_stream(a0, a1);
            a0 = a1;
        }
    } else {
        // ---- multipliers: half block u = tid -> (block pair (a, b), b <= a ; rows 5 hf .. 5 hf + 4)
        const bool act = tid < nf * (nf + 1);
        int sa, sb, sh;
        {
            int pi = act ? tid >> 1 : 0;
            sh = act ? tid & 1 : 0;
            sa = (int)((sqrtf(8.0f * (float)pi + 1.0f) - 1.0f) * 0.5f);
            while ((sa + 1) * (sa + 2) / 2 <= pi) ++sa;
            while (sa * (sa + 1) / 2 > pi) --sa;
            sb = pi - sa * (sa + 1) / 2;
        }
        const int offW = sa * L::WS + L::WH * sh, offY = sb * L::YS;
        T acc[5][10];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int cc = 0; cc < 10; ++cc) acc[i][cc] = 0;
        // Y stage map: item t = tid + 448 i -> (staging slot pl, frame a, frame variable fv); the same thread keeps the
        // rhs term W^T (E^-1 g) of its items (the PB slots of one (a, fv) meet in sRhs at the end)
        constexpr int YI = (PB * SRK_WS_NF * 10 + 447) / 448;
        int ypl[YI], ywo[YI], yyo[YI], ye[YI];
        double racc[YI];
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            const int t = tid + 448 * i;
            const int pl = t / nf10, e = t - pl * nf10;
            const int a = e / 10, fv = e - a * 10;
            ypl[i] = pl < PB ? pl : (1 << 30);
            ywo[i] = pl * W_LM + a * L::WS + fv + (fv >= 5 ? L::WH - 5 : 0);
            yyo[i] = pl * Y_LM + a * L::YS + fv;
            ye[i] = e;
            racc[i] = 0;
        }
        __syncthreads();
        for (int r = 0; r < R; ++r) {
            T* bw = sBuf + (r & 1) * BUF;
            T* by = bw + PB * W_LM;
            const int pb = r * PB;
            const int nb = np - pb < PB ? np - pb : PB;
#pragma unroll
            for (int i = 0; i < YI; ++i) {
                if (ypl[i] >= nb) continue;
                T* wp = bw + ywo[i];
                T* yp = by + yyo[i];
                if (ragged && !((pt_mask[p0 + pb + ypl[i]] >> (ye[i] / 10)) & 1u)) { // landmark misses this frame: zeros
                    wp[0] = wp[L::WM] = wp[2 * L::WM] = (T)0;
                    yp[0] = yp[L::YM] = yp[2 * L::YM] = (T)0;
                    continue;
                }
                const double2* E2 = reinterpret_cast<const double2*>(sE[pb + ypl[i]]); // rows are 96 B: 16-byte aligned
                const double2 e01 = E2[0], e23 = E2[1], e45 = E2[2], e67 = E2[3], e89 = E2[4], eab = E2[5];
                const double w0 = (double)wp[0], w1 = (double)wp[L::WM], w2 = (double)wp[2 * L::WM];
                yp[0] = (T)(e01.x * w0 + e01.y * w1 + e23.x * w2);
                yp[L::YM] = (T)(e23.y * w0 + e45.x * w1 + e45.y * w2);
                yp[2 * L::YM] = (T)(e67.x * w0 + e67.y * w1 + e89.x * w2);
                racc[i] += w0 * e89.y + w1 * eab.x + w2 * eab.y;
            }
            __syncthreads();
            if (act) {
                for (int pl = 0; pl < nb; ++pl) {
#ifdef SRK_SCH_NOACC
                    if (d.N >= 0) continue;
#endif
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        schur_tile_update(acc, bw + pl * W_LM + offW + L::WM * m, by + pl * Y_LM + offY + L::YM * m);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < YI; ++i)
            if (ypl[i] < PB && ypl[i] < np) atomicAdd(&sRhs[ye[i]], racc[i]);
        // flush: the tiles are transposed through LDS a few block rows at a time
        for (int a0 = 0; a0 < nf;) {
            int a1;
            flush_span(a0, a1);
            __syncthreads();
            if (act && sa >= a0 && sa < a1) {
                int off = 50 * (sa * (sa + 1) - a0 * (a0 + 1)); // sum_{a'=a0}^{a-1} 100 (a' + 1)
                int w = 10 * (sa + 1);
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int cc = 0; cc < 10; ++cc) sBuf[off + (5 * sh + i) * w + 10 * sb + cc] = acc[i][cc];
            }
            __syncthreads();
            flush_stream(a0, a1);
            a0 = a1;
        }
    }
    // rhs += sum F^T E^-1 g (the multipliers' sRhs adds precede the flush's barriers; nf >= 1 means at least one pass)
    if (tid < nf10) {
        const int a = tid / 10, r = tid - a * 10;
        const int64_t row = 10 * (int64_t)sF[a] + r;
        if (!srk_is_fixed_var(row, d)) atomicAdd(&rhs[row], sRhs[tid]);
    }
}

// ------------------------------------------------------------------ K3m: the run's sum as an fp64 MFMA product
// SQ counters of the register-tile kernels above (tools/schur_pmc.sh): LDS array 65 % busy, a quarter of that bank
// conflicts, vector ALU 50 % -- every FMA operand is an LDS read (0.3-0.4 doubles per FMA).  But the sum over a run's
// landmarks IS a matrix product: with E_i = L_i L_i^T and the run's Z_i = L_i^-1 W_i stacked as rows k = (landmark, point
// coordinate),
//     sum_i W_i^T E_i^-1 W_i = Z^T Z,   Z: (3 np) x (10 nf),
// a (10 nf) x (10 nf) x (3 np) fp64 SYRK.  v_mfma_f64_16x16x4 has the vector pipe's peak (78.6 TFLOP/s) but reads
// 2 operand doubles per 16 FMAs of a lane.  Rows of the round buffers are k, columns the frame variables, row stride 208
// doubles (= 16 mod 32: the four k rows an MFMA operand spans hit different banks).  The element of the lower BLOCK triangle
// that lies above the TILE diagonal (a 10x10 diagonal block cut by a tile boundary) is taken from its mirror image: the sum
// is symmetric.  Roles, rounds and what was measured: at the kernel (k_schur_mm) below.
#ifdef SRK_MM_STAMPS // development (tools/mm_stamps.sh): in-kernel clock stamps of one multiplier and one loader lane
__device__ long long g_mm_stamps[2048][16];
#define MM_STAMP(k) do { if (tid == 0 && blockIdx.x < 2048) g_mm_stamps[blockIdx.x][k] = wall_clock64(); } while (0)
#define MM_ACC(who, k, t0) do { if (tid == (who) && blockIdx.x < 2048) { long long t_ = wall_clock64(); g_mm_stamps[blockIdx.x][k] += t_ - (t0); (t0) = t_; } } while (0)
extern "C" void srk_dbg_mm_stamps(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mm_stamps), sizeof(long long) * 32768); }
#else
#define MM_STAMP(k)
#define MM_ACC(who, k, t0)
#endif
// workgroup barrier that orders LDS only: __syncthreads() would also drain vmcnt, i.e. make the helpers wait for the
// global loads they have just put in flight for a later round, and every flush pass wait for its atomics
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define SRK_MM_THREADS 1024
#define SRK_MM_CW 12    // multiplying waves
#define SRK_MM_SLOTS 8  // tiles per multiplying wave
#define SRK_MM_LDW 208  // row stride of the round buffers (doubles): 13 tiles of 16 columns
typedef double srk_double4 __attribute__((ext_vector_type(4)));
// the K = 4 steps of one round for a wave whose tiles are anywhere in the grid (frame sets smaller than the full 13 x 13 tile
// grid: tiles dealt out row-major): every tile with its own A (row) and B (column) operand, in half steps of four tiles.  The
// operand addresses (lane part + wave-uniform tile offset) are formed again at every half step -- opaque to the compiler, which
// would otherwise keep all 16 of them in VGPRs -- so that a second operand set fits the 128 registers: the next half step's LDS
// reads are in flight while this one's four MFMAs issue.
__device__ __forceinline__ void schur_mm_steps(srk_double4 (&acc)[SRK_MM_SLOTS], const double* bw, const double* by,
                                               const int (&ta)[SRK_MM_SLOTS], const int (&tb)[SRK_MM_SLOTS], int lbase)
{
    static_assert(SRK_MM_SLOTS == 8, "two half steps of four tiles");
    // half step hs = (K step hs / 2, tiles 4 (hs & 1) .. + 3)
    auto load = [&](double (&a)[4], double (&b)[4], int hs) {
        int lb = lbase;
        asm volatile("" : "+v"(lb));
        const int ko = (hs >> 1) * 4 * SRK_MM_LDW, s0 = (hs & 1) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = bw[ko + lb + ta[s0 + s]];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = by[ko + lb + tb[s0 + s]];
    };
    auto mac = [&](const double (&a)[4], const double (&b)[4], int hs) {
        const int s0 = (hs & 1) * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s0 + s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc[s0 + s], 0, 0, 0);
    };
    // always the three K steps of a full round: in a short last round the k rows of the missing landmarks are zeros, and one
    // code path keeps the accumulators in place
    double a0[4], b0[4], a1[4], b1[4];
    load(a0, b0, 0);
    load(a1, b1, 1); mac(a0, b0, 0);
    load(a0, b0, 2); mac(a1, b1, 1);
    load(a1, b1, 3); mac(a0, b0, 2);
    load(a0, b0, 4); mac(a1, b1, 3);
    load(a1, b1, 5); mac(a0, b0, 4);
    mac(a1, b1, 5);
}
// the same for a wave whose eight tiles are a 2 x 4 BLOCK of the tile grid (tiles 0 .. 3: row ta[0], columns tb[0 .. 3];
// tiles 4 .. 7: row ta[4], the same columns): two A and four B operands serve the eight MFMAs of a K step -- 6 LDS reads
// instead of 16 -- and the operand sets are per K step (the next one's reads in flight under eight MFMAs)
__device__ __forceinline__ void schur_mm_steps_blk(srk_double4 (&acc)[SRK_MM_SLOTS], const double* bw, const double* by,
                                                   const int (&ta)[SRK_MM_SLOTS], const int (&tb)[SRK_MM_SLOTS], int lbase,
                                                   bool skip3 /* wave-uniform: tile 3 lies above the diagonal, never flushed */)
{
    struct Ops { double alo, ahi, b[4]; };
    auto load = [&](Ops& o, int ks) {
        int lb = lbase;
        asm volatile("" : "+v"(lb));
        const int ko = ks * 4 * SRK_MM_LDW;
        o.alo = bw[ko + lb + ta[0]];
        o.ahi = bw[ko + lb + ta[4]];
#pragma unroll
        for (int j = 0; j < 4; ++j) o.b[j] = by[ko + lb + tb[j]];
    };
    auto mac = [&](const Ops& o) {
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.alo, o.b[j], acc[j], 0, 0, 0);
        if (!skip3) acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.alo, o.b[3], acc[3], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ahi, o.b[j], acc[4 + j], 0, 0, 0);
    };
    Ops o0, o1;
#ifdef SRK_MM_FEWREADS // ablation only (wrong results): one operand set serves the three K steps -- what the reads cost the stream
    load(o0, 0);
    mac(o0); mac(o0); mac(o0);
    (void)o1;
#else
    load(o0, 0);
    load(o1, 1); mac(o0);
    load(o0, 2); mac(o1);
    mac(o0);
#endif
}
// nt = 13 (20 frames): the 91 tiles of the lower triangle over the 12 multiplying waves.  Nine waves take a 2 x 4 block
// (two tile rows, four tile columns; where the block reaches over the diagonal that tile is computed and not flushed),
// wave 9 the first eight tiles of the last row, waves 10 and 11 what is left.  {ti, tj} per slot, ti < 0: idle slot.
#define SRK_MM_TILES13_ROWS \
    { { 12, 0 }, { 12, 1 }, { 12, 2 }, { 12, 3 }, { 12, 4 }, { 12, 5 }, { 12, 6 }, { -1, 0 } }, \
    { { 12, 8 }, { 12, 9 }, { 12, 10 }, { 12, 11 }, { 12, 12 }, { 8, 8 }, { 9, 8 }, { 9, 9 } }, \
    { { 4, 4 }, { 5, 4 }, { 5, 5 }, { 0, 0 }, { 1, 0 }, { 1, 1 }, { 12, 7 }, { -1, 0 } }
__device__ const signed char srk_mm_tiles13[SRK_MM_CW][SRK_MM_SLOTS][2] = {
#define SRK_BLK(r, c) { { r, c }, { r, c + 1 }, { r, c + 2 }, { r, c + 3 }, { r + 1, c }, { r + 1, c + 1 }, { r + 1, c + 2 }, { r + 1, c + 3 } }
    SRK_BLK(10, 0), SRK_BLK(10, 4), SRK_BLK(10, 8), SRK_BLK(8, 0), SRK_BLK(8, 4), SRK_BLK(6, 0), SRK_BLK(6, 4), SRK_BLK(4, 0), SRK_BLK(2, 0),
#undef SRK_BLK
    SRK_MM_TILES13_ROWS,
};
// The three waves that are not blocks (9, 10, 11), uniform runs: their tile lists are known at compile time, and in the SYRK
// form the A operand of tile row t and the B operand of tile column t are the SAME 16 columns of Z -- one LDS read per
// DISTINCT tile index of the wave per K step (wave 9: 9 for its 8 MFMAs, wave 10: 5 for 8, wave 11: 4 for 6; the pattern
// code they ran before: 9-10, 13, 14 and a wait before every MFMA), operand sets per K step like the block waves, idle
// slots skipped.  Same products in the same order per accumulator: the sums keep their bits.
struct SrkRowWave {
    signed char t[16]; // distinct tile indices
    signed char ia[SRK_MM_SLOTS], ib[SRK_MM_SLOTS]; // slot -> operand (index into t), -1: idle slot
    int nt;
};
constexpr SrkRowWave srk_row_wave(int w)
{
    constexpr signed char tab[3][SRK_MM_SLOTS][2] = { SRK_MM_TILES13_ROWS };
    SrkRowWave o{};
    o.nt = 0;
    for (int s = 0; s < SRK_MM_SLOTS; ++s) {
        o.ia[s] = o.ib[s] = -1;
        if (tab[w][s][0] < 0) continue;
        for (int side = 0; side < 2; ++side) {
            const signed char v = tab[w][s][side];
            int k = 0;
            while (k < o.nt && o.t[k] != v) ++k;
            if (k == o.nt) o.t[o.nt++] = v;
            (side == 0 ? o.ia[s] : o.ib[s]) = (signed char)k;
        }
    }
    return o;
}
template <int W>
__device__ __forceinline__ void schur_mm_steps_row(srk_double4 (&acc)[SRK_MM_SLOTS], const double* bw, int lbase)
{
    constexpr SrkRowWave RW = srk_row_wave(W);
    struct Ops { double v[RW.nt]; };
    auto load = [&](Ops& o, int ks) {
        int lb = lbase;
        asm volatile("" : "+v"(lb));
        const int ko = ks * 4 * SRK_MM_LDW;
#pragma unroll
        for (int k = 0; k < RW.nt; ++k) o.v[k] = bw[ko + lb + 16 * RW.t[k]];
    };
    auto mac = [&](const Ops& o) {
#pragma unroll
        for (int s = 0; s < SRK_MM_SLOTS; ++s)
            if constexpr (true) {
                if (RW.ia[s] >= 0) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.v[RW.ia[s] < 0 ? 0 : RW.ia[s]], o.v[RW.ib[s] < 0 ? 0 : RW.ib[s]], acc[s], 0, 0, 0);
            }
    };
    Ops o0, o1;
    load(o0, 0);
    load(o1, 1); mac(o0);
    load(o0, 2); mac(o1);
    mac(o0);
}
// KIND 0: every run of the scene is uniform (all its landmarks see the same frames: the bench scenes); KIND 1: the scene has
// runs over the UNION of different frame lists (ragged feature tracks) -- a cell (landmark, frame slot) the landmark does not
// see is staged as a block of zeros; a uniform run is the full-mask case of that, so a scene with both kinds (the dino
// stand-in) is ONE launch.  Both kinds (and both storage precisions of W's 21 factors) take
// the SYRK form: with E = L L^T the sum W^T E^-1 W is Z^T Z, Z = L^-1 W -- ONE staged array, both MFMA operands read it.
// (Round 3 retired the W + Y form: W and Y = E^-1 W staged side by side, Y formed by a second pass over the staged W.)
//
// Roles.  12 multiplying waves own the <= 91 16 x 16 tiles on and below the diagonal of the 13 x 13 tile grid (8 tile slots = 64
// accumulator registers a wave); 4 helper waves stage Z: helper lane (m, cell) forms ROW m (point coordinate) of the cell's
// observation -- z[m][:] = (L^-1 Ap)[m] Af[:] + (L^-1 Bp)[m] Bf[:] from the 21 factors, two multiply-adds per entry -- and writes it as five 16-byte LDS stores; the right-hand side term Z^T (L^-1 g) accumulates
// in its registers.  The arena holds six rounds of four landmarks (round u in buffer u % 6); a barrier closes a DOUBLE round:
// rounds r, r + 1 are multiplied (six K = 4 steps a wave) while the helpers stage r + 2 (loads issued half a double round
// ago) and r + 3.  Rounds 0 and 1 are staged by the whole workgroup in the prologue, behind one barrier.  The flush adds every
// accumulator entry on or below the diagonal of the sum to S with fp64 atomics, straight from the registers.
// What was measured on the way (DESIGN sections 5 and 8): the helpers' instructions do not overlap with the MFMA streams of
// their SIMD, so what counts is how FEW instructions both sides issue; latency hiding on the helper side, barrier-free
// progress words, leaner block-wave address arithmetic each made it slower.
template <typename WT, int KIND, bool DET = false>
__global__ __launch_bounds__(SRK_MM_THREADS) void k_schur_mm(
    SrkDims d, double c, const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ obs_pt,
    const uint8_t* __restrict__ obs_slot, const uint32_t* __restrict__ pt_mask, const WT* __restrict__ W,
    const double* __restrict__ Vg, double* __restrict__ S, double* __restrict__ rhs,
    const int32_t* __restrict__ grp_first, const int32_t* __restrict__ grp_count, const int32_t* __restrict__ grp_nf,
    const int32_t* __restrict__ grp_frames, int32_t* __restrict__ irr /* [0]: count, [1 ..]: landmarks handed back */,
    double* __restrict__ det_stage /* DET: [run][SRK_DET_STRIDE], block (sa, sb) of the run's sum at (sa (sa + 1) / 2 + sb) 100 */,
    double* __restrict__ det_rhs /* DET: [run][SRK_DET_LD] */)
{
    constexpr int PB = SRK_GRP_PB;             // landmarks of a round
    constexpr int KR = 3 * PB;                 // k rows of a round
    constexpr int LDW = SRK_MM_LDW;            // row stride (doubles)
    constexpr int WB = KR * LDW;               // one round of Z
    constexpr int CAP = 6 * WB;                // six rounds (round u in buffer u % 6)
    constexpr int QMAX = PB * SRK_WS_NF;       // cells (landmark, frame slot) of a round
    constexpr int NH = SRK_MM_THREADS - 64 * SRK_MM_CW; // helper lanes
    constexpr bool masked = KIND == 1;
    constexpr int NV = SRK_WF_PLANES; // values of an observation a row lane keeps in flight: its 21 factors
    static_assert(SRK_WS_NF * 10 <= LDW && KR % 4 == 0, "round buffers");
    static_assert(SRK_MM_CW * SRK_MM_SLOTS >= 13 * 14 / 2, "tiles do not fit the multiplying waves");
    static_assert((SRK_GRP_MAXPTS + PB - 1) / PB < 64, "a run's row_ptr entries do not fit a wave");
    static_assert(3 * QMAX <= NH && 2 * 3 * QMAX <= SRK_MM_THREADS, "row lanes");
    (void)obs_pt; (void)obs_slot;
    __shared__ __attribute__((aligned(16))) double sBuf[CAP];
    __shared__ __attribute__((aligned(16))) double sE[SRK_GRP_MAXPTS][12]; // lower triangle of L^-1 by rows | L^-1 g
    __shared__ double sRhs[SRK_WS_NF * 10];
    __shared__ int32_t sVar[SRK_WS_NF * 10]; // row / column of S of the sum's row / column e; -1: a gauge-fixed variable
    // KIND 1: first observation (relative to the run's) and frame-slot mask of every landmark of the run
    __shared__ int32_t sOff[masked ? SRK_GRP_MAXPTS : 1];
    __shared__ uint32_t sMask[masked ? SRK_GRP_MAXPTS : 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef SRK_MM_STAMPS
    if (tid == 64 * SRK_MM_CW && blockIdx.x < 2048) for (int k = 10; k < 16; ++k) g_mm_stamps[blockIdx.x][k] = 0;
    if (tid == 0 && blockIdx.x < 2048) { g_mm_stamps[blockIdx.x][5] = g_mm_stamps[blockIdx.x][6] = g_mm_stamps[blockIdx.x][7] = 0; g_mm_stamps[blockIdx.x][8] = clock64(); }
    long long tacc = 0;
#endif
    MM_STAMP(0);
#ifndef SRK_MM_NO_STAGGER
    // Workgroups of equal length started together flush together: the fp64 atomics then queue at the memory side
    // (~1.9 TB/s chip-wide; stamps: 24 us per flush, 15 us when the flushes are spread out) while the matrix pipes
    // idle, and idle again while everybody multiplies.  When the grid is at least two rounds of workgroups, the first
    // workgroup of a CU starts up to ~55 us late, by its place among its XCD's first workgroups, and the later ones
    // inherit the offset (C3: Schur phase -2.7 %; tools/schur_ablate.sh with -DSRK_MM_NO_STAGGER).
    if (blockIdx.x < 256 && gridDim.x >= 512)
        for (int t = (blockIdx.x >> 3) & 15; t > 0; --t) __builtin_amdgcn_s_sleep(127);
#endif
    const int64_t p0 = grp_first[blockIdx.x];
    const int np = grp_count[blockIdx.x];
    const int nfu = grp_nf[blockIdx.x];
    if (KIND == 0 && nfu < 0) return; // (never launched on such a scene)
    const int nf = nfu < 0 ? -nfu : nfu;
    if (nf > SRK_WS_NF) return; // k_schur_grouped takes the wider runs
    if (tid < nf * 10) {
        sRhs[tid] = 0.0;
        const int64_t var = 10 * (int64_t)grp_frames[(int64_t)blockIdx.x * SRK_GRP_MAXNF + tid / 10] + tid % 10;
        sVar[tid] = srk_is_fixed_var(var, d) ? -1 : (int)var;
    }
    const int R = (np + PB - 1) / PB;
    const int nf10 = nf * 10;
    const int64_t o0 = row_ptr[p0];
    // LDS column of cell (staged landmark pl, frame slot a), Z row k = 3 pl + m:  (3 pl + m) LDW + 10 a + i
    // (masked) the cell (landmark pidx of the run, frame slot a): which observation, if any -- a landmark's observations are
    // its mask's set bits in slot order
    auto cell_obs = [&](int pidx, int a, bool& seen) -> unsigned {
        seen = false;
        if (pidx >= np) return 0u;
        const uint32_t mk = sMask[pidx];
        seen = (mk >> a) & 1u;
        return (unsigned)(o0 + sOff[pidx] + __builtin_popcount(mk & ((1u << a) - 1u)));
    };
    if constexpr (masked) {
        if (tid < np) {
            sOff[tid] = (int32_t)(row_ptr[p0 + tid] - o0);
            sMask[tid] = pt_mask[p0 + tid];
        }
        __syncthreads();
    }
    // what a row of an observation's Z needs from memory: the 21 factors (every plane base wave-uniform, one 32-bit lane offset
    // for all loads; float storage widens here)
    auto load_src = [&](double (&v)[NV], unsigned voff) {
#pragma unroll
        for (int k = 0; k < SRK_WF_PLANES; ++k) v[k] = (double)(W + (int64_t)k * d.Os)[voff];
    };
    // row m of Z = L^-1 W from those values and the landmark's sE row E; returns hm = (L^-1 g)[m]
    auto z_row = [&](const double (&v)[NV], const double* E, int m, double (&z)[10]) -> double {
        const int r3 = m * (m + 1) / 2; // row m of the lower triangle of L^-1: 1, 2 or 3 entries
        const double li0 = E[r3], li1 = m >= 1 ? E[r3 + 1] : 0.0, li2 = m >= 2 ? E[r3 + 2] : 0.0;
        const double am = li0 * v[SRK_WF_AP] + li1 * v[SRK_WF_AP + 1] + li2 * v[SRK_WF_AP + 2];
        const double bm = li0 * v[SRK_WF_BP] + li1 * v[SRK_WF_BP + 1] + li2 * v[SRK_WF_BP + 2];
        z[0] = am * v[SRK_WF_AF0];
        z[1] = bm * v[SRK_WF_BF1];
        z[2] = am * v[SRK_WF_G];
        z[3] = bm * v[SRK_WF_G];
#pragma unroll
        for (int i = 4; i < 10; ++i) z[i] = am * v[SRK_WF_AF4 + i - 4] + bm * v[SRK_WF_BF4 + i - 4];
        return E[6 + m];
    };
    {
        // ---- prologue: rounds 0 and 1 by the whole workgroup.  Thread (rd, sm, sq) of the first 2 * 3 * QMAX takes row sm of
        // cell sq of round rd -- its loads are in flight together with the 3x3 blocks' loads, Z leaves after ONE barrier
        const int rd = tid / (3 * QMAX), rem = tid - rd * (3 * QMAX);
        const int sm = rem / QMAX, sq = rem - sm * QMAX;
        const int pl = sq / nf, a = sq - pl * nf;
        bool row_on = false, cell = false; // cell: written in any case (zeros when the landmark does not see the frame)
        unsigned voff = 0;
        if constexpr (masked) {
            cell = rd < 2 && pl < PB;
            if (cell) voff = cell_obs(rd * PB + pl, a, row_on);
        } else if (rd < 2) {
            const int la = rd * PB < np ? rd * PB : np, lb = (rd + 1) * PB < np ? (rd + 1) * PB : np;
            const int oa = (int)(row_ptr[p0 + la] - o0), nq = (int)(row_ptr[p0 + lb] - o0) - oa;
            row_on = sq < nq;
            voff = (unsigned)(o0 + oa + sq);
        }
        double v[NV];
        if (row_on) load_src(v, voff);
        if (tid < np) {
            // E = L L^T.  A skipped landmark (|det| <= 1e-12, :1877-1881) stages zeros; one whose block passes that check but has
            // no positive pivots stages zeros as well and is handed back (irr) to the per-landmark inverse path, which
            // k_assemble's tail workgroups run.
            double Lc[6], hh[3];
            const int st = point_block_cholesky(Vg, d.Ns, p0 + tid, c, Lc, hh);
#pragma unroll
            for (int k = 0; k < 6; ++k) sE[tid][k] = st == 1 ? Lc[k] : 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) sE[tid][6 + m] = st == 1 ? hh[m] : 0.0;
            if (st == 2) irr[1 + atomicAdd(&irr[0], 1)] = (int32_t)(p0 + tid);
        }
        __syncthreads(); // sE; sRhs is zeroed
        double2* wp = reinterpret_cast<double2*>(sBuf + rd * WB + (3 * pl + sm) * LDW + 10 * a);
        if (row_on) {
            double z[10];
            const double hm = z_row(v, sE[rd * PB + pl], sm, z);
#pragma unroll
            for (int i = 0; i < 5; ++i) wp[i] = make_double2(z[2 * i], z[2 * i + 1]);
            if (!DET) {
#pragma unroll
                for (int i = 0; i < 10; ++i) atomicAdd(&sRhs[10 * a + i], z[i] * hm);
            }
        } else if (masked && cell) {
#pragma unroll
            for (int i = 0; i < 5; ++i) wp[i] = make_double2(0.0, 0.0);
        }
    }
    const int nt = (nf10 + 15) >> 4; // tile rows of the sum
    __syncthreads(); // sE, sVar, sRhs and Z of rounds 0 and 1 are visible
    if (DET && tid < nf10) {
        // deterministic mode: the right-hand-side terms of rounds 0 and 1 from the staged Z rows, in row order (no LDS atomics)
        double t0 = 0;
        for (int rd = 0; rd < 2; ++rd)
            for (int k = 0; k < KR; ++k)
                if (rd * PB + k / 3 < np) t0 = fma(sBuf[rd * WB + k * LDW + tid], sE[rd * PB + k / 3][6 + k % 3], t0);
        sRhs[tid] = t0;
    }
    MM_STAMP(1);
    // two roles, two code paths, one barrier sequence (one, then one per double round, one at the end)
    if (wv >= SRK_MM_CW) {
        // ---- helpers: lane h of 256 = QMAX m + cell
        const int h = tid - 64 * SRK_MM_CW;
        __builtin_amdgcn_s_setprio(2); // their few instructions must not queue behind the MFMA streams (round 3, measured:
                                       // without it the Schur phase takes 462 instead of 432-443 us, with the multiplying
                                       // waves raised above the helpers 460)
        const int rel = (int)(row_ptr[p0 + (lane * PB < np ? lane * PB : np)] - o0); // lane r: first observation of round r
        // Why rows (round 3): in-kernel stamps had the multiplying waves through a round's MFMAs after 2.0 us and then 1.3 us at
        // the barrier -- waiting for the HELPERS, whose vector-ALU instructions only issue in the gaps of their SIMD's three MFMA
        // streams (tools/ubench/mfma64_side.hip): what a round costs them is their instruction count.  A lane per COLUMN class
        // (the three coordinates of four columns) needed ~90 vector-ALU instructions a round; a row needs three multiply-adds
        // for its two scalars and two for each of its ten entries, and its ten entries leave as five 16-byte LDS writes: ~40.
        const int sm = h / QMAX, sq = h - sm * QMAX;
        const int pl = sq / nf, a = sq - pl * nf;
        const int dst = (3 * pl + sm) * LDW + 10 * a;
        double v[NV], racc[10];
        bool row_on = false, cell = false;
#pragma unroll
        for (int i = 0; i < 10; ++i) racc[i] = 0;
        auto load_round = [&](int r) { // global loads of round r (left in flight)
            unsigned voff;
            if constexpr (masked) {
                cell = sm < 3 && pl < PB;
                row_on = false;
                voff = cell ? cell_obs(r * PB + pl, a, row_on) : 0u;
            } else {
                const int ra = __builtin_amdgcn_readlane(rel, r), rb = __builtin_amdgcn_readlane(rel, r + 1);
                row_on = sm < 3 && sq < rb - ra;
                voff = (unsigned)(o0 + ra + sq);
            }
            if (row_on) load_src(v, voff);
        };
        auto stage_round = [&](int r, double* bw) {
            const int pb = r * PB;
            const int nb = np - pb < PB ? np - pb : PB;
            double2* wp = reinterpret_cast<double2*>(bw + dst); // (3 pl + m) LDW + 10 a: 16-byte aligned
            if (row_on) {
                double z[10];
                const double hm = z_row(v, sE[pb + pl], sm, z);
#pragma unroll
                for (int i = 0; i < 5; ++i) wp[i] = make_double2(z[2 * i], z[2 * i + 1]);
#pragma unroll
                for (int i = 0; i < 10; ++i) racc[i] = fma(z[i], hm, racc[i]);
            } else if (masked && cell) { // the landmark does not see this frame (or the round is short): a block of zeros
#pragma unroll
                for (int i = 0; i < 5; ++i) wp[i] = make_double2(0.0, 0.0);
            }
            // a short last round: the k rows of the landmarks it does not have must not carry an earlier round's data
            if (nb < PB)
                for (int t = h; t < (KR - 3 * nb) * LDW; t += NH) bw[3 * nb * LDW + t] = 0;
        };
        for (int u = 0; u < 2 && u < R; ++u) { // short rounds among the prologue's
            const int nb = np - u * PB < PB ? np - u * PB : PB;
            if (nb < PB)
                for (int t = h; t < (KR - 3 * nb) * LDW; t += NH) sBuf[u * WB + 3 * nb * LDW + t] = 0;
        }
        if (R > 2) load_round(2);
        lds_barrier(); // (the multiplying waves' first barrier)
#ifdef SRK_MM_STAMPS
        tacc = wall_clock64();
#endif
        // DOUBLE rounds: a barrier closes two rounds of four landmarks (six K = 4 steps of the multiplying waves).  Rounds r,
        // r + 1 are multiplied while the helpers stage r + 2 (loads issued half a double round ago) and r + 3 (loads issued at
        // the start of this one).
        for (int r = 0; r < R; r += 2) {
#ifdef SRK_MM_NOHELP // ablation only (wrong results): the helpers stage nothing after the prologue -- what the MFMA streams take alone
            if (d.N >= 0) { lds_barrier(); continue; }
#endif
            if (r + 2 < R) stage_round(r + 2, sBuf + ((r + 2) % 6) * WB);
#ifdef SRK_MM_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            MM_ACC(64 * SRK_MM_CW, 10, tacc);
            if (r + 3 < R) {
                load_round(r + 3);
                stage_round(r + 3, sBuf + ((r + 3) % 6) * WB);
            }
            if (r + 4 < R) load_round(r + 4);
            MM_ACC(64 * SRK_MM_CW, 11, tacc);
            lds_barrier(); // the products of rounds r, r + 1; Z of rounds r + 2, r + 3 are visible
            MM_ACC(64 * SRK_MM_CW, 12, tacc);
        }
        if (sm < 3 && pl < PB) {
            if (DET) { // every product is done (the loop's last barrier): the arena's first round buffer takes the lanes' sums
#pragma unroll
                for (int i = 0; i < 10; ++i) sBuf[dst + i] = racc[i];
            } else {
#pragma unroll
                for (int i = 0; i < 10; ++i) atomicAdd(&sRhs[10 * a + i], racc[i]);
            }
        }
    } else {
        // ---- multipliers: wave wv owns up to eight tiles of the nt x nt grid
        const int n_tiles = nt * (nt + 1) / 2;
        const int lr = lane & 15, lk = lane >> 4;
        // tile coordinates are wave-uniform: kept in scalar registers (readfirstlane), the lane part of an operand's
        // LDS address is one VGPR
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        const int lbase = lk * LDW + lr;
        int ta[SRK_MM_SLOTS], tb[SRK_MM_SLOTS];
        // The full-size grid (nt = 13: 20 frames, the bench scenes) is dealt out by the table srk_mm_tiles13: nine waves a
        // 2 x 4 block that shares its operands (schur_mm_steps_blk), three waves their own compile-time tile lists
        // (schur_mm_steps_row); a smaller grid row-major, tile u = wave + 12 s, every tile with its own operands.
        const bool runs = nt == 13, blk = runs && wvu < 9;
        unsigned onmask = 0; // slots whose sum is flushed
#pragma unroll
        for (int s = 0; s < SRK_MM_SLOTS; ++s) {
            int ti, tj;
            bool on;
            if (runs) {
                ti = srk_mm_tiles13[wvu][s][0];
                tj = srk_mm_tiles13[wvu][s][1];
                on = ti >= 0 && tj <= ti;
                if (ti < 0) ti = tj = 0;
            } else {
                const int u = wvu + SRK_MM_CW * s;
                on = u < n_tiles;
                const int uu = on ? u : 0;
                ti = (int)((sqrtf(8.0f * (float)uu + 1.0f) - 1.0f) * 0.5f);
                while ((ti + 1) * (ti + 2) / 2 <= uu) ++ti;
                while (ti * (ti + 1) / 2 > uu) --ti;
                tj = uu - ti * (ti + 1) / 2;
            }
            ta[s] = __builtin_amdgcn_readfirstlane(16 * ti);
            tb[s] = __builtin_amdgcn_readfirstlane(16 * tj);
            onmask |= on ? 1u << s : 0u;
        }
        onmask = __builtin_amdgcn_readfirstlane(onmask);
        srk_double4 acc[SRK_MM_SLOTS];
#pragma unroll
        for (int s = 0; s < SRK_MM_SLOTS; ++s) acc[s] = (srk_double4){ 0, 0, 0, 0 };
        lds_barrier();
        MM_STAMP(2);
#ifdef SRK_MM_STAMPS
        tacc = wall_clock64();
#endif
        // the three waves that are not 2 x 4 blocks (one each on SIMDs 1 .. 3) read more operands per MFMA: alone at the end of
        // a double round they would crawl -- they run ahead of the block waves instead (Schur phase 402.5 -> 397.5 us, six
        // interleaved runs; priority 1, 2 or 3, helpers at 2 or 3: the same)
        if (runs && !blk) __builtin_amdgcn_s_setprio(1);
        auto double_rounds = [&](auto steps) { // steps(bw): the three K steps of the round staged at bw
            for (int r = 0; r < R; r += 2) {
                const double* bw = sBuf + (r % 6) * WB;
#ifdef SRK_SCH_NOACC // (ablation, wrong results)
                if (d.N < 0)
#endif
                {
                    steps(bw);
                    if (r + 1 < R) steps(bw + WB);
                }
                MM_ACC(0, 6, tacc);
                lds_barrier();
                MM_ACC(0, 7, tacc);
            }
        };
        if (blk) {
            // a block that reaches over the diagonal (rows r, r + 1, columns r - 2 .. r + 1): its tile 3 = (r, r + 1) is never
            // flushed -- not multiplied either (with the table's tile counts the SIMDs hold 23, 23, 22, 23 MFMAs a K step)
            const bool skip3 = __builtin_amdgcn_readfirstlane((int)(tb[3] > ta[3])) != 0;
            double_rounds([&](const double* bw) { schur_mm_steps_blk(acc, bw, bw, ta, tb, lbase, skip3); });
        } else if (runs) {
#ifdef SRK_MM_NO_ROWWAVES // ablation (wrong results): the three non-block waves multiply nothing
            double_rounds([&](const double*) {});
#else
            if (wvu == 9) double_rounds([&](const double* bw) { schur_mm_steps_row<0>(acc, bw, lbase); });
            else if (wvu == 10) double_rounds([&](const double* bw) { schur_mm_steps_row<1>(acc, bw, lbase); });
            else double_rounds([&](const double* bw) { schur_mm_steps_row<2>(acc, bw, lbase); });
#endif
        } else {
            double_rounds([&](const double* bw) { schur_mm_steps(acc, bw, bw, ta, tb, lbase); }); // idle slots multiply tile (0, 0)
        }
        MM_STAMP(3);
        // flush.  f64 16x16x4 accumulator map: column = lane & 15, row = (lane >> 4) + 4 reg.
        // Straight from the accumulator registers: every lane adds its entries on and below the diagonal of the sum (row >=
        // column: S is lower-triangle authoritative).  The LDS flush this replaced -- passes of tile rows through the arena, two
        // barriers each, then one wave per row of S -- took 12.8 us of a workgroup's 98; its atomics themselves cost nothing
        // since the staggered start (SRK_SCH_NOFLUSH: 429.5 against 430 us), so what it spent was the shuffling.  A wave's
        // atomic instruction covers 16 consecutive columns of four rows of S.
        // (the four rows of a lane and their places in S are formed once per TILE ROW -- consecutive slots of a wave mostly share
        // it -- not once per entry: every vector-ALU instruction here is paid in MFMA time by the next workgroup's neighbours)
        int prev_ta = -1, rr4[4];
        int64_t rowoff[4];
        int detrow[4], detva[4]; // DET: frame slot and variable of the lane's rows
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) { rr4[reg] = -1; rowoff[reg] = 0; detrow[reg] = detva[reg] = 0; }
        double* const dstage = DET ? det_stage + (int64_t)blockIdx.x * SRK_DET_STRIDE : nullptr;
#pragma unroll
        for (int s = 0; s < SRK_MM_SLOTS; ++s) {
            if (!((onmask >> s) & 1u)) continue;
            if (ta[s] != prev_ta) { // wave-uniform
                prev_ta = ta[s];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int Rr = ta[s] + lk + 4 * reg;
                    const int rowv = Rr < nf10 ? sVar[Rr] : -1;
                    rr4[reg] = rowv >= 0 ? Rr : -1; // -1: no such row, or a gauge-fixed variable
                    rowoff[reg] = (int64_t)rowv * d.ld;
                    if (DET) detrow[reg] = Rr / 10, detva[reg] = Rr - 10 * (Rr / 10);
                }
            }
            const int Cc = tb[s] + lr;
            const int colv = Cc < nf10 ? sVar[Cc] : -1;
            if (colv < 0) continue;
            const int detsb = DET ? Cc / 10 : 0, detvb = DET ? Cc - 10 * (Cc / 10) : 0;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (Cc > rr4[reg]) continue; // above the diagonal, or no row
#ifdef SRK_SCH_NOFLUSH
                if (d.N >= 0) continue;
#endif
                if (DET) // the run's sum goes to its own staging block; k_schur_det_gather adds the runs' blocks in run order
                    dstage[(detrow[reg] * (detrow[reg] + 1) / 2 + detsb) * 100 + detva[reg] * 10 + detvb] = -acc[s][reg];
                else atomicAdd(&S[rowoff[reg] + colv], -acc[s][reg]);
            }
        }
    }
    lds_barrier(); // the helpers' sRhs adds are complete
    // rhs += sum F^T E^-1 g
    if (tid < nf10 && sVar[tid] >= 0) {
        if (DET) {
            double t1 = sRhs[tid];
            for (int k = 0; k < KR; ++k) t1 += sBuf[k * LDW + tid];
            det_rhs[(int64_t)blockIdx.x * SRK_DET_LD + tid] = t1;
        } else atomicAdd(&rhs[sVar[tid]], sRhs[tid]);
    }
    MM_STAMP(4);
#ifdef SRK_MM_STAMPS
    if (tid == 0 && blockIdx.x < 2048) g_mm_stamps[blockIdx.x][9] = clock64() - g_mm_stamps[blockIdx.x][8];
#endif
}

// deterministic mode, second pass: block (fa, fb) of the reduced camera system receives the runs' blocks in run order (ent:
// run | slot of fa << 20 | slot of fb << 25), the right-hand side of frame f the runs' terms in run order (run | slot << 20)
__global__ __launch_bounds__(128) void k_schur_det_gather(SrkDims d, const double* __restrict__ stage, const double* __restrict__ stage_rhs,
                                                          const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ pair_fa,
                                                          const int32_t* __restrict__ pair_fb, const int32_t* __restrict__ pair_ent,
                                                          int32_t n_pairs, const int32_t* __restrict__ f_ptr, const int32_t* __restrict__ f_ent,
                                                          double* __restrict__ S, double* __restrict__ rhs)
{
    const int e = threadIdx.x;
    if ((int)blockIdx.x < n_pairs) {
        const int p = blockIdx.x, fa = pair_fa[p], fb = pair_fb[p];
        if (e >= 100) return;
        const int va = e / 10, vb = e - 10 * va;
        if (fa == fb && vb > va) return;
        const int64_t row = 10 * (int64_t)fa + va, col = 10 * (int64_t)fb + vb;
        if (srk_is_fixed_var(row, d) || srk_is_fixed_var(col, d)) return;
        double sum = 0;
        for (int k = pair_ptr[p]; k < pair_ptr[p + 1]; ++k) {
            const unsigned u = (unsigned)pair_ent[k];
            const int run = u & 0xFFFFF, sa = (u >> 20) & 31, sb = (u >> 25) & 31;
            sum += stage[(int64_t)run * SRK_DET_STRIDE + (sa * (sa + 1) / 2 + sb) * 100 + e];
        }
        S[row * d.ld + col] += sum;
    } else {
        const int f = (int)blockIdx.x - n_pairs;
        if (f >= d.M || e >= 10) return;
        const int64_t row = 10 * (int64_t)f + e;
        if (srk_is_fixed_var(row, d)) return;
        double sum = 0;
        for (int k = f_ptr[f]; k < f_ptr[f + 1]; ++k) {
            const unsigned u = (unsigned)f_ent[k];
            sum += stage_rhs[(int64_t)(u & 0xFFFFF) * SRK_DET_LD + 10 * (u >> 20) + e];
        }
        rhs[row] += sum;
    }
}

// ------------------------------------------------------------------ K3l: long tracks as frame-block pairs (fp64 MFMA)
// Landmarks seen by more than SRK_GRP_MAXNF frames (the demos' all-visible scenes: 36 and 60 frames; the MVF driver
// calls the path with every track in every frame) used to go through the per-landmark kernel k_schur: one workgroup
// per landmark, nf (nf + 1) / 2 blocks of global fp64 atomics each -- 9 ms an attempt on the 60-frame / 3321-point MVF
// flagfile scene.  The run sum as a matrix product (k_schur_mm) generalises when the frame set of a run is cut into
// blocks of SRK_LONG_FB = 8 frames (80 columns = five 16-column MFMA tiles exactly): a workgroup takes ONE pair
// (bi >= bj) of frame blocks of a run of <= SRK_LONG_PTS landmarks,
//     sum_i W_i[:, bi]^T E_i^-1 W_i[:, bj]   =   Wl[:, bi]^T Yl[:, bj],     an 80 x 80 x (3 np) product,
// with rows k = (landmark, point coordinate) as in k_schur_mm.  Five waves, wave w owns tile row w of the 5 x 5 tile
// grid (one A operand read serves five MFMAs); every lane also stages: rounds of eight landmarks (24 k rows = six
// K = 4 steps), lane (side, landmark, frame, half block) loads 15 entries of its observation's W block -- all three
// point coordinates, so it forms its part of Y = E^-1 W in registers -- with the next round's loads in flight during
// this round's MFMAs; two LDS buffers, one LDS-only barrier a round.  The run's frame set is a union (a landmark that
// misses a frame contributes zeros there); the table `run_obs` (host) maps (landmark, frame slot) to the observation.
// Pairs on the diagonal compute all 25 tiles and flush the lower block triangle; they also carry the block's part of
// the right-hand side.  The sums leave as fp64 atomics, once per pair and run.
#define SRK_LONG_RB 8     // landmarks per round
// per block size FB (8 or 16 frames): NC = 10 FB columns = NC / 16 MFMA tiles exactly, one wave per tile row; the LDS row stride
// is 16 mod 32 doubles so that the four k rows of an operand hit different banks (80; 176 for 160 columns)
template <int FB> struct LongCfg {
    static constexpr int NC = 10 * FB, NT = NC / 16, THREADS = 64 * NT, LD = NC % 32 == 16 ? NC : NC + 16;
    static_assert(NC % 16 == 0 && LD % 32 == 16, "a frame block is a whole number of MFMA tiles");
    static_assert(2 * SRK_LONG_RB * FB * 2 <= THREADS && SRK_LONG_PTS_HOST <= THREADS && 2 * NC <= THREADS, "staging lanes");
};
template <typename WT, int FB>
__global__ __launch_bounds__(LongCfg<FB>::THREADS) void k_schur_long(
    SrkDims d, double c, const WT* __restrict__ W, const double* __restrict__ Vg, double* __restrict__ S,
    double* __restrict__ rhs, const int32_t* __restrict__ item /* [n][4]: run, bi, bj, - */,
    const int32_t* __restrict__ run_np, const int32_t* __restrict__ run_nf,
    const int32_t* __restrict__ run_pts /* [run][SRK_LONG_PTS] */, const int32_t* __restrict__ run_frames /* [run][SRK_LONG_MAXNF] */,
    const int64_t* __restrict__ run_obs_off, const int32_t* __restrict__ run_obs /* [off + landmark * nfp + slot] or -1 */)
{
    using Cfg = LongCfg<FB>;
    constexpr int NC = Cfg::NC, NT = Cfg::NT, LD = Cfg::LD, RB = SRK_LONG_RB, KR = 3 * RB, BUF = KR * LD;
    constexpr int HALF = RB * FB * 2; // staging lanes of a side: (landmark of the round, frame of the block, half of its ten columns)
    __shared__ __attribute__((aligned(16))) double sW[2][BUF];
    __shared__ __attribute__((aligned(16))) double sY[2][BUF];
    __shared__ __attribute__((aligned(16))) double sE[SRK_LONG_PTS_HOST][12];
    __shared__ double sRhs[NC];
    __shared__ int32_t sVar[2][NC]; // row / column of S of the pair's row / column e; -1: gauge-fixed or beyond the run's frames
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int run = item[4 * blockIdx.x], bi = item[4 * blockIdx.x + 1], bj = item[4 * blockIdx.x + 2];
    const int np = run_np[run], nf = run_nf[run];
    const int nfp = FB * ((nf + FB - 1) / FB);
    const bool diag = bi == bj;
    const int32_t* pts = run_pts + (int64_t)run * SRK_LONG_PTS_HOST;
    const int32_t* obs = run_obs + run_obs_off[run];
    if (tid < 2 * NC) {
        const int side = tid / NC, e = tid - side * NC;
        const int slot = FB * (side ? bj : bi) + e / 10;
        int var = -1;
        if (slot < nf) {
            const int64_t v = 10 * (int64_t)run_frames[(int64_t)run * SRK_LONG_MAXNF_HOST + slot] + e % 10;
            var = srk_is_fixed_var(v, d) ? -1 : (int)v;
        }
        sVar[side][e] = var;
    }
    if (tid < NC) sRhs[tid] = 0.0;
    if (tid < np) { // 3x3 damped block inverses; a singular block contributes nothing (:1877-1881)
        double Einv[9], g[3];
        const bool ok = point_block_inverse(Vg, d.Ns, pts[tid], c, Einv, g);
#pragma unroll
        for (int k = 0; k < 9; ++k) sE[tid][k] = ok ? Einv[k] : 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m)
            sE[tid][9 + m] = ok ? Einv[3 * m] * g[0] + Einv[3 * m + 1] * g[1] + Einv[3 * m + 2] * g[2] : 0.0;
    }
    // staging lane: side 0 = block bi (-> W, and Y too on a diagonal pair), side 1 = block bj (-> Y)
    const int side = tid / HALF, within = tid - side * HALF;
    const int spl = within / (2 * FB), sa = (within >> 1) % FB, sh = within & 1;
    const bool stager = tid < 2 * HALF && (side == 0 || !diag);
    const int sslot = FB * (side ? bj : bi) + sa;
    const int sdst = 3 * spl * LD + 10 * sa + 5 * sh;
    double pre[15];
    auto load_round = [&](int r) { // global loads of round r into `pre` (left in flight); zeros where nothing is observed
        const int pl = r * RB + spl;
        int64_t o = -1;
        if (stager && pl < np) o = obs[(int64_t)pl * nfp + sslot];
#pragma unroll
        for (int i = 0; i < 15; ++i) pre[i] = 0.0;
        if (o >= 0) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int i = 0; i < 5; ++i) pre[5 * m + i] = w_entry<WT>(W, d.Os, o, 10 * m + 5 * sh + i);
        }
    };
    double racc[5] = { 0, 0, 0, 0, 0 };
    auto stage_round = [&](int r, int b) { // `pre` -> W / Y of round r in buffer b
        if (!stager) return;
        const int pl = r * RB + spl < np ? r * RB + spl : 0; // (a landmark past the run's end staged zeros)
        if (side == 0) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int i = 0; i < 5; ++i) sW[b][sdst + m * LD + i] = pre[5 * m + i];
        }
        if (side == 1 || diag) {
            const double* E = sE[pl];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    sY[b][sdst + m * LD + i] = E[3 * m] * pre[i] + E[3 * m + 1] * pre[5 + i] + E[3 * m + 2] * pre[10 + i];
        }
        if (diag) { // rhs += F^T E^-1 g (:1895-1897), this block's columns
#pragma unroll
            for (int i = 0; i < 5; ++i) racc[i] += pre[i] * sE[pl][9] + pre[5 + i] * sE[pl][10] + pre[10 + i] * sE[pl][11];
        }
    };
    const int R = (np + RB - 1) / RB;
    load_round(0);
    __syncthreads(); // sE, sVar, sRhs
    const int lr = lane & 15, lk = lane >> 4;
    // a diagonal pair needs the tiles on and below its diagonal only (S is lower-triangle authoritative; what a tile above it
    // would add to the upper halves of the 10 x 10 diagonal blocks it cuts is never read)
    const int tmax = diag ? __builtin_amdgcn_readfirstlane(wv) : NT - 1;
    srk_double4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (srk_double4){ 0, 0, 0, 0 };
    for (int r = 0; r < R; ++r) {
        const int b = r & 1;
        stage_round(r, b);
        if (r + 1 < R) load_round(r + 1);
        lds_barrier(); // round r is staged; every wave is done with round r - 1 (the buffer round r + 1 overwrites)
        const double* bw = sW[b] + lk * LD + 16 * wv + lr;
        const double* by = sY[b] + lk * LD + lr;
#pragma unroll
        for (int ks = 0; ks < KR / 4; ++ks) {
            const double a = bw[4 * ks * LD];
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += 5) { // five B operands in flight at a time
                double bb[5];
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    if (t0 + t <= tmax) bb[t] = by[4 * ks * LD + 16 * (t0 + t)];
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    if (t0 + t <= tmax) acc[t0 + t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb[t], acc[t0 + t], 0, 0, 0);
            }
        }
    }
    // flush.  f64 16x16x4 accumulator map: column = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t > tmax) continue;
        const int col = 16 * t + lr, vc = sVar[1][col];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 16 * wv + lk + 4 * reg, vr = sVar[0][row];
            if (vr < 0 || vc < 0) continue;
            if (diag && col / 10 > row / 10) continue; // lower block triangle (whole diagonal blocks, as the other kernels)
            atomicAdd(&S[(int64_t)vr * d.ld + vc], -acc[t][reg]);
        }
    }
    if (diag) {
        if (stager) {
#pragma unroll
            for (int i = 0; i < 5; ++i) atomicAdd(&sRhs[10 * sa + 5 * sh + i], racc[i]);
        }
        __syncthreads();
        if (tid < NC && sVar[0][tid] >= 0) atomicAdd(&rhs[sVar[0][tid]], sRhs[tid]);
    }
}

void srk_launch_schur_long(hipStream_t s, const SrkDims& d, double c, const double* W, const double* Vg, double* S, double* rhs,
                           const int32_t* item, int64_t n_items, const int32_t* run_np, const int32_t* run_nf,
                           const int32_t* run_pts, const int32_t* run_frames, const int64_t* run_obs_off, const int32_t* run_obs, int fb)
{
    if (n_items <= 0) return;
#define SRK_LONG_LAUNCH(FBV)                                                                                                       \
    do {                                                                                                                           \
        if (d.w_f32)                                                                                                               \
            hipLaunchKernelGGL((k_schur_long<float, FBV>), dim3((unsigned)n_items), dim3(LongCfg<FBV>::THREADS), 0, s, d, c,         \
                               reinterpret_cast<const float*>(W), Vg, S, rhs, item, run_np, run_nf, run_pts, run_frames, run_obs_off, \
                               run_obs);                                                                                           \
        else                                                                                                                       \
            hipLaunchKernelGGL((k_schur_long<double, FBV>), dim3((unsigned)n_items), dim3(LongCfg<FBV>::THREADS), 0, s, d, c, W, Vg, S, \
                               rhs, item, run_np, run_nf, run_pts, run_frames, run_obs_off, run_obs);                               \
    } while (0)
    if (fb == 16) SRK_LONG_LAUNCH(16);
    else SRK_LONG_LAUNCH(8);
#undef SRK_LONG_LAUNCH
}

void srk_launch_schur_grouped(hipStream_t s, const SrkDims& d, double c, const int64_t* row_ptr, const int32_t* obs_pt,
                              const uint8_t* obs_slot, const uint32_t* pt_mask, const double* W, const double* Vg, double* S,
                              double* rhs, const int32_t* grp_first, const int32_t* grp_count, const int32_t* grp_nf,
                              const int32_t* grp_frames, int64_t n_groups, int64_t n_wide, int64_t n_mid,
                              int fp32_accumulate, int32_t* irr, int64_t n_mm_uniform, int64_t n_mm_ragged, const SrkDetSchur* det)
{
    if (n_groups <= 0) return;
    // runs over at most SRK_WS_NF frames go to the MFMA kernel, fp64 only (the opt-in fp32 accumulation keeps the packed
    // FMA register-tile kernel).  SRK_SCHUR_NO_WS / SRK_SCHUR_VALU: development switches back to the register-tile kernels.
#ifdef SRK_DEV
    static const bool env_no_ws = getenv("SRK_SCHUR_NO_WS") != nullptr;
    static const bool env_valu = getenv("SRK_SCHUR_VALU") != nullptr; // development: the register-tile kernel k_schur_ws
#else
    const bool env_no_ws = false, env_valu = false;
#endif
    const bool no_ws = env_no_ws || fp32_accumulate;
    const int nf_skip = no_ws ? 0 : SRK_WS_NF;
#define SRK_SCHUR_ARGS(WP) d, c, row_ptr, obs_pt, obs_slot, pt_mask, WP, Vg, S, rhs, grp_first, grp_count, grp_nf, grp_frames
    const dim3 grid((unsigned)n_groups), block(SRK_GRP_THREADS);
    const float* Wf = reinterpret_cast<const float*>(W);
    // every kernel in two instantiations: W stored as double, or as float (srk_ba_set_storage_precision; loads widen)
#define SRK_SCHUR_LAUNCH(KERNEL, BLOCK, ...)                                                                        \
    do {                                                                                                            \
        if (d.w_f32) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ float>), grid, BLOCK, 0, s, SRK_SCHUR_ARGS(Wf));        \
        else hipLaunchKernelGGL((KERNEL<__VA_ARGS__ double>), grid, BLOCK, 0, s, SRK_SCHUR_ARGS(W));               \
    } while (0)
#define SRK_SCHUR_LAUNCH_SKIP(KERNEL, ...)                                                                          \
    do {                                                                                                            \
        if (d.w_f32) hipLaunchKernelGGL((KERNEL<__VA_ARGS__ float>), grid, block, 0, s, SRK_SCHUR_ARGS(Wf), nf_skip); \
        else hipLaunchKernelGGL((KERNEL<__VA_ARGS__ double>), grid, block, 0, s, SRK_SCHUR_ARGS(W), nf_skip);       \
    } while (0)
    if (!no_ws && n_wide + n_mid < n_groups) { // runs over at most SRK_WS_NF frames: the MFMA kernel
        if (env_valu) SRK_SCHUR_LAUNCH(k_schur_ws, block, double, );
        else {
#define SRK_MM_LAUNCH(KIND)                                                                                                      \
    do {                                                                                                                         \
        if (det) {                                                                                                               \
            if (d.w_f32) hipLaunchKernelGGL((k_schur_mm<float, KIND, true>), grid, dim3(SRK_MM_THREADS), 0, s, SRK_SCHUR_ARGS(Wf), irr, det->stage, det->stage_rhs); \
            else hipLaunchKernelGGL((k_schur_mm<double, KIND, true>), grid, dim3(SRK_MM_THREADS), 0, s, SRK_SCHUR_ARGS(W), irr, det->stage, det->stage_rhs);         \
        } else {                                                                                                                 \
            if (d.w_f32) hipLaunchKernelGGL((k_schur_mm<float, KIND, false>), grid, dim3(SRK_MM_THREADS), 0, s, SRK_SCHUR_ARGS(Wf), irr, nullptr, nullptr);          \
            else hipLaunchKernelGGL((k_schur_mm<double, KIND, false>), grid, dim3(SRK_MM_THREADS), 0, s, SRK_SCHUR_ARGS(W), irr, nullptr, nullptr);                  \
        }                                                                                                                        \
    } while (0)
            if (n_mm_ragged > 0) SRK_MM_LAUNCH(1);
            else if (n_mm_uniform > 0) SRK_MM_LAUNCH(0);
#undef SRK_MM_LAUNCH
            if (det && n_mm_ragged + n_mm_uniform > 0)
                hipLaunchKernelGGL(k_schur_det_gather, dim3((unsigned)(det->n_pairs + d.M)), dim3(128), 0, s, d, det->stage, det->stage_rhs,
                                   det->pair_ptr, det->pair_fa, det->pair_fb, det->pair_ent, det->n_pairs, det->f_ptr, det->f_ent, S, rhs);
        }
    }
    if (no_ws ? n_wide < n_groups : n_mid > 0) { // (SRK_WS_NF <) frames <= SRK_GRP_NF1: one half block per thread
        if (fp32_accumulate) SRK_SCHUR_LAUNCH_SKIP(k_schur_grouped, 1, float, );
        else SRK_SCHUR_LAUNCH_SKIP(k_schur_grouped, 1, double, );
    }
    if (n_wide > 0) { // more than SRK_GRP_NF1 frames: two half blocks per thread
        if (fp32_accumulate) SRK_SCHUR_LAUNCH_SKIP(k_schur_grouped, 2, float, );
        else SRK_SCHUR_LAUNCH_SKIP(k_schur_grouped, 2, double, );
    }
#undef SRK_SCHUR_LAUNCH
#undef SRK_SCHUR_LAUNCH_SKIP
#undef SRK_SCHUR_ARGS
}

// G (block diagonal of the frame blocks, diagonal * (1+c), gauge rows/cols dropped) is added after the landmark
// sums; fixed variables and padding rows get an identity diagonal (:1780-1823, :1902-1908).
// `ident`: the identity diagonal of fixed / padding variables (1; 0 on all ranks but one when the assembled systems
// of several landmark shards are summed afterwards).
// The last SRK_ASM_TAIL workgroups serve the landmarks k_schur_mm's SYRK form handed back (a point block that passes the
// determinant check without positive pivots; irr[0] = how many, normally none): the per-landmark inverse path.
#define SRK_ASM_TAIL 8
template <typename WT>
__global__ __launch_bounds__(256) void k_assemble(SrkDims d, double c, const double* __restrict__ Ug, double* __restrict__ S,
                                                  double* __restrict__ rhs, double ident, const int64_t* __restrict__ row_ptr,
                                                  const int32_t* __restrict__ obs_frame, const WT* __restrict__ W,
                                                  const double* __restrict__ Vg, const int32_t* __restrict__ irr)
{
    if (blockIdx.x >= gridDim.x - SRK_ASM_TAIL) {
        const int n = irr ? irr[0] : 0;
        if (n <= 0) return;
        __shared__ SchurLmScratch sm;
        for (int li = (int)(blockIdx.x - (gridDim.x - SRK_ASM_TAIL)); li < n; li += SRK_ASM_TAIL)
            schur_one_landmark<WT>(d, c, row_ptr, obs_frame, W, Vg, S, rhs, irr[1 + li], sm);
        return;
    }
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nblk = (int64_t)d.M * 110;
    if (t < nblk) {
        int64_t j = t / 110;
        int e = (int)(t - j * 110);
        const double* u = Ug + j * SRK_UG;
        if (e < 100) {
            int v1 = e / 10, v2 = e - v1 * 10;
            int64_t row = 10 * j + v1, col = 10 * j + v2;
            bool fr = srk_is_fixed_var(row, d), fc = srk_is_fixed_var(col, d);
            if (fr || fc) {
                if (v1 == v2) S[row * d.ld + col] = ident;
            } else {
                int a = v1 < v2 ? v1 : v2, b = v1 < v2 ? v2 : v1;
                double val = u[a * 10 - a * (a - 1) / 2 + (b - a)];
                if (v1 == v2) val *= 1 + c;
                S[row * d.ld + col] += val;
            }
        } else {
            int v = e - 100;
            int64_t row = 10 * j + v;
            if (srk_is_fixed_var(row, d)) rhs[row] = 0.0;
            else rhs[row] -= u[55 + v];
        }
    } else {
        int64_t p = 10 * (int64_t)d.M + (t - nblk);
        if (p < d.ld) {
            S[p * d.ld + p] = ident;
            rhs[p] = 0.0;
        }
    }
}

void srk_launch_assemble(hipStream_t s, const SrkDims& d, double c, const double* Ug, double* S, double* rhs, double ident,
                         const int64_t* row_ptr, const int32_t* obs_frame, const double* W, const double* Vg, const int32_t* irr)
{
    int64_t n = (int64_t)d.M * 110 + (d.ld - 10 * (int64_t)d.M);
    const dim3 grid((unsigned)((n + 255) / 256 + SRK_ASM_TAIL));
    if (d.w_f32)
        hipLaunchKernelGGL(k_assemble<float>, grid, dim3(256), 0, s, d, c, Ug, S, rhs, ident, row_ptr, obs_frame,
                           reinterpret_cast<const float*>(W), Vg, irr);
    else hipLaunchKernelGGL(k_assemble<double>, grid, dim3(256), 0, s, d, c, Ug, S, rhs, ident, row_ptr, obs_frame, W, Vg, irr);
}

// mirror the lower triangle into the upper one (downloads / exchange of the full matrix)
__global__ void k_symmetrize(int64_t n, int64_t ld, double* __restrict__ S)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t r = blockIdx.y;
    if (c < n && r < n && c > r) S[r * ld + c] = S[c * ld + r];
}
void srk_launch_symmetrize(hipStream_t s, int64_t n, int64_t ld, double* S)
{
    hipLaunchKernelGGL(k_symmetrize, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s, n, ld, S);
}

// ------------------------------------------------------------------ K5: back-substitution + landmark update
// thread per observation: t = W_ij dc_j (3-vector), segmented wave reduction by landmark, one atomic per segment
template <typename WT>
__global__ __launch_bounds__(256) void k_backsub_obs(SrkDims d, const int32_t* __restrict__ obs_frame,
                                                     const int32_t* __restrict__ obs_pt, const WT* __restrict__ W,
                                                     const double* __restrict__ dc, double* __restrict__ acc)
{
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int lane = threadIdx.x & (WAVE - 1);
    int32_t pt = -1;
    double t[3] = { 0, 0, 0 };
    if (o < d.O) {
        pt = obs_pt[o];
        const double* x = dc + 10 * (int64_t)obs_frame[o];
        double xv[10];
#pragma unroll
        for (int fv = 0; fv < 10; ++fv) xv[fv] = x[fv];
        {
            // W x = Ap (Af . x) + Bp (Bf . x): 21 loads and 22 multiply-adds instead of 30 and 30
            const WT* wp = W + o;
            const double gq = wp[(int64_t)SRK_WF_G * d.Os];
            double sa = wp[(int64_t)SRK_WF_AF0 * d.Os] * xv[0] + gq * xv[2];
            double sb = wp[(int64_t)SRK_WF_BF1 * d.Os] * xv[1] + gq * xv[3];
#pragma unroll
            for (int fv = 4; fv < 10; ++fv) {
                sa += wp[(int64_t)(SRK_WF_AF4 + fv - 4) * d.Os] * xv[fv];
                sb += wp[(int64_t)(SRK_WF_BF4 + fv - 4) * d.Os] * xv[fv];
            }
#pragma unroll
            for (int pv = 0; pv < 3; ++pv)
                t[pv] = wp[(int64_t)(SRK_WF_AP + pv) * d.Os] * sa + wp[(int64_t)(SRK_WF_BP + pv) * d.Os] * sb;
        }
    }
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        int32_t okey = __shfl_down(pt, off, WAVE);
        bool take = (lane + off < WAVE) && (okey == pt);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double other = __shfl_down(t[k], off, WAVE);
            if (take) t[k] += other;
        }
    }
    int32_t prev = __shfl_up(pt, 1, WAVE);
    bool head = (lane == 0) || (prev != pt);
    if (head && pt >= 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&acc[(int64_t)k * d.Ns + pt], t[k]);
    }
}

// dx_i = -E^-1 (F_i dc + g_i)  (:1951), zero when E is not invertible (:1939-1943); X_trial = X + dx (:2012-2016)
__global__ void k_point_update(SrkDims d, double c, const double* __restrict__ Vg, double* __restrict__ acc,
                               const double* __restrict__ pts, double* __restrict__ pts_trial, double* __restrict__ dx,
                               int* __restrict__ info)
{
    int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= d.N) return;
    double Einv[9], g[3];
    double dxi[3] = { 0, 0, 0 };
    if (point_block_inverse(Vg, d.Ns, pt, c, Einv, g)) {
        double b0 = acc[pt] + g[0], b1 = acc[d.Ns + pt] + g[1], b2 = acc[2 * d.Ns + pt] + g[2];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            dxi[m] = -(Einv[3 * m] * b0 + Einv[3 * m + 1] * b1 + Einv[3 * m + 2] * b2);
            if (!isfinite(dxi[m])) atomicOr(info, 2); // :1953-1954
        }
    }
    acc[pt] = acc[d.Ns + pt] = acc[2 * d.Ns + pt] = 0.0; // consumed: k_backsub_obs of the slot's next attempt adds from zero
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        dx[3 * pt + m] = dxi[m];
        pts_trial[3 * pt + m] = pts[3 * pt + m] + dxi[m];
    }
}

void srk_launch_backsub(hipStream_t s, const SrkDims& d, double c, const int32_t* obs_frame, const int32_t* obs_pt,
                        const double* W, const double* Vg, const double* dc, double* acc, const double* pts,
                        double* pts_trial, double* dx)
{
    // acc must be zero on entry (the caller's memset, or k_point_update of the slot's previous attempt); the info word is
    // the int right behind acc
    if (d.O > 0) {
        int64_t blocks = (d.O + 255) / 256;
        if (d.w_f32)
            hipLaunchKernelGGL(k_backsub_obs<float>, dim3((unsigned)blocks), dim3(256), 0, s, d, obs_frame, obs_pt,
                               reinterpret_cast<const float*>(W), dc, acc);
        else
            hipLaunchKernelGGL(k_backsub_obs<double>, dim3((unsigned)blocks), dim3(256), 0, s, d, obs_frame, obs_pt, W, dc, acc);
    }
    if (d.N > 0) {
        int* info = reinterpret_cast<int*>(acc + 3 * d.Ns);
        hipLaunchKernelGGL(k_point_update, dim3((unsigned)((d.N + 255) / 256)), dim3(256), 0, s, d, c, Vg, acc, pts,
                           pts_trial, dx, info);
    }
}

// ------------------------------------------------------------------ K6: camera update
// T_direct += dT ; R_direct <- Rodrigues(dW) R_direct unless |dW| ~ 0 ; store the inverse pose (:2021-2062, :59-92)
__global__ void k_cam_apply(int32_t M, const double* __restrict__ R, const double* __restrict__ T,
                            const double* __restrict__ dc, double* __restrict__ Rn, double* __restrict__ Tn,
                            const double* __restrict__ K, double f0, double* __restrict__ pack /* of the new pose, or null */)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double* r = R + 9 * (int64_t)j;
    const double* t = T + 3 * (int64_t)j;
    const double* x = dc + 10 * (int64_t)j;
    double Rd[9], Td[3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Rd[3 * a + b] = r[3 * b + a];
    for (int a = 0; a < 3; ++a) Td[a] = -(Rd[3 * a] * t[0] + Rd[3 * a + 1] * t[1] + Rd[3 * a + 2] * t[2]);
    Td[0] += x[4]; Td[1] += x[5]; Td[2] += x[6];
    double w0 = x[7], w1 = x[8], w2 = x[9];
    double ang = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    double Rnew[9];
    // IsClose(0, ang): |ang| <= 1e-8 + 1e-5 * |max(0, ang)|  (approx-alg.h:8-16, obs-geom.cpp:555-556)
    bool zero = fabs(0.0 - ang) <= (1.0e-8 + 1.0e-5 * fabs(ang > 0.0 ? ang : 0.0));
    if (zero) {
        for (int i = 0; i < 9; ++i) Rnew[i] = Rd[i];
    } else {
        double d0 = w0 / ang, d1 = w1 / ang, d2 = w2 / ang;
        double sn = sin(ang), cs = cos(ang);
        double Kx[9] = { 0, -d2, d1, d2, 0, -d0, -d1, d0, 0 };
        double KK[9], rot[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                KK[3 * a + b] = Kx[3 * a] * Kx[b] + Kx[3 * a + 1] * Kx[3 + b] + Kx[3 * a + 2] * Kx[6 + b];
        for (int i = 0; i < 9; ++i) rot[i] = ((i == 0 || i == 4 || i == 8) ? 1.0 : 0.0) + sn * Kx[i] + (1 - cs) * KK[i];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                Rnew[3 * a + b] = rot[3 * a] * Rd[b] + rot[3 * a + 1] * Rd[3 + b] + rot[3 * a + 2] * Rd[6 + b];
    }
    double* ro = Rn + 9 * (int64_t)j;
    double* to = Tn + 3 * (int64_t)j;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) ro[3 * a + b] = Rnew[3 * b + a];
    for (int a = 0; a < 3; ++a) to[a] = -(ro[3 * a] * Td[0] + ro[3 * a + 1] * Td[1] + ro[3 * a + 2] * Td[2]);
    if (pack) { // the same values k_cam_pack would read back
        double rl[9], tl[3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) rl[3 * a + b] = Rnew[3 * b + a];
        for (int a = 0; a < 3; ++a) tl[a] = -(rl[3 * a] * Td[0] + rl[3 * a + 1] * Td[1] + rl[3 * a + 2] * Td[2]);
        cam_pack_one(rl, tl, K + 9 * (int64_t)j, f0, pack + (int64_t)SRK_CAM_PACK * j);
    }
}

void srk_launch_cam_apply(hipStream_t s, int32_t M, const double* R, const double* T, const double* dc, double* Rn,
                          double* Tn, const double* K, double f0, double* pack)
{
    hipLaunchKernelGGL(k_cam_apply, dim3((M + 63) / 64), dim3(64), 0, s, M, R, T, dc, Rn, Tn, K, f0, pack);
}

// ------------------------------------------------------------------ K1: reprojection error
#define SRK_ERR_BLOCKS 1024

__global__ __launch_bounds__(256) void k_error(SrkDims d, const double* __restrict__ pts,
                                               const double* __restrict__ cam, const int32_t* __restrict__ obs_frame,
                                               const int32_t* __restrict__ obs_pt, const double* __restrict__ obs_uv,
                                               double* __restrict__ partial)
{
    __shared__ double red[4];
    double sum = 0;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < d.O; o += (int64_t)gridDim.x * 256) {
        int32_t pt = obs_pt[o];
        const double* c = cam + (int64_t)SRK_CAM_PACK * obs_frame[o];
        double2 uv = reinterpret_cast<const double2*>(obs_uv)[o];
        const double* X = pts + 3 * (int64_t)pt;
        double X0 = X[0], X1 = X[1], X2 = X[2];
        double xc0 = c[0] * X0 + c[1] * X1 + c[2] * X2 + c[9];
        double xc1 = c[3] * X0 + c[4] * X1 + c[5] * X2 + c[10];
        double xc2 = c[6] * X0 + c[7] * X1 + c[8] * X2 + c[11];
        double p = c[12] * xc0 + c[13] * xc1 + c[14] * xc2;
        double q = c[15] * xc0 + c[16] * xc1 + c[17] * xc2;
        double r = c[18] * xc0 + c[19] * xc1 + c[20] * xc2;
        double f0 = c[47];
        double ex = p / r - uv.x / f0, ey = q / r - uv.y / f0;
        sum += ex * ex + ey * ey;
    }
    sum = wave_sum(sum);
    int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = sum;
    lds_barrier();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Staged variant: when the host found that every run of SRK_JF_OBS consecutive observations touches at most
// SRK_JF_SLOTS consecutive frames (the fused Jacobian's condition; wg_jmin = first frame of each run), the 22 camera
// values an observation needs come from LDS instead of 22 gathers through L1 -- the gathers, not HBM, bound k_error.
// Same arithmetic, same per-observation order inside a thread; the partial sums are per run.
__global__ __launch_bounds__(256) void k_error_staged(SrkDims d, const double* __restrict__ pts,
                                                      const double* __restrict__ cam, const int32_t* __restrict__ obs_frame,
                                                      const int32_t* __restrict__ obs_pt, const double* __restrict__ obs_uv,
                                                      const int32_t* __restrict__ wg_jmin, double* __restrict__ partial)
{
    __shared__ double sCam[SRK_JF_SLOTS][23]; // R, T, K (0..20), f0; odd stride: conflict-free across frames
    __shared__ double red[4];
    const int jmin = wg_jmin[blockIdx.x];
    {
        const int nfr = d.M - jmin < SRK_JF_SLOTS ? d.M - jmin : SRK_JF_SLOTS;
        for (int t = threadIdx.x; t < nfr * 22; t += 256) {
            const int js = t / 22, e = t - js * 22;
            sCam[js][e] = cam[(int64_t)SRK_CAM_PACK * (jmin + js) + (e < 21 ? e : 47)];
        }
    }
    lds_barrier();
    const int64_t o_first = (int64_t)blockIdx.x * SRK_JF_OBS;
    double sum = 0;
#pragma unroll
    for (int ch = 0; ch < SRK_JF_CHUNKS; ++ch) {
        const int64_t o = o_first + ch * 256 + threadIdx.x;
        if (o >= d.O) break;
        const double* c = sCam[obs_frame[o] - jmin];
        double2 uv = reinterpret_cast<const double2*>(obs_uv)[o];
        const double* X = pts + 3 * (int64_t)obs_pt[o];
        double X0 = X[0], X1 = X[1], X2 = X[2];
        double xc0 = c[0] * X0 + c[1] * X1 + c[2] * X2 + c[9];
        double xc1 = c[3] * X0 + c[4] * X1 + c[5] * X2 + c[10];
        double xc2 = c[6] * X0 + c[7] * X1 + c[8] * X2 + c[11];
        double p = c[12] * xc0 + c[13] * xc1 + c[14] * xc2;
        double q = c[15] * xc0 + c[16] * xc1 + c[17] * xc2;
        double r = c[18] * xc0 + c[19] * xc1 + c[20] * xc2;
        double f0 = c[21];
        double ex = p / r - uv.x / f0, ey = q / r - uv.y / f0;
        sum += ex * ex + ey * ey;
    }
    sum = wave_sum(sum);
    int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = sum;
    lds_barrier();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// scoring variant (MultiViewIterativeFactorizer::ReprojError, multi-view-factorization.cpp:415-475): observations whose
// homogeneous image point has |z| <= z_tol are skipped (:455-457) and the summands are counted; z_tol < 0 keeps all.
// partial[0 .. grid) = error sums, partial[grid .. 2 grid) = counts.
__global__ __launch_bounds__(256) void k_error_score(int64_t O, const double* __restrict__ pts,
                                                     const double* __restrict__ cam, const int32_t* __restrict__ obs_frame,
                                                     const int32_t* __restrict__ obs_pt, const double* __restrict__ obs_uv,
                                                     double z_tol, double* __restrict__ partial)
{
    __shared__ double red[8];
    double sum = 0, cnt = 0;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < O; o += (int64_t)gridDim.x * 256) {
        const double* c = cam + (int64_t)SRK_CAM_PACK * obs_frame[o];
        double2 uv = reinterpret_cast<const double2*>(obs_uv)[o];
        const double* X = pts + 3 * (int64_t)obs_pt[o];
        double X0 = X[0], X1 = X[1], X2 = X[2];
        double xc0 = c[0] * X0 + c[1] * X1 + c[2] * X2 + c[9];
        double xc1 = c[3] * X0 + c[4] * X1 + c[5] * X2 + c[10];
        double xc2 = c[6] * X0 + c[7] * X1 + c[8] * X2 + c[11];
        double p = c[12] * xc0 + c[13] * xc1 + c[14] * xc2;
        double q = c[15] * xc0 + c[16] * xc1 + c[17] * xc2;
        double r = c[18] * xc0 + c[19] * xc1 + c[20] * xc2;
        if (fabs(r) <= z_tol) continue;
        double f0 = c[47];
        double ex = p / r - uv.x / f0, ey = q / r - uv.y / f0;
        sum += ex * ex + ey * ey;
        cnt += 1.0;
    }
    sum = wave_sum(sum);
    cnt = wave_sum(cnt);
    int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = sum, red[4 + wave] = cnt;
    lds_barrier();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        partial[gridDim.x + blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

// fixed-order final sum: the LM accept/reject decision must not depend on atomic arrival order
// info / info2 given: {solver info, point-update info} go next to the error scalar (one 24-byte read-back per LM attempt)
// and are cleared, ready for the slot's next attempt (no memset launches in the LM loop)
__global__ __launch_bounds__(256) void k_error_final(int32_t n, const double* __restrict__ partial,
                                                     double* __restrict__ out, int* __restrict__ info,
                                                     int* __restrict__ info2)
{
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    lds_barrier();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        lds_barrier();
    }
    if (threadIdx.x == 0) {
        out[0] = red[0];
        if (info) {
            out[1] = (double)info[0];
            out[2] = (double)info2[0];
            info[0] = 0;
            info2[0] = 0;
        }
    }
}

int32_t srk_error_partials(const SrkDims& d)
{
    int64_t blocks = (d.O + 255) / 256;
    if (blocks < 1) blocks = 1;
    return (int32_t)(blocks < SRK_ERR_BLOCKS ? blocks : SRK_ERR_BLOCKS);
}
int64_t srk_error_partials_staged(const SrkDims& d) { return d.O > 0 ? (d.O + SRK_JF_OBS - 1) / SRK_JF_OBS : 1; }

void srk_launch_error(hipStream_t s, const SrkDims& d, const double* pts, const double* cam,
                      const int32_t* obs_frame, const int32_t* obs_pt, const double* obs_uv, double* partial,
                      int32_t n_partial, double* err_out, const int32_t* wg_jmin, int* info, int* info2)
{
    if (wg_jmin && d.O > 0) { // staged cameras: one partial sum per run of SRK_JF_OBS observations
        const int64_t nb = srk_error_partials_staged(d);
        hipLaunchKernelGGL(k_error_staged, dim3((unsigned)nb), dim3(256), 0, s, d, pts, cam, obs_frame, obs_pt, obs_uv, wg_jmin,
                           partial);
        hipLaunchKernelGGL(k_error_final, dim3(1), dim3(256), 0, s, (int32_t)nb, partial, err_out, info, info2);
        return;
    }
    hipLaunchKernelGGL(k_error, dim3((unsigned)n_partial), dim3(256), 0, s, d, pts, cam, obs_frame, obs_pt, obs_uv,
                       partial);
    hipLaunchKernelGGL(k_error_final, dim3(1), dim3(256), 0, s, n_partial, partial, err_out, info, info2);
}

void srk_launch_error_score(hipStream_t s, int64_t O, const double* pts, const double* cam, const int32_t* obs_frame,
                            const int32_t* obs_pt, const double* obs_uv, double z_tol, double* partial /* 2 n */,
                            int32_t n_partial, double* out2)
{
    hipLaunchKernelGGL(k_error_score, dim3((unsigned)n_partial), dim3(256), 0, s, O, pts, cam, obs_frame, obs_pt, obs_uv,
                       z_tol, partial);
    hipLaunchKernelGGL(k_error_final, dim3(1), dim3(256), 0, s, n_partial, partial, out2, (int*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL(k_error_final, dim3(1), dim3(256), 0, s, n_partial, partial + n_partial, out2 + 1, (int*)nullptr,
                       (int*)nullptr);
}


// ------------------------------------------------------------------ skyline (envelope) helpers for the RCS
// The reduced camera system is non-zero only where two frames share a landmark.  env_col[t] is the first column
// (multiple of 256) that can be non-zero in the 128-row tile row t; Cholesky fill stays inside that skyline.
// zero / pack / unpack touch only rows' segments [env_col[t], 128 (t + 1)) -- the lower triangle inside the skyline.
__global__ __launch_bounds__(256) void k_env_zero(int64_t ld, const int64_t* __restrict__ env_col, double* __restrict__ S,
                                                  double* __restrict__ rhs /* zeroed too when given */,
                                                  int32_t* __restrict__ irr /* k_schur_mm's hand-back counter, cleared too */)
{
    int64_t t = blockIdx.y;
    if (irr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) irr[0] = 0;
    int64_t c0 = env_col[t], c1 = 128 * (t + 1);
    int64_t row = 128 * t + blockIdx.x;
    if (rhs && threadIdx.x == 0) rhs[row] = 0.0;
    double* p = S + row * ld;
    for (int64_t c = c0 + 2 * threadIdx.x; c < c1; c += 512) *reinterpret_cast<double2*>(p + c) = make_double2(0.0, 0.0);
}
void srk_launch_env_zero(hipStream_t s, int64_t ld, const int64_t* env_col, double* S, double* rhs, int32_t* irr)
{
    hipLaunchKernelGGL(k_env_zero, dim3(128, (unsigned)(ld / 128)), dim3(256), 0, s, ld, env_col, S, rhs, irr);
}

// pack: out[env_off[t] + r * w + (c - c0)] = S[row][c], w = c1 - c0 ; dir = 0 pack, 1 unpack
__global__ __launch_bounds__(256) void k_env_pack(int64_t ld, const int64_t* __restrict__ env_col,
                                                  const int64_t* __restrict__ env_off, double* __restrict__ S,
                                                  double* __restrict__ packed, int dir)
{
    int64_t t = blockIdx.y;
    int64_t c0 = env_col[t], c1 = 128 * (t + 1), w = c1 - c0;
    int64_t r = blockIdx.x;
    double* ps = S + (128 * t + r) * ld + c0;
    double* pp = packed + env_off[t] + r * w;
    for (int64_t c = threadIdx.x; c < w; c += 256) {
        if (dir == 0) pp[c] = ps[c];
        else ps[c] = pp[c];
    }
}
void srk_launch_env_pack(hipStream_t s, int64_t ld, const int64_t* env_col, const int64_t* env_off, double* S,
                         double* packed, int dir)
{
    hipLaunchKernelGGL(k_env_pack, dim3(128, (unsigned)(ld / 128)), dim3(256), 0, s, ld, env_col, env_off, S, packed,
                       dir);
}

// Exchange format of the ASSEMBLED system between landmark shards: before the factorisation row r of the lower triangle
// is non-zero only in [band_col[r], r] (10 x the first covisible frame), which is ~3x tighter than the 256-aligned
// skyline the factorisation fills.  out[band_off[r] + (c - band_col[r])] = S[r][c]; dir = 0 pack, 1 unpack.
__global__ __launch_bounds__(256) void k_band_pack(int64_t ld, const int64_t* __restrict__ band_col,
                                                   const int64_t* __restrict__ band_off, double* __restrict__ S,
                                                   double* __restrict__ packed, int dir)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // one wave per row
    if (r >= ld) return;
    const int64_t c0 = band_col[r], w = r - c0 + 1;
    double* ps = S + r * ld + c0;
    double* pp = packed + band_off[r];
    for (int64_t c = threadIdx.x & 63; c < w; c += 64) {
        if (dir == 0) pp[c] = ps[c];
        else ps[c] = pp[c];
    }
}
void srk_launch_band_pack(hipStream_t s, int64_t ld, const int64_t* band_col, const int64_t* band_off, double* S,
                          double* packed, int dir)
{
    hipLaunchKernelGGL(k_band_pack, dim3((unsigned)((ld + 3) / 4)), dim3(256), 0, s, ld, band_col, band_off, S, packed, dir);
}

// checksum of a buffer, {sum, sum of magnitudes}, in a fixed order (256 contiguous pieces, a fixed tree inside each, a fixed
// tree over the pieces): the self-check of the first damping-parallel round (srk_ba_host.hip: dp_selfcheck)
__global__ __launch_bounds__(256) void k_checksum_part(const double* __restrict__ p, int64_t n, double* __restrict__ part)
{
    __shared__ double ss[256], sa[256];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    double s = 0, a = 0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const double v = p[i];
        s += v;
        a += fabs(v);
    }
    ss[threadIdx.x] = s, sa[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) ss[threadIdx.x] += ss[threadIdx.x + w], sa[threadIdx.x] += sa[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[2 * blockIdx.x] = ss[0], part[2 * blockIdx.x + 1] = sa[0];
}
__global__ __launch_bounds__(256) void k_checksum_final(const double* __restrict__ part, double* __restrict__ out2)
{
    __shared__ double ss[256], sa[256];
    ss[threadIdx.x] = part[2 * threadIdx.x], sa[threadIdx.x] = part[2 * threadIdx.x + 1];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) ss[threadIdx.x] += ss[threadIdx.x + w], sa[threadIdx.x] += sa[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out2[0] = ss[0], out2[1] = sa[0];
}
void srk_launch_checksum(hipStream_t s, const double* p, int64_t n, double* part /* 512 doubles of scratch */, double* out2)
{
    hipLaunchKernelGGL(k_checksum_part, dim3(256), dim3(256), 0, s, p, n, part);
    hipLaunchKernelGGL(k_checksum_final, dim3(1), dim3(256), 0, s, part, out2);
}

// ------------------------------------------------------------------ multi-view-factorization steps
// Estimate3DPointDepthFromFrames (multi-view-factorization.cpp:223-253, MASKS 8.44), one thread per track: observation
// 0 of a track is its base frame, frame_from_base = SE3AFromB(frame_i_from_world, base_from_world) (:205-213).
__global__ __launch_bounds__(256) void k_mvf_depth(int64_t n, const int64_t* __restrict__ row_ptr,
                                                   const int32_t* __restrict__ frame, const double* __restrict__ x,
                                                   const double* __restrict__ R, const double* __restrict__ T,
                                                   double* __restrict__ depth)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t o0 = row_ptr[i], o1 = row_ptr[i + 1];
    if (o1 - o0 < 2) { depth[i] = __builtin_nan(""); return; }
    const double* Rb = R + 9 * (int64_t)frame[o0];
    const double* Tb = T + 3 * (int64_t)frame[o0];
    const double x10 = x[3 * o0], x11 = x[3 * o0 + 1], x12 = x[3 * o0 + 2];
    double num = 0, den = 0;
    for (int64_t o = o0 + 1; o < o1; ++o) {
        const double* Ri = R + 9 * (int64_t)frame[o];
        const double* Ti = T + 3 * (int64_t)frame[o];
        double Tr[3], v[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double r0 = Ri[3 * a] * Rb[0] + Ri[3 * a + 1] * Rb[1] + Ri[3 * a + 2] * Rb[2];
            double r1 = Ri[3 * a] * Rb[3] + Ri[3 * a + 1] * Rb[4] + Ri[3 * a + 2] * Rb[5];
            double r2 = Ri[3 * a] * Rb[6] + Ri[3 * a + 1] * Rb[7] + Ri[3 * a + 2] * Rb[8];
            Tr[a] = Ti[a] - (r0 * Tb[0] + r1 * Tb[1] + r2 * Tb[2]);
            v[a] = r0 * x10 + r1 * x11 + r2 * x12;
        }
        const double xi0 = x[3 * o], xi1 = x[3 * o + 1], xi2 = x[3 * o + 2];
        const double h10 = xi1 * Tr[2] - xi2 * Tr[1], h11 = xi2 * Tr[0] - xi0 * Tr[2], h12 = xi0 * Tr[1] - xi1 * Tr[0];
        const double h20 = xi1 * v[2] - xi2 * v[1], h21 = xi2 * v[0] - xi0 * v[2], h22 = xi0 * v[1] - xi1 * v[0];
        num += h10 * h20 + h11 * h21 + h12 * h22;
        den += h10 * h10 + h11 * h11 + h12 * h12;
    }
    depth[i] = 1.0 / (-num / den);
}

// FindRelativeMotionMultiPoints (:107-189): Gram matrix A^T A (12 x 12, 78 unique sums) of the 3P x 12 system whose
// rows are [x2]x (kron(x1^T, I) | alpha I); the right singular vector of A's smallest singular value is the
// eigenvector of the smallest eigenvalue of A^T A (host, 12 x 12).  One block per 256 points, fixed-order sums.
__global__ __launch_bounds__(256) void k_mvf_gram(int64_t P, const double* __restrict__ xa, const double* __restrict__ xt,
                                                  const double* __restrict__ depth, double* __restrict__ partial /* [grid][78] */)
{
    __shared__ double red[4][78];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double row[3][12];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 12; ++c) row[r][c] = 0;
    if (i < P) {
        const double c1[3] = { xa[3 * i], xa[3 * i + 1], xa[3 * i + 2] };
        const double c2[3] = { xt[3 * i], xt[3 * i + 1], xt[3 * i + 2] };
        const double sk[3][3] = { { 0, -c2[2], c2[1] }, { c2[2], 0, -c2[0] }, { -c2[1], c2[0], 0 } };
        const double alpha = 1.0 / depth[i];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int comp = 0; comp < 3; ++comp)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) row[r][3 * comp + cc] = c1[comp] * sk[r][cc];
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) row[r][9 + cc] = alpha * sk[r][cc];
        }
    }
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    int e = 0;
#pragma unroll
    for (int a = 0; a < 12; ++a)
#pragma unroll
        for (int b = a; b < 12; ++b) {
            double g = row[0][a] * row[0][b] + row[1][a] * row[1][b] + row[2][a] * row[2][b];
            g = wave_sum(g);
            if (lane == 0) red[wave][e] = g;
            ++e;
        }
    lds_barrier();
    if (threadIdx.x < 78)
        partial[(int64_t)blockIdx.x * 78 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

void srk_launch_mvf_depth(hipStream_t s, int64_t n, const int64_t* row_ptr, const int32_t* frame, const double* x,
                          const double* R, const double* T, double* depth)
{
    if (n > 0) hipLaunchKernelGGL(k_mvf_depth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, row_ptr, frame, x, R, T, depth);
}
void srk_launch_mvf_gram(hipStream_t s, int64_t P, const double* xa, const double* xt, const double* depth, double* partial)
{
    if (P > 0) hipLaunchKernelGGL(k_mvf_gram, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, xa, xt, depth, partial);
}
