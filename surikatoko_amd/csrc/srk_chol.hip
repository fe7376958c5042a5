// srk_chol.hip -- dense SPD solve of the reduced camera system on gfx950 (fp64).
//
// Replaces `decomp_lin_sys_left_side.householderQr().solve(rhs)` (bundle-adj-kanatani.cpp:1911).  The reduced
// camera system is symmetric positive definite under the multiplicative LM damping, so a blocked Cholesky is used.
// A non-positive or non-finite pivot sets *info (the caller maps it to the reference's "solve failed" path).
//
// Two-level right-looking blocking (A row-major ld x ld, LOWER triangle authoritative, overwritten by L,
// ld % 256 == 0):
//   outer panel  = 256 columns.  Its trailing update  C -= P P^T  (K = 256) is the only large contraction on the
//                  whole BA path (n^3/3 flops) and runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64):
//                  128x128 tile per workgroup, 64x64 per wave (4x4 accumulator tiles), K streamed through a
//                  double-buffered LDS ring in chunks of 16.  K = 256 makes it 32 flop per HBM byte of C traffic
//                  (a 64-deep update is HBM-bound at 8 flop/B).
//   inner panel  = 64 columns (4 per outer panel).  Default: k_step256 runs the four inner panels of an outer step as
//                  ONE launch whose workgroups hand factored tiles to one another (roles, hand-off protocol and the
//                  deadlock argument: at the kernel).  The launch sequence it replaces stays (development switch,
//                  repeat after a hand-off timeout, launches above 256 workgroups) and is bit-identical: k_panel
//                  factorises the 64x64 diagonal tile (every workgroup redundantly, in registers + LDS), solves its
//                  rows of the panel by substitution and folds the forward substitution of the right-hand side in;
//                  k_upd64 applies the 64-deep update to the remaining columns of the outer panel (MFMA as well).
// The per-outer-panel row limit `row_end` lets the caller skip the structurally zero part of a banded / skyline
// system (the envelope of a Cholesky factor equals the envelope of the matrix); dense = ld for every panel.
//
// Every factorisation kernel works on a BATCH of matrices (CholBatch, blockIdx.z = item): a banded system is cut into
// chunks with separators between them (nested dissection, srk_chol_solve_chunked further down), the chunks of a level
// advance in lock step through one launch sequence, and the separator system is chunked again.  A plain solve is a
// batch of one.  The solver has no atomics and a fixed summation order (bit-reproducible).
#include "srk_dev.hpp"
#include <cstdlib>

#define NB 64
#define NBO SRK_CHOL_NB      // 256, outer panel
#define TL 128               // trailing-update tile
#define KC 16                // trailing-update K chunk
#define KCP (KC + 2)         // padded LDS row (18 doubles: conflict-free ds_read_b64 for the MFMA operand map)
#define LDSP (NB + 2)

typedef double double4_t __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. it waits
// for every outstanding global load / store of the wave.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------- 64x64 diagonal tile, register-resident
// thread t owns row i = t>>2, columns c = 4m + (t&3), m = 0..15.  Outer-product Cholesky with deferred scaling:
// at step j the pivot d_j = a_jj is final and a_ic -= a_ij a_cj / d_j for c > j; column j is published through a
// double-buffered LDS vector (one barrier per step).  On exit sD holds L (lower), sInv[j] = 1 / L_jj.
__device__ __forceinline__ double fast_rcp(double d)
{
    // v_rcp_f64 seed + two Newton steps (full double precision; an IEEE divide costs ~35 dependent instructions and
    // sits on the per-pivot critical path)
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// broadcast lane (4 * quad + K) of every quad to its four lanes: two DPP moves (quad_perm = [K,K,K,K])
template <int K> __device__ __forceinline__ double quad_bcast(double v)
{
    constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_bcast_dyn(double v, int k)
{
    switch (k & 3) {
    case 0: return quad_bcast<0>(v);
    case 1: return quad_bcast<1>(v);
    case 2: return quad_bcast<2>(v);
    default: return quad_bcast<3>(v);
    }
}

__device__ __forceinline__ double fast_rsqrt(double d)
{
    double r = __builtin_amdgcn_rsq(d);
    // Newton for 1/sqrt: r <- r * (1.5 - 0.5 d r^2), twice
    r = r * fma(-0.5 * d * r, r, 1.5);
    r = r * fma(-0.5 * d * r, r, 1.5);
    return r;
}

// Four pivots per barrier pair.  A per-pivot scheme pays one LDS round trip + barrier (~150 cycles) and one reciprocal
// chain per pivot (~340 cycles in all, 64 times).  Here the columns are taken in groups of four (the four columns a
// quad owns in one register slot):
//   A. every thread publishes its entry of the group's 64 x 4 panel; after the barrier all threads read the group's
//      4 x 4 diagonal block and factor it redundantly in registers (4 dependent reciprocals, no LDS in between);
//   B. every quad gathers its row of the panel with DPP broadcasts, eliminates it against that 4 x 4 factor (final,
//      unscaled entries y_i and multipliers l_i = y_i / d), publishes y_i; after the second barrier each thread reads
//      y_c for its column-rows c and applies the rank-4 update a_ic -= sum_k l_ik y_ck.
// The slot of the NEXT group is updated and published first, so its barrier round trip hides under the other updates.
// Entries above the diagonal are updated too (never read): no per-element predicates.
// (round 3) The rank-4 update of the slots behind the next one used to sit between the next panel's publication and its
// barrier: every wave reached that barrier ~200 cycles (on average) later than the panel needed.  Those updates are now
// DEFERRED behind the barrier and issued among the dependent instructions of the next group's 4 x 4 factorisation (independent
// work for that latency-bound chain); the finals Y are double-buffered for it.  Every entry still receives the groups'
// updates in the same order with the same operands: results are bit-identical to the former order.
// The reciprocals of the 4 x 4 factor take ONE Newton step on the v_rcp_f64 seed (tools/ubench/rcp_acc.hip on gfx950: seed
// 4.6e-8, one step 2.1e-15, two steps 1.1e-16 relative): they only scale the multipliers of a rank-1 term, i.e. act as a
// relative perturbation 2e-15 of that term -- an order of magnitude below the n eps |L| |L^T| rounding bound of a 256-column
// block; everything a hand-off partner recomputes (1 / L_tt of the stored factor) keeps the two-step reciprocal.
__device__ __forceinline__ double fast_rcp1(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    return fma(fma(-d, r, 1.0), r, r);
}
#ifndef SRK_POTRF_RCP
#define SRK_POTRF_RCP fast_rcp1
#endif
// PUB (the diagonal-block workgroup of k_step256, whose other waves carry passenger rows): the threads also publish every
// group's 4 x 4 factor (multipliers l_jk = u_jk / d_k below the diagonal, 1 / d_k) in sF [2][64] before the group's SECOND barrier, and 1 / sqrt(d) in sRsq.
template <bool PUB>
__device__ __forceinline__ bool potrf64(double (*sD)[NB + 2], double* sPY /*[3][64][4]: panel P, finals Y (two buffers)*/,
                                        double* sDiag /*[64]*/, double* sInv /*[64]*/, double* sF, double* sRsq)
{
    const int t = threadIdx.x, i = t >> 2, q = t & 3;
    double (*sP)[4] = reinterpret_cast<double (*)[4]>(sPY);
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = sD[i][4 * m + q];
    // pivot health as two running scalars (a per-pivot flag would keep all 64 pivots live until the end):
    // dmin <= 0 catches non-positive pivots, 0 * d turns Inf / NaN into NaN
    double dmin = 1.0, dchk = 0.0;
    sP[i][q] = a[0];
    lds_barrier();
    double zp0 = 0, zp1 = 0, zp2 = 0, zp3 = 0; // multipliers of the previous group: its deferred updates
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        double (*sY)[4] = reinterpret_cast<double (*)[4]>(sPY + 4 * NB * (1 + (g & 1)));      // finals of this group
        // ---- A: the 4 x 4 diagonal block of the group (rows 4g .. 4g+3 of the panel), LDL^T with unscaled entries
        const double2* Dp = reinterpret_cast<const double2*>(&sP[4 * g][0]);
        const double2 d0 = Dp[0], d1 = Dp[2], d2a = Dp[4], d2b = Dp[5], d3a = Dp[6], d3b = Dp[7];
        // ---- the previous group's rank-4 update of the slots behind this one (deferred: independent of the chain below)
        if (g > 0) {
            double (*sYp)[4] = reinterpret_cast<double (*)[4]>(sPY + 4 * NB * (1 + ((g - 1) & 1)));
#pragma unroll
            for (int m = g + 1; m < 16; ++m) {
                const double2* yp = reinterpret_cast<const double2*>(&sYp[4 * m + q][0]);
                const double2 y01 = yp[0], y23 = yp[1];
                a[m] = fma(-zp3, y23.y, fma(-zp2, y23.x, fma(-zp1, y01.y, fma(-zp0, y01.x, a[m]))));
            }
        }
#ifndef SRK_POTRF_SEQ_PIVOTS
        // The four pivots in TWO reciprocal stages instead of four (round 4): the second pivot of a pair is a quotient of
        // determinants, u11 = (d00 d11 - d10^2) / d00, so 1 / u11 = d00 / det starts beside 1 / d00 instead of behind it (the
        // subtraction cancels exactly as d11 - d10^2 / d00 does: same relative error eps d11 / u11); likewise 1 / u33 beside
        // 1 / u22 on the Schur complement of the first pair.  A group's chain is bound by these dependent reciprocals.
        const double u00 = d0.x, u10 = d1.x, u20 = d2a.x, u30 = d3a.x;
        const double det01 = fma(u00, d1.y, -(u10 * u10));
        const double r0 = SRK_POTRF_RCP(u00);
        const double r1 = u00 * SRK_POTRF_RCP(det01);
        const double l20 = u20 * r0, l30 = u30 * r0;
        const double u21 = fma(-l20, u10, d2a.y), u31 = fma(-l30, u10, d3a.y);
        const double l21 = u21 * r1, l31 = u31 * r1;
        const double u22 = fma(-l21, u21, fma(-l20, u20, d2b.x));
        const double u32 = fma(-l31, u21, fma(-l30, u20, d3b.x));
        const double u33p = fma(-l31, u31, fma(-l30, u30, d3b.y)); // u33 before the last pivot's term
        const double det23 = fma(u22, u33p, -(u32 * u32));
        const double r2 = SRK_POTRF_RCP(u22);
        const double r3 = u22 * SRK_POTRF_RCP(det23);
        // pivot health: u11 and u33 have the sign of their determinants (given the pivot before them is positive)
        dmin = fmin(fmin(dmin, u00), fmin(det01, fmin(u22, det23)));
        dchk = fma(0.0, u00, fma(0.0, det01, fma(0.0, u22, fma(0.0, det23, dchk))));
#else
        const double u00 = d0.x;
        const double r0 = SRK_POTRF_RCP(u00);
        const double u10 = d1.x, u20 = d2a.x, u30 = d3a.x;
        const double l10 = u10 * r0, l20 = u20 * r0, l30 = u30 * r0;
        const double u11 = fma(-l10, u10, d1.y);
        const double r1 = SRK_POTRF_RCP(u11);
        const double u21 = fma(-l20, u10, d2a.y), u31 = fma(-l30, u10, d3a.y);
        const double l21 = u21 * r1, l31 = u31 * r1;
        const double u22 = fma(-l21, u21, fma(-l20, u20, d2b.x));
        const double r2 = SRK_POTRF_RCP(u22);
        const double u32 = fma(-l31, u21, fma(-l30, u20, d3b.x));
        const double l32 = u32 * r2;
        const double u33 = fma(-l32, u32, fma(-l31, u31, fma(-l30, u30, d3b.y)));
        const double r3 = SRK_POTRF_RCP(u33);
        dmin = fmin(fmin(dmin, u00), fmin(u11, fmin(u22, u33)));
        dchk = fma(0.0, u00, fma(0.0, u11, fma(0.0, u22, fma(0.0, u33, dchk))));
#endif
        // ---- B: this row against the factor.  y_k = p_k - sum_{k' < k} (y_k' / d_k') u_kk'
        const double p0 = quad_bcast<0>(a[g]), p1 = quad_bcast<1>(a[g]), p2 = quad_bcast<2>(a[g]), p3 = quad_bcast<3>(a[g]);
#ifndef SRK_POTRF_SEQ_PIVOTS
        // (with the factor's multipliers l_jk = u_jk / d_k -- the same for every row -- a row's chain is three dependent
        // multiply-adds instead of seven operations; its own multipliers z_k = y_k / d_k follow in parallel)
        const double m10 = u10 * r0, m32 = u32 * r2;
        const double y0 = p0;
        const double y1 = fma(-y0, m10, p1);
        const double y2 = fma(-y1, l21, fma(-y0, l20, p2));
        const double y3 = fma(-y2, m32, fma(-y1, l31, fma(-y0, l30, p3)));
        const double z0 = y0 * r0, z1 = y1 * r1, z2 = y2 * r2, z3 = y3 * r3;
#else
        const double m10 = u10 * r0, m32 = u32 * r2;
        const double y0 = p0, z0 = y0 * r0;
        const double y1 = fma(-z0, u10, p1), z1 = y1 * r1;
        const double y2 = fma(-z1, u21, fma(-z0, u20, p2)), z2 = y2 * r2;
        const double y3 = fma(-z2, u32, fma(-z1, u31, fma(-z0, u30, p3))), z3 = y3 * r3;
#endif
        {
            const double y01 = (q & 1) ? y1 : y0, y23 = (q & 1) ? y3 : y2;
            a[g] = (q & 2) ? y23 : y01; // final (unscaled) entry of column 4g + q
        }
        double fv = 0; // PUB: lane l < 10 of every wave publishes value l of the factor (no branch inside the round, no two
        // lanes on one word; the waves store the same numbers) -- behind the group's first barrier, where the chain waits
        // for LDS anyway (the last reciprocal is not needed before it)
        if (PUB) {
            const int l = t & 63;
            fv = m10; // the factor's multipliers l_jk = u_jk / d_k (what a row's elimination multiplies by)
            fv = l == 1 ? l20 : fv, fv = l == 2 ? l30 : fv, fv = l == 3 ? l21 : fv, fv = l == 4 ? l31 : fv, fv = l == 5 ? m32 : fv;
            fv = l == 6 ? r0 : fv, fv = l == 7 ? r1 : fv, fv = l == 8 ? r2 : fv, fv = l == 9 ? r3 : fv;
            if (g == 15) sF[64 * (g & 1) + l] = fv;
        }
        if (g == 15) break;
        sY[i][q] = a[g];
        lds_barrier();
        if (PUB) sF[64 * (g & 1) + (t & 63)] = fv;
        // the next group's slot: update, publish, barrier -- the remaining slots follow behind that barrier (above)
        {
            const double2* yp = reinterpret_cast<const double2*>(&sY[4 * (g + 1) + q][0]);
            const double2 y01 = yp[0], y23 = yp[1];
            a[g + 1] = fma(-z3, y23.y, fma(-z2, y23.x, fma(-z1, y01.y, fma(-z0, y01.x, a[g + 1]))));
            sP[i][q] = a[g + 1];
        }
        zp0 = z0; zp1 = z1; zp2 = z2; zp3 = z3;
        lds_barrier(); // panel of group g + 1 is complete
    }
    // the owner of a diagonal entry holds its pivot d_i
#pragma unroll
    for (int m = 0; m < 16; ++m)
        if (4 * m + q == i) sDiag[i] = a[m];
    lds_barrier();
    if (t < NB) {
        sInv[t] = fast_rsqrt(sDiag[t]); // 1 / L_tt
        if (PUB) sRsq[t] = sInv[t];
    }
    lds_barrier();
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        int c = 4 * m + q;
        double v = a[m] * sInv[c];          // c < i : a_ic / sqrt(d_c) ; c == i : d_i / sqrt(d_i) = L_ii
        v = (c <= i) ? v : 0.0;
        sD[i][c] = v;
    }
    lds_barrier();
    // what a sweep multiplies by is 1 / L_tt formed from the STORED L_tt: a workgroup that receives the factored tile from
    // another one (k_step256) computes exactly this, so the fused and the unfused launch sequences agree bit for bit
    if (t < NB) sInv[t] = fast_rcp(sD[t][t]);
    lds_barrier();
    return !(dmin > 0.0) || (dchk != 0.0);
}

// ---------------------------------------------------------------- batched launches
// Every factorisation kernel works on a BATCH of independent matrices (blockIdx.z = item): the chunks of a chunked
// solve advance in lock step through the same launches, so the launch count is that of ONE chunk and no extra streams
// are needed.  A plain solve is a batch of one.  The descriptors travel by value in the kernel arguments.
struct CholItem {
    double* A;   // matrix (ld x ld, row-major, lower part authoritative)
    double* w;   // running right-hand side
    double* y;   // forward solution
    double* x;   // solution
    double* dinv; // inverses of the diagonal 64-tiles
    int64_t ld, ncols, r2_begin, r2_end; // eliminate columns [0, ncols); border rows [r2_begin, r2_end) ride along
};
struct CholBatch { CholItem it[SRK_MAX_CHUNKS]; };
struct CholStep { int64_t v[SRK_MAX_CHUNKS]; }; // one per-launch value per item; < 0 = item takes no part
struct CholSub { unsigned char v[SRK_MAX_CHUNKS]; }; // k_step256: an item's sub-steps on real columns (1 .. 4)
// host side of an item: its skylines (may be NULL) and which of its border rows are structurally zero when.  The border
// rows [r2_begin, r2_end) of a chunk are [separator above | separator below], r2_split between them.  A chunk without a
// separator above (the first one) never needs the first part; the rows of the separator below stay zero until the
// elimination reaches column bot_first_col (they couple with the chunk's last bandwidth of columns only), and for good
// in the last chunk.  Sweeping and updating all-zero rows is exact but wasted: at the first step of a 512-column chunk
// a third of the rows and more than half of the trailing update.
struct CholHostItem {
    const int64_t* row_end;
    const int64_t* col_begin;
    int64_t r2_split = 0, bot_first_col = 0;
    bool has_top = true, has_bot = true;
    // columns from n_real on are padding: an identity diagonal, zeros elsewhere, which no update changes (a multiple of 64)
    int64_t n_real = INT64_MAX;
};

// ---------------------------------------------------------------- inverse of a factored 64x64 diagonal tile
// Z = L^-1 (lower triangular) by blocked inversion: the four 16x16 diagonal blocks by register-resident forward
// substitution (16 dependent steps), then two levels of
//     [ L11  0  ]^-1   [ Z11            0  ]
//     [ L21 L22 ]    = [ -Z22 L21 Z11  Z22 ]
// as small LDS matrix products over all 256 threads.  It turns the 64 dependent steps of each backward tile solve into
// one matrix-vector product (k_bwd256).  sL: the factor (zeros above the diagonal), row stride NB + 2; out: row-major
// 64 x 64 in global memory.  Runs in the extra workgroup of k_panel, beside the row sweeps of the other workgroups.
__device__ __forceinline__ void tile_inverse(const double (*sL)[NB + 2], double (*sZ)[NB + 1], double (*sT)[33],
                                             double* sR, double* __restrict__ out)
{
    const int t = threadIdx.x;
    for (int e = t; e < NB * NB; e += 256) sZ[e >> 6][e & 63] = 0.0;
    // reciprocals of the diagonal once, in parallel: an IEEE divide inside each of the 16 dependent substitution steps
    // below would cost more than the rest
    if (t < NB) sR[t] = 1.0 / sL[t][t];
    __syncthreads();
    if (t < 64) { // thread (b, c): column c of the inverse of diagonal block b, right-looking in registers
        const int o = 16 * (t >> 4), c = t & 15;
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double z = (i >= c) ? acc[i] * sR[o + i] : 0.0;
            sZ[o + i][o + c] = z;
#pragma unroll
            for (int k = i + 1; k < 16; ++k) acc[k] = fma(-sL[o + k][o + i], z, acc[k]);
        }
    }
    __syncthreads();
    // level 32: pairs (block 1 | block 0) and (block 3 | block 2); thread -> outputs idx = t, t + 256
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int idx = t + 256 * h, pr = idx >> 8, i = (idx >> 4) & 15, j = idx & 15;
        const int o1 = 32 * pr, o2 = o1 + 16;
        double sum = 0;
#pragma unroll
        for (int m = 0; m < 16; ++m) sum = fma(sL[o2 + i][o1 + m], sZ[o1 + m][o1 + j], sum);
        sT[16 * pr + i][j] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int idx = t + 256 * h, pr = idx >> 8, i = (idx >> 4) & 15, j = idx & 15;
        const int o1 = 32 * pr, o2 = o1 + 16;
        double sum = 0;
#pragma unroll
        for (int m = 0; m < 16; ++m) sum = fma(sZ[o2 + i][o2 + m], sT[16 * pr + m][j], sum);
        sZ[o2 + i][o1 + j] = -sum;
    }
    __syncthreads();
    // level 64: T = L21 Z11 (32x32), X = -Z22 T; thread -> outputs (i, j), (i + 8, j), (i + 16, j), (i + 24, j)
    {
        const int i0 = t >> 5, j = t & 31;
        double sum[4] = { 0, 0, 0, 0 };
#pragma unroll 8
        for (int m = 0; m < 32; ++m) {
            const double zv = sZ[m][j];
#pragma unroll
            for (int h = 0; h < 4; ++h) sum[h] = fma(sL[32 + i0 + 8 * h][m], zv, sum[h]);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) sT[i0 + 8 * h][j] = sum[h];
    }
    __syncthreads();
    {
        const int i0 = t >> 5, j = t & 31;
        double sum[4] = { 0, 0, 0, 0 };
#pragma unroll 8
        for (int m = 0; m < 32; ++m) {
            const double tv = sT[m][j];
#pragma unroll
            for (int h = 0; h < 4; ++h) sum[h] = fma(sZ[32 + i0 + 8 * h][32 + m], tv, sum[h]);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) sZ[32 + i0 + 8 * h][j] = -sum[h];
    }
    __syncthreads();
    for (int e = t; e < NB * NB; e += 256) out[e] = sZ[e >> 6][e & 63];
}

// ---------------------------------------------------------------- inner panel: potrf + trsm + forward substitution
// d = index of the 64-wide diagonal tile.  Rows (d+1)*64 .. row_end-1 of columns [64 d, 64 d + 64) become L.
// w is the running right-hand side: y_d = L_dd^-1 w_d is published to y, and w_r -= L[r, d-cols] . y_d for the
// rows below (so the forward substitution L y = b costs no extra launches).
// Workgroup 0 of every item is the INVERSE workgroup: it factors the tile like the others, then forms L_dd^-1 for the
// backward substitution (k_bwd256) while the other workgroups sweep their rows -- about as long as a sweep, so the
// five k_dinv launches a solve used to need (12 us each) cost nothing.  Workgroups 1.. take PANEL_ROWS rows each.
#ifdef SRK_PANEL_STAMPS
__device__ long long g_panel_stamps[16];
#define STAMP(k) do { if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && d == 40) g_panel_stamps[k] = wall_clock64(); } while (0)
#define STAMPI(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && d == 40) g_panel_stamps[k] = wall_clock64(); } while (0) // inverse workgroup
#else
#define STAMP(k)
#define STAMPI(k)
#endif
#define PANEL_ROWS 63 // matrix rows per workgroup; the 64th quad carries the right-hand side as one more row
__global__ __launch_bounds__(256) void k_panel(const CholBatch B, const CholStep rend, const CholStep r2b, const CholStep r2e,
                                                int64_t d, int* __restrict__ info)
{
    const int64_t row_end = rend.v[blockIdx.z];
    if (row_end < 0) return;
    double* __restrict__ A = B.it[blockIdx.z].A;
    double* __restrict__ w = B.it[blockIdx.z].w;
    double* __restrict__ y = B.it[blockIdx.z].y;
    const int64_t ld = B.it[blockIdx.z].ld, r2_begin = r2b.v[blockIdx.z], r2_end = r2e.v[blockIdx.z]; // live border rows of this step
    const bool inv_wg = blockIdx.x == 0;
    const int64_t rblk = (int64_t)blockIdx.x - 1; // row block of a sweeping workgroup
    int64_t blocks;
    {
        int64_t rows = row_end - (d + 1) * NB;
        if (rows < 0) rows = 0;
        rows += r2_end - r2_begin;
        blocks = (rows + PANEL_ROWS - 1) / PANEL_ROWS;
        if (!inv_wg && rblk >= blocks) return; // the grid is sized for the largest item of the batch
    }
    __shared__ __attribute__((aligned(16))) double sD[NB][NB + 2];
    __shared__ double sZ[NB][NB + 1]; // inverse workgroup only
    __shared__ double sT[32][33];
    __shared__ __attribute__((aligned(16))) double sCol[12 * NB]; // potrf64: panel [64][4] + finals 2 x [64][4]
    __shared__ double sDiag[NB];
    __shared__ double sInv[NB];
    __shared__ double sy[NB];
    const int64_t k0 = d * NB;
    double* Ab = A + k0 * ld + k0;
    STAMP(0);
    STAMPI(10);
    {
        // row i = t >> 2, 16 columns from (t & 3) * 16: eight independent 16-byte loads per thread; entries above the
        // diagonal are masked by select (never branched on)
        const int i = threadIdx.x >> 2, cb = (threadIdx.x & 3) * 16;
        const double2* src = reinterpret_cast<const double2*>(Ab + (int64_t)i * ld + cb);
        double2 v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = src[t];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            int c = cb + 2 * t;
            sD[i][c] = (c <= i) ? v[t].x : 0.0;
            sD[i][c + 1] = (c + 1 <= i) ? v[t].y : 0.0;
        }
    }
    __syncthreads();
    STAMP(1);
    // Row sweep X = A[rows, panel] L_dd^-T, four columns at a time.  A row is split over the 4 lanes of a quad (lane q
    // owns columns c = 4m + q).  Quad 63 carries the right-hand side w_d as one more row: its sweep IS the forward
    // substitution y_d = L_dd^-1 w_d, at no extra latency.
    const int q = threadIdx.x & 3, qd = threadIdx.x >> 2;
    const bool is_rhs = qd == PANEL_ROWS;
    // rows below the diagonal tile inside the skyline [.., row_end) first, then the border rows [r2_begin, r2_end)
    // (the separator rows of a chunked factorisation; empty otherwise)
    int64_t rows1 = row_end - (k0 + NB);
    if (rows1 < 0) rows1 = 0;
    // (an item without rows below this tile has no sweeping workgroup: the inverse workgroup then runs the sweep for the
    // right-hand side quad alone, with every matrix row dead, before it inverts the tile)
    const int64_t ridx = (inv_wg ? 0 : rblk) * PANEL_ROWS + qd;
    const int64_t r = ridx < rows1 ? k0 + NB + ridx : r2_begin + (ridx - rows1);
    const bool live = !is_rhs && !inv_wg && ridx < rows1 + (r2_end - r2_begin);
    double* row = is_rhs ? (w + k0) : (A + (live ? r : k0) * ld + k0); // dead rows read the diagonal tile (harmless)
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = row[4 * m + q];
    // (issued before the tile is factored: the loads' latency, ~1.5 us after the factorisation, hides under it)
    bool bad = potrf64<false>(sD, sCol, sDiag, sInv, nullptr, nullptr);
    STAMP(2);
    if (bad && inv_wg && threadIdx.x == 0) atomicOr(info, 1);
    // The factored tile is never stored back into A: the other workgroups of this launch read the unfactored tile from
    // A when they start, and nothing orders their start before such a store -- a grid larger than the chip, or a GPU
    // shared with other processes, starts some of them late.  No later kernel reads a diagonal tile of A or L_dd itself:
    // the backward substitution works with the inverse, which goes to this tile's slot of the Dinv buffer.
    if (inv_wg && blocks > 0) {
        STAMPI(11);
        tile_inverse(sD, sZ, sT, sDiag, B.it[blockIdx.z].dinv + d * NB * NB);
        STAMPI(12);
        return;
    }
    STAMP(3);
    STAMP(4);
#pragma unroll
    for (int b = 0; b < 16; ++b) {
#ifdef SRK_PANEL_STAMPS
        if (b == 4) STAMP(7);
        if (b == 8) STAMP(8);
        if (b == 12) STAMP(9);
#endif
        const int c0 = 4 * b;
        // the four entries of this row in columns c0..c0+3, on every lane of the quad
        double v0 = quad_bcast<0>(a[b]), v1 = quad_bcast<1>(a[b]), v2 = quad_bcast<2>(a[b]), v3 = quad_bcast<3>(a[b]);
        // 4x4 lower-triangular solve against L_dd[c0.., c0..]
        double x0 = v0 * sInv[c0];
        double x1 = fma(-x0, sD[c0 + 1][c0], v1) * sInv[c0 + 1];
        double x2 = fma(-x1, sD[c0 + 2][c0 + 1], fma(-x0, sD[c0 + 2][c0], v2)) * sInv[c0 + 2];
        double x3 = fma(-x2, sD[c0 + 3][c0 + 2], fma(-x1, sD[c0 + 3][c0 + 1], fma(-x0, sD[c0 + 3][c0], v3))) * sInv[c0 + 3];
        {
            const double x01 = (q & 1) ? x1 : x0, x23 = (q & 1) ? x3 : x2;
            a[b] = (q & 2) ? x23 : x01;
        }
        // rank-4 update of the columns to the right: a[c] -= sum_k x_k L[c][c0 + k].  The LDS addresses are static, so
        // without fences the scheduler hoists every read to the top and spills (seen: 256 VGPR + 256 AGPR + scratch);
        // a fence every four columns keeps <= 8 sixteen-byte reads in flight.
#pragma unroll
        for (int m = b + 1; m < 16; ++m) {
            const double2* lp = reinterpret_cast<const double2*>(&sD[4 * m + q][c0]);
            double2 l01 = lp[0], l23 = lp[1];
            a[m] = fma(-x3, l23.y, fma(-x2, l23.x, fma(-x1, l01.y, fma(-x0, l01.x, a[m]))));
            if (((m - b) & 3) == 0) asm volatile("" ::: "memory");
        }
        asm volatile("" ::: "memory");
    }
    STAMP(5);
    if (is_rhs) {
#pragma unroll
        for (int m = 0; m < 16; ++m) sy[4 * m + q] = a[m];
    }
    __syncthreads();
    if (is_rhs && (inv_wg || rblk == 0)) {
#pragma unroll
        for (int m = 0; m < 16; ++m) y[k0 + 4 * m + q] = a[m];
    }
    if (inv_wg) { // only reached when the item has no rows below the tile
        __syncthreads();
        tile_inverse(sD, sZ, sT, sDiag, B.it[blockIdx.z].dinv + d * NB * NB);
        return;
    }
    if (is_rhs) return;
    // w_r -= L[r, panel] . y_d  (the forward substitution's update of the rows below)
    double dot = 0;
#pragma unroll
    for (int m = 0; m < 16; ++m) dot = fma(a[m], sy[4 * m + q], dot);
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    if (live) {
#pragma unroll
        for (int m = 0; m < 16; ++m) row[4 * m + q] = a[m];
        if (q == 0) w[r] -= dot;
    }
    STAMP(6);
}
#ifdef SRK_PANEL_STAMPS
extern "C" void srk_dbg_panel_stamps(long long* out) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_panel_stamps), sizeof(long long) * 16); }
#endif

// ---------------------------------------------------------------- 64-deep update inside the outer panel (MFMA)
// A[rt, ct] -= L[rt, d] L[ct, d]^T for row tiles rt > d (rows < row_end) and column tiles d < ct <= c_hi, ct <= rt.
// grid = (row tiles, column tiles).  4 waves, each a 32x32 quadrant = 2x2 accumulator tiles, K = 64.
__global__ __launch_bounds__(256) void k_upd64(const CholBatch B, const CholStep rend, const CholStep r2b, const CholStep r2e,
                                                int64_t d, int64_t c_hi)
{
    __shared__ double sA[NB][LDSP];
    __shared__ double sB[NB][LDSP];
    if (rend.v[blockIdx.z] < 0) return;
    double* __restrict__ A = B.it[blockIdx.z].A;
    const int64_t ld = B.it[blockIdx.z].ld, r2_begin = r2b.v[blockIdx.z];
    int64_t tiles1 = (rend.v[blockIdx.z] - (d + 1) * NB) / NB;
    if (tiles1 < 0) tiles1 = 0;
    if ((int64_t)blockIdx.x >= tiles1 + (r2e.v[blockIdx.z] - r2_begin) / NB) return;
    // row tiles: `tiles1` tiles right below the diagonal tile, then the border tiles starting at r2_begin
    const bool in1 = (int64_t)blockIdx.x < tiles1;
    int64_t rt = d + 1 + blockIdx.x;
    int64_t ct = d + 1 + blockIdx.y;
    if (ct > c_hi || (in1 && ct > rt)) return;
    int64_t k0 = d * NB, c0 = ct * NB;
    int64_t r0 = in1 ? rt * NB : r2_begin + ((int64_t)blockIdx.x - tiles1) * NB;
    {
        int row = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;
        const double* pa = A + (r0 + row) * ld + k0 + seg;
        const double* pb = A + (c0 + row) * ld + k0 + seg;
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            double2 va = *reinterpret_cast<const double2*>(pa + t);
            double2 vb = *reinterpret_cast<const double2*>(pb + t);
            sA[row][seg + t] = va.x;
            sA[row][seg + t + 1] = va.y;
            sB[row][seg + t] = vb.x;
            sB[row][seg + t + 1] = vb.y;
        }
    }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int wr = wave >> 1, wc = wave & 1;
    int lr = lane & 15, lk = lane >> 4;
    // the C tile is read-modify-write: fetch it now, its latency hides under the barrier and the MFMAs
    // (f64 16x16x4 accumulator map: col = lane & 15, row = (lane >> 4) + 4 * reg)
    double* pc0 = A + (r0 + wr * 32 + lk) * ld + c0 + wc * 32 + lr;
    double cv[2][2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) cv[m][n][reg] = pc0[(int64_t)(m * 16 + 4 * reg) * ld + n * 16];
    __syncthreads();
    double4_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){ 0, 0, 0, 0 };
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        double a0 = sA[wr * 32 + lr][kk * 4 + lk];
        double a1 = sA[wr * 32 + 16 + lr][kk * 4 + lk];
        double b0 = sB[wc * 32 + lr][kk * 4 + lk];
        double b1 = sB[wc * 32 + 16 + lr][kk * 4 + lk];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) pc0[(int64_t)(m * 16 + 4 * reg) * ld + n * 16] = cv[m][n][reg] - acc[m][n][reg];
}

// ---------------------------------------------------------------- fused outer step: the four inner panels in ONE launch
// The k_panel / k_upd64 sequence costs an outer step 4 x 20 + 3 x 8.5 us although its critical path -- factor tile d,
// sweep the 64 rows of tile d + 1, update tile (d + 1, d + 1), factor it -- is far shorter: every panel launch loads,
// factors the diagonal tile redundantly in every workgroup, sweeps ALL rows and stores before the update launch may
// start.  k_step256 runs the whole outer step as one launch of 512-thread workgroups with roles (blockIdx.x), ordered so
// that a workgroup only ever waits for the workgroup of LOWER index of its own item (blockIdx.z):
//   x = 0   DIAGONAL-BLOCK workgroup (round 4): ONE workgroup owns the whole 256 x 256 diagonal block for all four
//           sub-steps, so the factorisation chain crosses no global-memory hand-off (rounds 2 / 3: tile row t lived in
//           workgroup t and every sub-step paid flag 0.9 + tile load 2.2 + publish X 1.4 + publish L 1.3 us on the chain).
//           Sub-step d:  P(d)  waves 0-3 factor tile (d, d) (potrf64, in LDS + registers) while waves 4-7 carry the rows of
//                              the tiles (t, d), t > d, below it as PASSENGER rows: they eliminate their rows against each
//                              group's 4 x 4 factor (published through LDS) one group behind the chain, so X_td = A_td
//                              L_dd^-T is complete when L_dd is -- the 4 us sweep of the next tile row is off the chain;
//                        pub   waves 0-3 issue the write-through stores of L_dd and the X_td and set F(d) when drained, beside
//                        U(d)  the rank-64 updates C -= X_td X_(d+1)d^T of the NEXT PANEL's tiles (t, d + 1), as fp64 MFMA
//                              sub-tiles dealt to the eight waves by a static table (srk_step_tables.inc); the results are
//                              held in registers until every wave is done with the X tiles in LDS, then go there.  (One CU's
//                              fp64 pipe takes 0.17 us a sub-tile: with ALL tiles of the block here the updates cost what
//                              the hand-offs had -- the tiles behind the next panel are the helper workgroups'.)
//   x = 1   inverse workgroup: L_dd^-1 of the four tiles as they appear (for the backward substitution).
//   x = 2   forward-substitution workgroup: y_d = L_dd^-1 w_d as the tiles appear (flag Y(d)), w_t -= X_td y_d for the block's
//           rows below.  (Inside the diagonal-block workgroup the 64 dependent steps took 4 - 8 us of a wave whose SIMD was
//           streaming MFMAs.)
//   x = 3   helper workgroup: the tiles (2, 2), (3, 2), (3, 3) of the diagonal block -- the ones behind the next panel -- take
//           their rank-64 updates of the sub-steps d <= c - 2 here, from the published X tiles, and go back with flags G: the ONLY
//           wait of the diagonal-block workgroup, for index 3 of its own item, with a whole factorisation of slack.
//   x >= 4  row workgroups: 64 rows each of the rows below the diagonal block (skyline rows, then the live border rows);
//           per sub-step: wait for F(d), sweep against L_dd, update w with y_(d-1), update the tiles (., d + 1 .. 3) with the
//           published X_td.  One-way consumers: nothing waits for them inside the launch.
// Hand-offs follow the guide's recipe (cdna_hip_programming.md, Guideline 16 R1): payload stored write-through (sc1
// stores), every storing wave drains and counts itself in (LDS counter), the LAST one stores the flag (sc1); the consumer
// polls that word relaxed from one lane, ONE agent-scope acquire, barrier, then plain loads.  A flag word holds the EPOCH of
// the launch that set it (the host counts launches per stream), so nothing is ever reset.  Every spin is bounded: a timeout
// sets bit 8 of *info and the workgroup carries on (wrong numbers, flagged; the host repeats the solve with the unfused
// kernels).  Deadlock: dispatch is in order of the linear workgroup index per XCD; every workgroup but index 0 waits only
// for index 0 of its item, index 0 only for index 3, which waits only for index 0: whenever the lowest unfinished
// workgroup of an item is resident, the ones it can wait for are resident or first in line; the host keeps a launch within
// 256 workgroups.
// Arithmetic: potrf64 as in k_panel; the block's own rows are eliminated in the unscaled (L D L^T) form of potrf64 instead of
// k_panel's substitution against the stored factor, so the fused and the unfused sequences agree to rounding, not bit for bit.
#include "srk_step_tables.inc"
#define ST_F(d) (d)                                 // flag words of an item: L_dd and the X_td published
#define ST_Y(d) (4 + (d))                           // y_d published
#define ST_G(k) (8 + (k))                           // helper k is done with its tile: (2, 2), (3, 2), (3, 3)
#define ST_WORDS 16
#define ST_SPIN_MAX (1 << 17)
#ifndef ST_POLL_SLEEP
#define ST_POLL_SLEEP 2
#endif
#define STP_THREADS 512
#define STP_TILE (NB * LDSP)                        // a 64 x 64 tile in LDS, row stride NB + 2
// LDS map of the kernel (doubles)
#define STP_COL (4 * STP_TILE)                      // potrf64 exchange: panel + two buffers of finals [12 NB]
#define STP_F (STP_COL + 12 * NB)                   // 4 x 4 factors of two groups [2][64] (ten words used)
#define STP_DIAG (STP_F + 2 * NB)
#define STP_INV (STP_DIAG + NB)
#define STP_RSQ (STP_INV + NB)
#define STP_Y (STP_RSQ + NB)
#define STP_W (STP_Y + NB)                          // (inverse workgroup) w_d
#define STP_CNT (STP_W + NB)                        // publication counters (unsigned [8]) + fault word
#define STP_LDS_DOUBLES (STP_CNT + 8)
static_assert(STP_LDS_DOUBLES * sizeof(double) <= 160 * 1024, "one workgroup per CU");
typedef unsigned int __attribute__((address_space(1))) gu32_t;
__device__ __forceinline__ void st_store(double* p, double v) // write-through store (global_store_dwordx2 sc1)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef double dbl2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_store16(double* p, dbl2_t v) // 16-byte write-through store
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ double readlane_f64(double v, int l) // l: wave-uniform
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
// wait until every flag word of `mask` holds `epoch`; one lane polls, one acquire, then the workgroup's barrier
__device__ __forceinline__ void st_wait(unsigned* fl, unsigned mask, unsigned epoch, int* info)
{
    if (threadIdx.x == 0) {
        unsigned pending = mask;
        int spins = 0;
        while (pending) {
            const int k = __ffs(pending) - 1;
            if (__hip_atomic_load(fl + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) {
                pending &= pending - 1;
                continue;
            }
            // (bounded; and once any workgroup has given up, nobody waits out its own bound)
            if (++spins > ST_SPIN_MAX || ((spins & 63) == 0 && (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 8))) {
                atomicOr(info, 8);
                break;
            }
            __builtin_amdgcn_s_sleep(ST_POLL_SLEEP);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
// the same for ONE wave (no barrier): lane 0 polls, the wave acquires
__device__ __forceinline__ void st_wait_wave(unsigned* fl, unsigned mask, unsigned epoch, int* info)
{
    if ((threadIdx.x & 63) == 0) {
        unsigned pending = mask;
        int spins = 0;
        while (pending) {
            const int k = __ffs(pending) - 1;
            if (__hip_atomic_load(fl + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) {
                pending &= pending - 1;
                continue;
            }
            if (++spins > ST_SPIN_MAX || ((spins & 63) == 0 && (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 8))) {
                atomicOr(info, 8);
                break;
            }
            __builtin_amdgcn_s_sleep(ST_POLL_SLEEP);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// 64 x 64 tile at G (row stride ld) -> sD, by the first 256 threads; LOWER: zeros above the diagonal
template <bool LOWER> __device__ __forceinline__ void st_load_tile(double (*sD)[LDSP], const double* __restrict__ G, int64_t ld)
{
    const int i = threadIdx.x >> 2, cb = (threadIdx.x & 3) * 16;
    const double2* src = reinterpret_cast<const double2*>(G + (int64_t)i * ld + cb);
    double2 v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = src[t];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int c = cb + 2 * t;
        sD[i][c] = (!LOWER || c <= i) ? v[t].x : 0.0;
        sD[i][c + 1] = (!LOWER || c + 1 <= i) ? v[t].y : 0.0;
    }
}
// the same by all 512 threads (no masking): thread -> row tid >> 3, eight doubles from column 8 (tid & 7)
__device__ __forceinline__ void st_load_tile512(double (*sD)[LDSP], const double* __restrict__ G, int64_t ld)
{
    const int i = threadIdx.x >> 3, cb = (threadIdx.x & 7) * 8;
    const double2* src = reinterpret_cast<const double2*>(G + (int64_t)i * ld + cb);
    double2 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = src[t];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        sD[i][cb + 2 * t] = v[t].x;
        sD[i][cb + 2 * t + 1] = v[t].y;
    }
}
// test hook (srk_dbg_step_fault): the next N launches' diagonal-block workgroup of item 0 stops after its first tile and
// publishes nothing -- every consumer of that item runs into its spin bound, bit 8 of *info is set and the host repeats
// the solve unfused
// -- in the development build only (make dev: libsrk_ba_dev.so, -DSRK_DEV; tests/_dev_worker.py loads it in a subprocess)
#ifdef SRK_DEV
__device__ int g_step_fault = 0;
extern "C" void srk_dbg_step_fault(int launches) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_step_fault), &launches, sizeof(int)); }
#endif
#ifdef SRK_STEP_STAMPS // development (tools/step_stamps.sh): wall-clock stamps of item 0's workgroups of one launch
__device__ long long g_step_stamps[8][32];
#define SST(k) do { if (blockIdx.z == 0 && K == 0 && (threadIdx.x & 255) == 0 && blockIdx.x < 4) g_step_stamps[2 * blockIdx.x + (threadIdx.x >> 8)][k] = wall_clock64(); } while (0)
extern "C" void srk_dbg_step_stamps(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_stamps), sizeof(long long) * 256); }
#else
#define SST(k)
#endif

// ---- passenger rows of potrf64 (waves 4-7 of the diagonal-block workgroup)
// One group of four columns on NP passenger rows per thread: eliminate the rows' entries of group g against the group's
// 4 x 4 factor (u_jk, 1 / d_k as the chain published them), keep the finals, apply the rank-4 term to the slots behind.
template <int NP> __device__ __forceinline__ void passenger_group(double (&a)[3][16], const int g, const double* sPY, const double* sF, const int q)
{
    const double2* F = reinterpret_cast<const double2*>(sF + 64 * (g & 1));
    const double2* sY = reinterpret_cast<const double2*>(sPY + 4 * NB * (1 + (g & 1))) + 2 * q; // [row][2]: finals of this lane's column rows
    const double2 f0 = F[0], f1 = F[1], f2 = F[2], f3 = F[3], f4 = F[4];
    // the finals of the first eight slots behind the group are requested before the elimination (they do not depend on it): a
    // fenced read-then-update loop left the wave four exposed LDS round trips a round, as long as the chain's own round
    constexpr int H = 8;
    double2 ya[H], yb[H];
#pragma unroll
    for (int k = 0; k < H; ++k)
        if (g + 1 + k < 16) ya[k] = sY[2 * 4 * (g + 1 + k)], yb[k] = sY[2 * 4 * (g + 1 + k) + 1];
    const double l10 = f0.x, l20 = f0.y, l30 = f1.x, l21 = f1.y, l31 = f2.x, l32 = f2.y, r0 = f3.x, r1 = f3.y, r2 = f4.x, r3 = f4.y;
    double z[NP > 0 ? NP : 1][4];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const double p0 = quad_bcast<0>(a[p][g]), p1 = quad_bcast<1>(a[p][g]), p2 = quad_bcast<2>(a[p][g]), p3 = quad_bcast<3>(a[p][g]);
        const double y0 = p0;
        const double y1 = fma(-y0, l10, p1);
        const double y2 = fma(-y1, l21, fma(-y0, l20, p2));
        const double y3 = fma(-y2, l32, fma(-y1, l31, fma(-y0, l30, p3)));
        const double z0 = y0 * r0, z1 = y1 * r1, z2 = y2 * r2, z3 = y3 * r3;
        const double y01 = (q & 1) ? y1 : y0, y23 = (q & 1) ? y3 : y2;
        a[p][g] = (q & 2) ? y23 : y01; // final (unscaled) entry of column 4g + q
        z[p][0] = z0, z[p][1] = z1, z[p][2] = z2, z[p][3] = z3;
    }
    asm volatile("" ::: "memory");
    double2 yc[H], yd[H]; // the slots behind those: in flight while the first eight are updated
#pragma unroll
    for (int k = 0; k < H; ++k)
        if (g + 1 + H + k < 16) yc[k] = sY[2 * 4 * (g + 1 + H + k)], yd[k] = sY[2 * 4 * (g + 1 + H + k) + 1];
#pragma unroll
    for (int k = 0; k < H; ++k)
        if (g + 1 + k < 16) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
                a[p][g + 1 + k] = fma(-z[p][3], yb[k].y, fma(-z[p][2], yb[k].x, fma(-z[p][1], ya[k].y, fma(-z[p][0], ya[k].x, a[p][g + 1 + k]))));
        }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < H; ++k)
        if (g + 1 + H + k < 16) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
                a[p][g + 1 + H + k] = fma(-z[p][3], yd[k].y, fma(-z[p][2], yd[k].x, fma(-z[p][1], yc[k].y, fma(-z[p][0], yc[k].x, a[p][g + 1 + H + k]))));
        }
}
// The barrier sequence of potrf64<true> with the passenger rows one group behind the chain: group g's factor and finals are
// in LDS after the chain's first barrier of round g; they are consumed here during round g + 1 (the finals are
// double-buffered, the chain overwrites a buffer two rounds later).  On exit a[p][m] = X[row][4m + q] (scaled).
template <int NP> __device__ __forceinline__ void potrf64_passengers(double (&a)[3][16], const double* sPY, const double* sF, const double* sRsq, const int q)
{
    lds_barrier(); // the panel of group 0 is published
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g > 0 && NP > 0) passenger_group<NP>(a, g - 1, sPY, sF, q);
        if (g == 15) break;
        lds_barrier(); // finals of group g
        lds_barrier(); // panel of group g + 1
    }
    lds_barrier(); // (pivots published) -- the factor of group 15 is there
    if (NP > 0) passenger_group<NP>(a, 15, sPY, sF, q);
    lds_barrier(); // 1 / sqrt(d) is there
    if (NP > 0) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const double s = sRsq[4 * m + q];
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p][m] *= s;
        }
    }
}

// ---- rank-64 update sub-tiles of the diagonal-block workgroup
// acc[s] = X_t[16 r .., :] X_c[16 (s0 + s) .., :]^T, s < NS: pa -> T_t[16 r + lr][lk], pb -> T_c[16 s0 + lr][lk]; the operands
// of the next four K steps are in flight while four are multiplied (one step ahead left the pipe waiting for LDS: 60 %).  (f64 16x16x4: a = A[lane & 15][lane >> 4], b = B[lane & 15][lane >> 4],
// accumulator register reg = C[(lane >> 4) + 4 reg][lane & 15].)
template <int NS> __device__ __forceinline__ void mfma_strip(const double* pa, const double* pb, double4_t (&acc)[4])
{
    constexpr int PD = 4; // K steps per operand chunk; chunk c + 1 is in flight while chunk c is multiplied
    double av[2][PD], bv[2][NS][PD];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = (double4_t){ 0, 0, 0, 0 };
#pragma unroll
    for (int j = 0; j < PD; ++j) {
        av[0][j] = pa[4 * j];
#pragma unroll
        for (int s = 0; s < NS; ++s) bv[0][s][j] = pb[s * 16 * LDSP + 4 * j];
    }
#pragma unroll
    for (int ch = 0; ch < NB / 4 / PD; ++ch) {
        const int cur = ch & 1, nxt = cur ^ 1;
        if (ch + 1 < NB / 4 / PD) {
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                av[nxt][j] = pa[4 * (PD * (ch + 1) + j)];
#pragma unroll
                for (int s = 0; s < NS; ++s) bv[nxt][s][j] = pb[s * 16 * LDSP + 4 * (PD * (ch + 1) + j)];
            }
        }
#pragma unroll
        for (int j = 0; j < PD; ++j)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[cur][j], bv[cur][s][j], acc[s], 0, 0, 0);
    }
}
__device__ __forceinline__ void mfma_strip_n(int ns, const double* pa, const double* pb, double4_t (&acc)[4])
{
    switch (ns) { // wave-uniform
    case 1: mfma_strip<1>(pa, pb, acc); break;
    case 2: mfma_strip<2>(pa, pb, acc); break;
    case 3: mfma_strip<3>(pa, pb, acc); break;
    default: mfma_strip<4>(pa, pb, acc); break;
    }
}

struct StepDiag { // what the sub-steps of the diagonal-block workgroup share
    double* sm;          // LDS
    double* Ablk;        // A + k0 * ld + k0: the 256 x 256 diagonal block
    int64_t ld;
    unsigned* fl;
    unsigned epoch;
    int* info;
    int z;
    int64_t K;
    int nsub;            // sub-steps on real columns (1 .. 4); the tiles of the others are padding: identity, zeros
    unsigned hu[3][SRK_STEP_MAXHOLD]; // this wave's hold units of the three updates (wave-uniform: scalar registers)
};

// (test hook: a lost hand-off.  The counter was read by ONE lane when the workgroup started -- with two attempt slots in flight
// another launch may change it between two lanes' reads -- into an LDS word the starting barrier published: nothing of it
// is on the chain)
__device__ __forceinline__ bool diag_fault(const StepDiag& S)
{
    return reinterpret_cast<const unsigned*>(S.sm + STP_CNT)[7] != 0;
}

// publication of L_DD and the X_tD (LDS tiles D .. 3 -> their places in A), write-through and coalesced (32 lanes store a whole
// row of a tile), by NW waves of 64: thread tp of them takes rows (tp >> 5) + (NW * 2) k
template <int D, int NW> __device__ __forceinline__ void diag_publish(const StepDiag& S, int tp)
{
    asm volatile("" : "+v"(tp)); // (opaque: no address of this is formed while potrf64 needs every register)
    const int r = tp >> 5, c = (tp & 31) * 2;
    const double* sT = S.sm + D * STP_TILE + r * LDSP + c;
    double* G = S.Ablk + (int64_t)(D * NB + r) * S.ld + D * NB + c;
#pragma unroll 1
    for (int b = D; b < 4; ++b) {
        // (all LDS reads of a tile first: the stores are asm statements with a memory clobber, nothing moves across them --
        // read, wait, store, read, ... took 1.8 - 2.4 us to ISSUE four tiles)
        double2 v[NB / (2 * NW)];
#pragma unroll
        for (int k = 0; k < NB / (2 * NW); ++k) v[k] = *reinterpret_cast<const double2*>(sT + 2 * NW * k * LDSP);
#pragma unroll
        for (int k = 0; k < NB / (2 * NW); ++k) {
            st_store16(G, (dbl2_t){ v[k].x, v[k].y });
            G += 2 * NW * S.ld;
        }
        sT += STP_TILE;
    }
}

// U(D): the rank-64 updates of the NEXT PANEL's tiles (t, D + 1), t > D, by all eight waves (sub-tiles dealt out by
// srk_step_tables.inc), then those tiles to LDS.  The tiles of the columns behind are the helper workgroups'.
template <int D, bool CHAIN> __device__ __forceinline__ void diag_update(const StepDiag& S)
{
    constexpr int HS = D == 0 ? 4 : D == 1 ? 3 : 2;
    int tid = threadIdx.x;
    // (opaque: nothing that depends on the thread index -- table entries, tile addresses -- is formed before this point, i.e.
    // while potrf64 needs every register)
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    double* const sm = S.sm;
    const int64_t ld = S.ld;
    const int64_t K = S.K; (void)K;
    double4_t hacc[HS][2];
    double hc[HS][2][4];
    unsigned hu[HS];
    // the tiles of block column D + 1 below its first one were updated with the X of the sub-steps before by the helper
    // workgroups, long ago (a whole factorisation lies between): one poll, by every wave for itself (a workgroup barrier here
    // would make the passenger waves wait for the chain waves) and before anything of this wave is in flight
    if (D >= 1) st_wait_wave(S.fl, D == 1 ? (1u << ST_G(0)) | (1u << ST_G(1)) : 1u << ST_G(2), S.epoch, S.info);
#pragma unroll
    for (int h = 0; h < HS; ++h) {
        const unsigned u = S.hu[D][h];
        hu[h] = u;
        const int t = u & 3, c = (u >> 2) & 3, r = (u >> 4) & 3, s0 = (u >> 6) & 3, ns = u >> 8;
        // the tiles' values so far (this workgroup's own data: earlier launches or its own stores before the last barriers)
        const double* pc = S.Ablk + (int64_t)(t * NB + 16 * r + lk) * ld + c * NB + 16 * s0 + lr;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) hc[h][s][reg] = s < ns ? pc[(int64_t)(4 * reg) * ld + 16 * s] : 0.0;
    }
    SST(29 + D);
#pragma unroll
    for (int h = 0; h < HS; ++h) {
        const unsigned u = hu[h];
        const int t = u & 3, c = (u >> 2) & 3, r = (u >> 4) & 3, s0 = (u >> 6) & 3, ns = u >> 8;
        double4_t acc[4];
        acc[0] = acc[1] = (double4_t){ 0, 0, 0, 0 };
        if (ns) { // wave-uniform
            const double* pa = sm + t * STP_TILE + (16 * r + lr) * LDSP + lk;
            const double* pb = sm + c * STP_TILE + (16 * s0 + lr) * LDSP + lk;
            if (ns == 1) mfma_strip<1>(pa, pb, acc);
            else mfma_strip<2>(pa, pb, acc);
        }
        hacc[h][0] = acc[0];
        hacc[h][1] = acc[1];
    }
    SST(6 + 6 * D);
    // publication: the chain waves issue the write-through stores of L_DD and the X_tD -- 128 KB at D = 0, which keep the CU's
    // memory pipe busy for ~3 us: issued before the loads above they held every wave's loads back by 2 - 4 us, so they come
    // BEHIND the chain waves' (small) share of the sub-tiles, while the passenger waves, which got the larger share for it
    // (srk_step_tables.inc), are still multiplying.  They drain under the rest of this function; F(D) is set at its end.
    if (CHAIN) diag_publish<D, 4>(S, tid);
    SST(26 + D);
    SST(4 + 6 * D);
    lds_barrier(); // every wave is done multiplying: the X tiles are free
    // the next panel's tiles to LDS (tile (D + 1, D + 1): zeros above the diagonal inside the sub-tiles on it; the sub-tiles
    // above keep stale numbers -- potrf64 never reads above the diagonal)
#pragma unroll
    for (int h = 0; h < HS; ++h) {
        const unsigned u = hu[h];
        const int t = u & 3, r = (u >> 4) & 3, s0 = (u >> 6) & 3, ns = u >> 8;
        double* pt = sm + t * STP_TILE + (16 * r + lk) * LDSP + 16 * s0 + lr;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (s < ns) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int rr = 16 * r + lk + 4 * reg, cc = 16 * (s0 + s) + lr;
                    pt[(4 * reg) * LDSP + 16 * s] = (t > D + 1 || cc <= rr) ? hc[h][s][reg] - hacc[h][s][reg] : 0.0;
                }
            }
    }
    lds_barrier();
    if (CHAIN) { // the last chain wave to see its stores drained sets F(D)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            unsigned* sCnt = reinterpret_cast<unsigned*>(sm + STP_CNT);
            const unsigned old = atomicAdd(&sCnt[D], 1u);
            if (old == 3) __hip_atomic_store(S.fl + ST_F(D), S.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    SST(5 + 6 * D);
}

// ---- the chain waves (0-3) of the diagonal-block workgroup, sub-step D
template <int D> __device__ __forceinline__ bool diag_chain_substep(const StepDiag& S)
{
    const int tid = threadIdx.x, lane = tid & 63;
    double* const sm = S.sm;
    double (*Td)[LDSP] = reinterpret_cast<double (*)[LDSP]>(sm + D * STP_TILE);
    double* sInv = sm + STP_INV;
    unsigned* sCnt = reinterpret_cast<unsigned*>(sm + STP_CNT);
    const int64_t K = S.K; (void)K;
    __builtin_amdgcn_s_setprio(3); // the chain's instructions before the passenger wave's of the same SIMD
    const bool bad = potrf64<true>(Td, sm + STP_COL, sm + STP_DIAG, sInv, sm + STP_F, sm + STP_RSQ);
    __builtin_amdgcn_s_setprio(0);
    if (bad && tid == 0) atomicOr(S.info, 1);
    SST(1 + 6 * D);
    if (D == 0 && diag_fault(S)) return false;
    // (the publication of L_DD and the X_tD is diag_update's first step; the last sub-step has no update: here.  The last
    // sub-step on real columns ends the chain: the tiles behind it are padding -- an identity diagonal, zeros below, which
    // the updates with the all-zero X rows of padding leave as they are -- so global memory already holds their L = I and
    // X = 0, and their flags go up with this one's: a 360-variable system pays for six tile factorisations, not eight)
    if (D == 3 || S.nsub == D + 1) {
        diag_publish<D, 4>(S, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned old = atomicAdd(&sCnt[D], 1u);
            if (old == 3)
                for (int d = D; d < NBO / NB; ++d) __hip_atomic_store(S.fl + ST_F(d), S.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SST(2 + 6 * D);
        return false;
    }
    return true;
}
static __device__ __forceinline__ void diag_chain(const StepDiag& S)
{
    if (!diag_chain_substep<0>(S)) return;
    diag_update<0, true>(S);
    if (!diag_chain_substep<1>(S)) return;
    diag_update<1, true>(S);
    if (!diag_chain_substep<2>(S)) return;
    diag_update<2, true>(S);
    diag_chain_substep<3>(S);
}

// ---- the passenger waves (4-7), sub-step D: thread (i, q) carries row i of the tiles (t, D), t = D + 1 .. 3
template <int D> __device__ __forceinline__ bool diag_passenger_substep(const StepDiag& S)
{
    constexpr int NP = 3 - D;
    const int lt = threadIdx.x & 255, i = lt >> 2, q = lt & 3;
    double* const sm = S.sm;
    const int64_t K = S.K; (void)K;
    double a[3][16];
#pragma unroll
    for (int p = 0; p < NP; ++p) { // (the initial load / the previous sub-step's update left the tiles in LDS)
        const double (*Tt)[LDSP] = reinterpret_cast<const double (*)[LDSP]>(sm + (D + 1 + p) * STP_TILE);
#pragma unroll
        for (int m = 0; m < 16; ++m) a[p][m] = Tt[i][4 * m + q];
    }
    potrf64_passengers<NP>(a, sm + STP_COL, sm + STP_F, sm + STP_RSQ, q);
    // X_tD to LDS (the updates' operands) -- before the chain's last two barriers, like its own store of L_DD
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        double (*Tt)[LDSP] = reinterpret_cast<double (*)[LDSP]>(sm + (D + 1 + p) * STP_TILE);
#pragma unroll
        for (int m = 0; m < 16; ++m) Tt[i][4 * m + q] = a[p][m];
    }
    lds_barrier();
    lds_barrier();
    SST(1 + 6 * D);
    if (D == 0 && diag_fault(S)) return false;
    if (S.nsub == D + 1) return false; // (the chain ends here: see diag_chain_substep)
    return true;
}
static __device__ __forceinline__ void diag_passengers(const StepDiag& S)
{
    if (!diag_passenger_substep<0>(S)) return;
    diag_update<0, false>(S);
    if (!diag_passenger_substep<1>(S)) return;
    diag_update<1, false>(S);
    if (!diag_passenger_substep<2>(S)) return;
    diag_update<2, false>(S);
    // (no rows are left below tile (3, 3): these waves end here, and the last factorisation's barriers count four waves)
}

__global__ __launch_bounds__(STP_THREADS) void k_step256(const CholBatch B, const CholStep rend, const CholStep r2b, const CholStep r2e,
                                                        int64_t K, unsigned* __restrict__ flags, unsigned epoch, int* __restrict__ info,
                                                        const CholSub nsubs)
{
    const int z = blockIdx.z;
    const int64_t row_end = rend.v[z];
    if (row_end < 0) return;
    double* __restrict__ A = B.it[z].A;
    double* __restrict__ w = B.it[z].w;
    double* __restrict__ y = B.it[z].y;
    const int64_t ld = B.it[z].ld, k0 = K * NBO;
    unsigned* fl = flags + ST_WORDS * z;
    const int role = blockIdx.x, tid = threadIdx.x;
    __shared__ __attribute__((aligned(16))) double sm[STP_LDS_DOUBLES];
    SST(0);
    if (role == 0) { // ---- the diagonal block
        StepDiag S{ sm, A + k0 * ld + k0, ld, fl, epoch, info, z, K, (int)nsubs.v[z], {} };
        {
            const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int h = 0; h < SRK_STEP_MAXHOLD; ++h) S.hu[d][h] = c_step_hold[d][wave_s][h];
        }
        // block column 0 to LDS, coalesced: tile (0, 0) (zeros above its diagonal) and the passenger tiles (t, 0)
        {
            unsigned* sCnt = reinterpret_cast<unsigned*>(sm + STP_CNT);
            if (tid < 7) sCnt[tid] = 0;
            if (tid == 7) { // (test hook, see diag_fault)
                int f = 0;
#ifdef SRK_DEV
                if (z == 0) {
                    f = g_step_fault;
                    if (f > 0) atomicSub(&g_step_fault, 1);
                }
#endif
                sCnt[7] = f > 0 ? 1u : 0u;
            }
            const int r = tid >> 3, cb = (tid & 7) * 8;
            double2 v[4][4];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[b][k] = reinterpret_cast<const double2*>(S.Ablk + (int64_t)(b * NB + r) * ld + cb)[k];
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = cb + 2 * k;
                    double* dst = sm + b * STP_TILE + r * LDSP + c;
                    dst[0] = (b > 0 || c <= r) ? v[b][k].x : 0.0;
                    dst[1] = (b > 0 || c + 1 <= r) ? v[b][k].y : 0.0;
                }
        }
        __syncthreads();
        if (tid < 256)
            diag_chain(S);
        else
            diag_passengers(S);
        return;
    }
    double (*sD)[LDSP] = reinterpret_cast<double (*)[LDSP]>(sm);
    double (*sA)[LDSP] = reinterpret_cast<double (*)[LDSP]>(sm + STP_TILE);
    if (role == 1) { // ---- inverses of the diagonal tiles, as they are published
        if (tid >= 256) return; // (tile_inverse is written for four waves)
        double (*sT)[33] = reinterpret_cast<double (*)[33]>(sm + 2 * STP_TILE);
        for (int d = 0; d < NBO / NB; ++d) {
            st_wait(fl, 1u << ST_F(d), epoch, info);
            st_load_tile<true>(sD, A + (k0 + d * NB) * ld + k0 + d * NB, ld);
            __syncthreads();
            tile_inverse(sD, reinterpret_cast<double (*)[NB + 1]>(&sA[0][0]), sT, sm + STP_DIAG, B.it[z].dinv + (K * (NBO / NB) + d) * NB * NB);
            __syncthreads();
        }
        return;
    }
    if (role == 2) { // ---- forward substitution of the block's own rows: y_d = L_dd^-1 w_d (64 dependent steps on one wave,
        // lane = row, its row of L in registers), published with flag Y(d); then w_t -= X_td y_d for the tile rows t > d below
        if (tid >= 256) return;
        double* sy = sm + STP_Y;
        const int i = tid >> 2, q = tid & 3;
        for (int d = 0; d < NBO / NB; ++d) {
            const double wv = tid < NB ? w[k0 + d * NB + tid] : 0.0; // (this workgroup's own updates below, or earlier kernels)
            st_wait(fl, 1u << ST_F(d), epoch, info);
            st_load_tile<true>(sD, A + (k0 + d * NB) * ld + k0 + d * NB, ld);
            __syncthreads();
            if (tid < NB) {
                double Lr[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) Lr[j] = sD[tid][j];
                const double myinv = fast_rcp(sD[tid][tid]);
                double v = wv;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    v = tid == j ? v * myinv : v; // y_j, on lane j
                    const double yj = readlane_f64(v, j);
                    v = tid > j ? fma(-Lr[j], yj, v) : v;
                }
                sy[tid] = v;
                st_store(y + k0 + d * NB + tid, v);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (tid == 0) __hip_atomic_store(fl + ST_Y(d), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            for (int t = d + 1; t < NBO / NB; ++t) {
                const int64_t row = k0 + t * NB + i;
                const double* X = A + row * ld + k0 + d * NB;
                double dot = 0;
#pragma unroll
                for (int m = 0; m < 16; ++m) dot = fma(X[4 * m + q], sy[4 * m + q], dot);
                dot += __shfl_xor(dot, 1, 64);
                dot += __shfl_xor(dot, 2, 64);
                if (q == 0) w[row] -= dot;
            }
            __syncthreads(); // (the w stores are complete before the next sub-step's load)
        }
        return;
    }
    if (role == 3) { // ---- helper workgroup: the tiles of the diagonal block behind the next panel, (t, c) = (2, 2), (3, 2),
        // (3, 3).  Tile (t, c) takes C -= X_td X_cd^T for the sub-steps d <= c - 2 here (the update of sub-step c - 1 is the
        // diagonal-block workgroup's: by then the tile is the next panel); a tile of MFMA work each, with a whole
        // factorisation of slack, then flag G(k).  (One workgroup for the three tiles: a launch's workgroup count decides
        // whether the two solves of a speculative pair fit the chip side by side.)
        const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
        const int sr = wave >> 1, sc = (wave & 1) * 32; // a wave takes a 16 x 32 strip
        double* Ablk = A + k0 * ld + k0;
        auto update = [&](int t, int c, int d) {
            double* pc0 = Ablk + (int64_t)(t * NB + 16 * sr + lk) * ld + c * NB + sc + lr;
            double cv[2][4];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) cv[n][reg] = pc0[(int64_t)(4 * reg) * ld + n * 16];
            st_load_tile512(sD, Ablk + (int64_t)(t * NB) * ld + d * NB, ld);
            if (t != c) st_load_tile512(sA, Ablk + (int64_t)(c * NB) * ld + d * NB, ld);
            __syncthreads();
            double4_t acc[4];
            mfma_strip<2>(&sD[16 * sr + lr][lk], t != c ? &sA[sc + lr][lk] : &sD[sc + lr][lk], acc);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) st_store(pc0 + (int64_t)(4 * reg) * ld + n * 16, cv[n][reg] - acc[n][reg]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads(); // (every wave has drained its stores; the LDS tiles are free)
        };
        st_wait(fl, 1u << ST_F(0), epoch, info);
        update(2, 2, 0);
        if (tid == 0) __hip_atomic_store(fl + ST_G(0), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        update(3, 2, 0);
        if (tid == 0) __hip_atomic_store(fl + ST_G(1), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        update(3, 3, 0);
        st_wait(fl, 1u << ST_F(1), epoch, info);
        update(3, 3, 1);
        if (tid == 0) __hip_atomic_store(fl + ST_G(2), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // (row workgroups of 128 rows -- two groups of 64, all eight waves multiplying -- were measured: a first-level launch
    // then has 128 instead of 192 workgroups and the two solves of a speculative pair fit the chip side by side, but a row
    // workgroup's sub-step (sweep + twice the MFMA work) then takes as long as the diagonal block's and the launch tail grows:
    // solve 0.921 against 0.847 ms, pairs unchanged at 1.93 ms)
    // ---- row workgroups
    int64_t r0;
    {
        int64_t n1 = (row_end - (k0 + NBO)) / NB;
        if (n1 < 0) n1 = 0;
        const int64_t r2_begin = r2b.v[z], n2 = (r2e.v[z] - r2_begin) / NB, ridx = role - 4;
        if (ridx >= n1 + n2) return;
        r0 = ridx < n1 ? k0 + NBO + ridx * NB : r2_begin + (ridx - n1) * NB;
    }
    double* sInv = sm + STP_INV;
    double* sy = sm + STP_Y;
    const bool sweeper = tid < 256; // waves 0-3 sweep (a row per quad); all eight waves multiply
    const int i = (tid & 255) >> 2, q = tid & 3;
    double* rowp = A + (r0 + i) * ld + k0; // this thread's row, first column of the outer block
    double wi = (sweeper && q == 0) ? w[r0 + i] : 0.0;
    const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    // forward substitution's update of this workgroup's rows with the PREVIOUS sub-step's panel, w_r -= X[r, tile p] . y_p:
    // y_p appears a few microseconds after L_pp, so its wait rides on the next wait this workgroup has anyway; X of the
    // previous sub-step is still in sA then
    auto w_update = [&](int p) {
        if (tid < NB) sy[tid] = y[k0 + p * NB + tid];
        __syncthreads();
        double dot = 0;
#pragma unroll
        for (int m = 0; m < 16; ++m) dot = fma(sA[i][4 * m + q], sy[4 * m + q], dot);
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        wi -= dot;
    };
    for (int d = 0; d < NBO / NB; ++d) {
        double a[16];
        if (sweeper) {
#pragma unroll
            for (int m = 0; m < 16; ++m) a[m] = rowp[d * NB + 4 * m + q]; // own data (earlier kernels / this workgroup's updates)
        }
        SST(1 + 6 * d);
        st_wait(fl, (1u << ST_F(d)) | (d > 0 ? 1u << ST_Y(d - 1) : 0u), epoch, info);
        SST(2 + 6 * d);
        if (d > 0) w_update(d - 1);
        st_load_tile512(sD, A + (k0 + d * NB) * ld + k0 + d * NB, ld);
        __syncthreads();
        if (tid < NB) sInv[tid] = fast_rcp(sD[tid][tid]);
        __syncthreads();
        SST(3 + 6 * d);
        if (sweeper) {
            // X = A[rows, tile d] L_dd^-T
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const int c0 = 4 * b;
                double v0 = quad_bcast<0>(a[b]), v1 = quad_bcast<1>(a[b]), v2 = quad_bcast<2>(a[b]), v3 = quad_bcast<3>(a[b]);
                double x0 = v0 * sInv[c0];
                double x1 = fma(-x0, sD[c0 + 1][c0], v1) * sInv[c0 + 1];
                double x2 = fma(-x1, sD[c0 + 2][c0 + 1], fma(-x0, sD[c0 + 2][c0], v2)) * sInv[c0 + 2];
                double x3 = fma(-x2, sD[c0 + 3][c0 + 2], fma(-x1, sD[c0 + 3][c0 + 1], fma(-x0, sD[c0 + 3][c0], v3))) * sInv[c0 + 3];
                {
                    const double x01 = (q & 1) ? x1 : x0, x23 = (q & 1) ? x3 : x2;
                    a[b] = (q & 2) ? x23 : x01;
                }
#pragma unroll
                for (int m = b + 1; m < 16; ++m) {
                    const double2* lp = reinterpret_cast<const double2*>(&sD[4 * m + q][c0]);
                    double2 l01 = lp[0], l23 = lp[1];
                    a[m] = fma(-x3, l23.y, fma(-x2, l23.x, fma(-x1, l01.y, fma(-x0, l01.x, a[m]))));
                    if (((m - b) & 3) == 0) asm volatile("" ::: "memory");
                }
                asm volatile("" ::: "memory");
            }
            // X is final: to sA (the updates' A operand; the next w update reads it there) and to its place in A
#pragma unroll
            for (int m = 0; m < 16; ++m) sA[i][4 * m + q] = a[m];
#pragma unroll
            for (int m = 0; m < 16; ++m) rowp[d * NB + 4 * m + q] = a[m];
        }
        SST(4 + 6 * d);
        // rank-64 update of this workgroup's tiles (., t), t = d + 1 .. 3:  C -= X X_td^T.  The B operand X_td (published with
        // L_dd) goes through two LDS buffers; the next one is fetched into registers while this one is multiplied.  A wave
        // takes a 16 x 32 strip of the 64 x 64 tile.
        const int bi = tid >> 3, bc = (tid & 7) * 8;
        double2 bv[4];
        auto fetch_b = [&](int t) {
            const double2* src = reinterpret_cast<const double2*>(A + (k0 + t * NB + bi) * ld + k0 + d * NB + bc);
#pragma unroll
            for (int k = 0; k < 4; ++k) bv[k] = src[k];
        };
        if (d + 1 < NBO / NB) fetch_b(d + 1);
        const int sr = wave >> 1, sc = (wave & 1) * 32;
        for (int t = d + 1; t < NBO / NB; ++t) {
            double (*sB)[LDSP] = reinterpret_cast<double (*)[LDSP]>(sm + (2 + ((t - d - 1) & 1)) * STP_TILE);
            double* pc0 = A + (r0 + 16 * sr + lk) * ld + k0 + t * NB + sc + lr;
            double cv[2][4];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) cv[n][reg] = pc0[(int64_t)(4 * reg) * ld + n * 16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sB[bi][bc + 2 * k] = bv[k].x;
                sB[bi][bc + 2 * k + 1] = bv[k].y;
            }
            __syncthreads(); // (also: sA is written; every wave is done with the buffer of two tiles ago)
            if (t + 1 < NBO / NB) fetch_b(t + 1);
            double4_t acc[4];
            mfma_strip<2>(&sA[16 * sr + lr][lk], &sB[sc + lr][lk], acc);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) pc0[(int64_t)(4 * reg) * ld + n * 16] = cv[n][reg] - acc[n][reg];
        }
        __syncthreads(); // the updates' stores are complete before the next sub-step loads its columns; sD, sA are free
        SST(5 + 6 * d);
    }
    st_wait(fl, 1u << ST_Y(NBO / NB - 1), epoch, info);
    w_update(NBO / NB - 1);
    SST(25);
    if (sweeper && q == 0) w[r0 + i] = wi;
}

// ---------------------------------------------------------------- trailing update of an outer panel (MFMA, K = 256)
// C[ti, tj] -= P[ti] P[tj]^T over the 128x128 tile pairs ti >= tj of rows/cols [c_first, row_end), P = the 256
// panel columns starting at k0.  One workgroup per tile pair (linear index -> triangular pair).
__global__ __launch_bounds__(256, 2) void k_trail(const CholBatch B, const CholStep rend, const CholStep r2b, const CholStep r2e,
                                                   int64_t k0, int64_t c_first)
{
    __shared__ double sA[2][TL][KCP];
    __shared__ double sB[2][TL][KCP];
    if (rend.v[blockIdx.z] < 0) return;
    double* __restrict__ A = B.it[blockIdx.z].A;
    const int64_t ld = B.it[blockIdx.z].ld, r2_begin = r2b.v[blockIdx.z];
    int64_t T1 = (rend.v[blockIdx.z] - c_first) / TL;
    if (T1 < 0) T1 = 0;
    int64_t p = blockIdx.x;
    {
        const int64_t T = T1 + (r2e.v[blockIdx.z] - r2_begin) / TL;
        if (p >= T * (T + 1) / 2) return;
    }
    int ti = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= p) ++ti;
    while ((int64_t)ti * (ti + 1) / 2 > p) --ti;
    int tj = (int)(p - (int64_t)ti * (ti + 1) / 2);
    // tile i of the list: T1 tiles from c_first (inside the skyline), then the border tiles from r2_begin
    const int64_t r0 = ti < T1 ? c_first + (int64_t)ti * TL : r2_begin + ((int64_t)ti - T1) * TL;
    const int64_t c0 = tj < T1 ? c_first + (int64_t)tj * TL : r2_begin + ((int64_t)tj - T1) * TL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    // staging map: thread t loads row (t >> 1), half (t & 1) of a KC-wide chunk: 8 doubles = 4 x 16 B
    const int srow = threadIdx.x >> 1, shalf = (threadIdx.x & 1) * 8;
    const double* ga = A + (r0 + srow) * ld + k0 + shalf;
    const double* gb = A + (c0 + srow) * ld + k0 + shalf;
    double2 ra[4], rb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ra[t] = reinterpret_cast<const double2*>(ga)[t];
        rb[t] = reinterpret_cast<const double2*>(gb)[t];
    }
    double4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (double4_t){ 0, 0, 0, 0 };
    const int nchunk = NBO / KC;
    const double* pa = &sA[0][wr * 64 + lr][lk];
    const double* pb = &sB[0][wc * 64 + lr][lk];
    for (int ch = 0; ch < nchunk; ++ch) {
        const int boff = (ch & 1) * TL * KCP;
        double* wa = &sA[0][srow][shalf] + boff;
        double* wb = &sB[0][srow][shalf] + boff;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            wa[2 * t] = ra[t].x;
            wa[2 * t + 1] = ra[t].y;
            wb[2 * t] = rb[t].x;
            wb[2 * t + 1] = rb[t].y;
        }
        __syncthreads();
        if (ch + 1 < nchunk) { // prefetch the next chunk from L2/HBM while this one is multiplied
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ra[t] = reinterpret_cast<const double2*>(ga + (ch + 1) * KC)[t];
                rb[t] = reinterpret_cast<const double2*>(gb + (ch + 1) * KC)[t];
            }
        }
        // operand ping-pong: the LDS reads of k-step kk+1 are in flight while the 16 MFMAs of kk issue
        double av[2][4], bv[2][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            av[0][m] = pa[boff + m * 16 * KCP];
            bv[0][m] = pb[boff + m * 16 * KCP];
        }
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KC / 4) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    av[nxt][m] = pa[boff + m * 16 * KCP + (kk + 1) * 4];
                    bv[nxt][m] = pb[boff + m * 16 * KCP + (kk + 1) * 4];
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[cur][m], bv[cur][n], acc[m][n], 0, 0, 0);
        }
        // the buffer written two iterations from now is this one: the barrier of the next iteration orders it
    }
    // f64 16x16x4 accumulator map: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* pc0 = A + (r0 + wr * 64 + lk) * ld + c0 + wc * 64 + lr;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            double* prow = pc0 + (int64_t)(m * 16 + 4 * reg) * ld;
            double c4[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) c4[n] = prow[n * 16];
#pragma unroll
            for (int n = 0; n < 4; ++n) prow[n * 16] = c4[n] - acc[m][n][reg];
        }
}

// ---------------------------------------------------------------- trailing update, small-skyline variant
// Same contraction as k_trail on 64x64 tiles (4x the workgroups, a quarter of the serial work each): when only a few
// 128-tiles are inside the skyline the update is latency-bound and the grid, not the tile shape, sets its time.
__global__ __launch_bounds__(256) void k_trail64(const CholBatch B, const CholStep rend, const CholStep r2b, const CholStep r2e,
                                                  int64_t k0, int64_t c_first)
{
    __shared__ double sA[NB][LDSP];
    __shared__ double sB[NB][LDSP];
    if (rend.v[blockIdx.z] < 0) return;
    double* __restrict__ A = B.it[blockIdx.z].A;
    const int64_t ld = B.it[blockIdx.z].ld, r2_begin = r2b.v[blockIdx.z];
    int64_t T1 = (rend.v[blockIdx.z] - c_first) / NB;
    if (T1 < 0) T1 = 0;
    int64_t p = blockIdx.x;
    {
        const int64_t T = T1 + (r2e.v[blockIdx.z] - r2_begin) / NB;
        if (p >= T * (T + 1) / 2) return;
    }
    int ti = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= p) ++ti;
    while ((int64_t)ti * (ti + 1) / 2 > p) --ti;
    int tj = (int)(p - (int64_t)ti * (ti + 1) / 2);
    const int64_t r0 = ti < T1 ? c_first + (int64_t)ti * NB : r2_begin + ((int64_t)ti - T1) * NB;
    const int64_t c0 = tj < T1 ? c_first + (int64_t)tj * NB : r2_begin + ((int64_t)tj - T1) * NB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    const int row = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;
    double4_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){ 0, 0, 0, 0 };
    // the C tile is read-modify-write: fetch it first, its latency hides under the four K chunks
    double* pc0 = A + (r0 + wr * 32 + lk) * ld + c0 + wc * 32 + lr;
    double cv[2][2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) cv[m][n][reg] = pc0[(int64_t)(m * 16 + 4 * reg) * ld + n * 16];
    // software pipeline over the four 64-deep K chunks: the next chunk's global loads are in flight during the MFMAs
    const double* pa = A + (r0 + row) * ld + k0 + seg;
    const double* pb = A + (c0 + row) * ld + k0 + seg;
    double2 va[8], vb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        va[t] = reinterpret_cast<const double2*>(pa)[t];
        vb[t] = reinterpret_cast<const double2*>(pb)[t];
    }
#pragma unroll
    for (int ch = 0; ch < NBO / NB; ++ch) {
        __syncthreads(); // previous chunk fully consumed
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            sA[row][seg + 2 * t] = va[t].x;
            sA[row][seg + 2 * t + 1] = va[t].y;
            sB[row][seg + 2 * t] = vb[t].x;
            sB[row][seg + 2 * t + 1] = vb[t].y;
        }
        if (ch + 1 < NBO / NB) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                va[t] = reinterpret_cast<const double2*>(pa + (ch + 1) * NB)[t];
                vb[t] = reinterpret_cast<const double2*>(pb + (ch + 1) * NB)[t];
            }
        }
        lds_barrier(); // LDS only: the prefetch stays in flight
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk) {
            double a0 = sA[wr * 32 + lr][kk * 4 + lk];
            double a1 = sA[wr * 32 + 16 + lr][kk * 4 + lk];
            double b0 = sB[wc * 32 + lr][kk * 4 + lk];
            double b1 = sB[wc * 32 + 16 + lr][kk * 4 + lk];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) pc0[(int64_t)(m * 16 + 4 * reg) * ld + n * 16] = cv[m][n][reg] - acc[m][n][reg];
}

// ---------------------------------------------------------------- trailing update, few-tiles variant
// The same contraction on 32 x 32 tiles, one 16 x 16 MFMA tile per wave.  At the deeper levels of a nested solve only the
// 512 x 512 border block of two to eight chunks is updated: 36 tile pairs of 64 x 64 a chunk leave most of the chip idle
// while every workgroup works through 256 MFMAs per wave (7.8 us) behind four dependent K chunks of loads (k_trail64:
// 19-31 us a launch there).  Here a workgroup has a quarter of that work, and ALL four K chunks of its operands are
// requested up front (one memory round trip: 128 registers of loads in flight).  Same products in the same order per entry
// of C as k_trail64 (chunks ascending, k ascending inside): bit-identical results.
#define NT 32
__global__ __launch_bounds__(256) void k_trail32(const CholBatch B, const CholStep rend, const CholStep r2b, const CholStep r2e,
                                                  int64_t k0, int64_t c_first)
{
    __shared__ double sA[NT][LDSP];
    __shared__ double sB[NT][LDSP];
    if (rend.v[blockIdx.z] < 0) return;
    double* __restrict__ A = B.it[blockIdx.z].A;
    const int64_t ld = B.it[blockIdx.z].ld, r2_begin = r2b.v[blockIdx.z];
    int64_t T1 = (rend.v[blockIdx.z] - c_first) / NT;
    if (T1 < 0) T1 = 0;
    int64_t p = blockIdx.x;
    {
        const int64_t T = T1 + (r2e.v[blockIdx.z] - r2_begin) / NT;
        if (p >= T * (T + 1) / 2) return;
    }
    int ti = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= p) ++ti;
    while ((int64_t)ti * (ti + 1) / 2 > p) --ti;
    int tj = (int)(p - (int64_t)ti * (ti + 1) / 2);
    const int64_t r0 = ti < T1 ? c_first + (int64_t)ti * NT : r2_begin + ((int64_t)ti - T1) * NT;
    const int64_t c0 = tj < T1 ? c_first + (int64_t)tj * NT : r2_begin + ((int64_t)tj - T1) * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    // staging map of a 64-deep K chunk: thread -> row (t >> 3) of 32, eight doubles from column 8 (t & 7)
    const int row = threadIdx.x >> 3, seg = (threadIdx.x & 7) * 8;
    const double* pa = A + (r0 + row) * ld + k0 + seg;
    const double* pb = A + (c0 + row) * ld + k0 + seg;
    double2 va[NBO / NB][4], vb[NBO / NB][4];
#pragma unroll
    for (int ch = 0; ch < NBO / NB; ++ch)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            va[ch][t] = reinterpret_cast<const double2*>(pa + ch * NB)[t];
            vb[ch][t] = reinterpret_cast<const double2*>(pb + ch * NB)[t];
        }
    // the C tile is read-modify-write: f64 16x16x4 accumulator map col = lane & 15, row = (lane >> 4) + 4 reg
    double* pc0 = A + (r0 + wr * 16 + lk) * ld + c0 + wc * 16 + lr;
    double cv[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cv[reg] = pc0[(int64_t)(4 * reg) * ld];
    double4_t acc = (double4_t){ 0, 0, 0, 0 };
#pragma unroll
    for (int ch = 0; ch < NBO / NB; ++ch) {
        if (ch) __syncthreads(); // previous chunk fully consumed
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sA[row][seg + 2 * t] = va[ch][t].x;
            sA[row][seg + 2 * t + 1] = va[ch][t].y;
            sB[row][seg + 2 * t] = vb[ch][t].x;
            sB[row][seg + 2 * t + 1] = vb[ch][t].y;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[wr * 16 + lr][kk * 4 + lk], sB[wc * 16 + lr][kk * 4 + lk], acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) pc0[(int64_t)(4 * reg) * ld] = cv[reg] - acc[reg];
}

// ---------------------------------------------------------------- backward substitution L^T x = y
// (the inverses of the diagonal tiles come from the inverse workgroup of k_panel)
// step K (descending, 256 rows): x_K = L_KK^-T y_K by four tile back-substitutions with the explicit tile inverses
// (every workgroup, redundantly), then y_j -= sum_i L[256 K + i, j] x_K[i] for this workgroup's 64 columns
// j in [col_begin, 256 K).  Every product is "one column per lane, the rows split over the four waves, partial sums
// combined through LDS": short independent load chains instead of 64- and 256-long ones.
#define BWD_COLS 64
// finite: the reference's allFinite check of the solution (:1912-1913) where this level's x IS the final solution (NULL below)
__global__ __launch_bounds__(256) void k_bwd256(const CholBatch B, const CholStep Kst, const CholStep cbeg, int* __restrict__ finite)
{
    __shared__ double sv[NBO];
    __shared__ double sp[4][NBO - NB];
    const int64_t K = Kst.v[blockIdx.z], col_begin = cbeg.v[blockIdx.z];
    if (K < 0) return;
    const int64_t k0 = K * NBO;
    {
        const int64_t cols = k0 - col_begin;
        if ((int64_t)blockIdx.x >= (cols > 0 ? (cols + BWD_COLS - 1) / BWD_COLS : 1)) return;
    }
    const double* __restrict__ A = B.it[blockIdx.z].A;
    const double* __restrict__ Dinv = B.it[blockIdx.z].dinv;
    double* __restrict__ y = B.it[blockIdx.z].y;
    double* __restrict__ x = B.it[blockIdx.z].x;
    const int64_t ld = B.it[blockIdx.z].ld;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    sv[t] = y[k0 + t];
    __syncthreads();
#pragma unroll
    for (int sub = NBO / NB - 1; sub >= 0; --sub) {
        // x_sub = Dinv_sub^T v_sub : lane = entry i, wave wv sums m in [16 wv, 16 wv + 16) (coalesced rows of Dinv)
        {
            const double* D = Dinv + (K * (NBO / NB) + sub) * NB * NB + (int64_t)(16 * wv) * NB + lane;
            double part = 0;
#pragma unroll
            for (int m = 0; m < 16; ++m) part = fma(D[m * NB], sv[sub * NB + 16 * wv + m], part);
            sp[wv][lane] = part;
        }
        __syncthreads();
        if (t < NB) sv[sub * NB + t] = (sp[0][t] + sp[1][t]) + (sp[2][t] + sp[3][t]);
        __syncthreads();
        if (sub == 0) break;
        // v_c -= sum_r L[tile sub row r, col c] x_sub[r] for the columns c of the earlier tiles of this 256 block
        {
            const double* Lr = A + (k0 + sub * NB + 16 * wv) * ld + k0 + lane;
            double acc[NBO / NB - 1];
#pragma unroll
            for (int p = 0; p < NBO / NB - 1; ++p) acc[p] = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double xr = sv[sub * NB + 16 * wv + r];
#pragma unroll
                for (int p = 0; p < NBO / NB - 1; ++p)
                    if (p < sub) acc[p] = fma(Lr[(int64_t)r * ld + p * NB], xr, acc[p]);
            }
#pragma unroll
            for (int p = 0; p < NBO / NB - 1; ++p)
                if (p < sub) sp[wv][p * NB + lane] = acc[p];
        }
        __syncthreads();
        if (t < sub * NB) sv[t] -= (sp[0][t] + sp[1][t]) + (sp[2][t] + sp[3][t]);
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        x[k0 + t] = sv[t];
        if (finite && !isfinite(sv[t])) atomicOr(finite, 4);
    }
    const int64_t j = col_begin + (int64_t)blockIdx.x * BWD_COLS + lane;
    double acc = 0;
    if (j < k0) {
        const double* Lc = A + (k0 + 64 * wv) * ld + j;
#pragma unroll 16
        for (int r = 0; r < 64; ++r) acc = fma(Lc[(int64_t)r * ld], sv[64 * wv + r], acc);
    }
    sp[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && j < k0) y[j] -= (sp[0][lane] + sp[1][lane]) + (sp[2][lane] + sp[3][lane]);
}

// ---------------------------------------------------------------- host drivers
// SrkSolveProf (optional): event pairs around every MFMA trailing-update launch and the flops those launches execute;
// in `dry` mode nothing is launched and only the flops are counted.
#define LAUNCH(...) do { if (!(prof && prof->dry)) hipLaunchKernelGGL(__VA_ARGS__); } while (0)
// chol_factor: for every item of the batch, eliminate its first `ncols` columns (multiple of 256).  Rows taking part
// in outer panel K are the skyline rows [256 K, row_end[K]) (clipped to ncols) plus the border rows
// [r2_begin, r2_end) (multiples of 128; empty when r2_begin == r2_end).  The forward substitution of w rides along
// (k_panel); y receives L^-1 w for the eliminated columns, the border part of w its Schur-complement update.
// Items advance in lock step: launch K serves outer panel K of every item that has one.
static void chol_factor(hipStream_t s, const CholBatch& B, int n, const CholHostItem* H, int* d_info, SrkSolveProf* prof,
                        SrkCholSync* sync)
{
    int64_t nout = 0;
    for (int i = 0; i < n; ++i) nout = std::max(nout, B.it[i].ncols / NBO);
    for (int64_t K = 0; K < nout; ++K) {
        const int64_t k0 = K * NBO;
        CholStep st, r2b, r2e;
        for (int i = 0; i < SRK_MAX_CHUNKS; ++i) st.v[i] = -1, r2b.v[i] = r2e.v[i] = 0;
        for (int i = 0; i < n; ++i) {
            const int64_t ncols = B.it[i].ncols;
            if (K >= ncols / NBO) continue;
            int64_t rend = H[i].row_end ? H[i].row_end[K] : ncols;
            if (rend < k0 + NBO) rend = k0 + NBO;
            if (rend > ncols) rend = ncols;
            st.v[i] = rend;
            // live border rows of this step
            r2b.v[i] = B.it[i].r2_begin;
            r2e.v[i] = B.it[i].r2_end;
            if (B.it[i].r2_end > B.it[i].r2_begin) {
                if (!H[i].has_top) r2b.v[i] = H[i].r2_split;
                if (!H[i].has_bot || k0 + NBO <= H[i].bot_first_col) r2e.v[i] = H[i].r2_split;
                if (r2e.v[i] < r2b.v[i]) r2e.v[i] = r2b.v[i];
            }
        }
        // the four inner panels as ONE launch (k_step256) when its workgroups stay within one wave of the chip
        int64_t row_wgs = 0;
        for (int i = 0; i < n; ++i) {
            if (st.v[i] < 0) continue;
            int64_t n1 = (st.v[i] - (k0 + NBO)) / NB;
            if (n1 < 0) n1 = 0;
            row_wgs = std::max(row_wgs, n1 + (r2e.v[i] - r2b.v[i]) / NB);
        }
        const bool fused = sync && sync->flags && sync->fused && (4 + row_wgs) * n <= 256;
        if (fused) {
            ++sync->epoch;
            if (sync->epoch == 0) ++sync->epoch; // the flag words start at 0
            CholSub nsubs;
            for (int i = 0; i < SRK_MAX_CHUNKS; ++i) {
                const int64_t real = i < n && H[i].n_real > k0 ? (std::min(H[i].n_real, k0 + NBO) - k0 + NB - 1) / NB : NBO / NB;
                nsubs.v[i] = (unsigned char)std::max<int64_t>(1, std::min<int64_t>(NBO / NB, real));
            }
            LAUNCH(k_step256, dim3((unsigned)(4 + row_wgs), 1, (unsigned)n), dim3(STP_THREADS), 0, s, B, st, r2b, r2e, K, sync->flags,
                   sync->epoch, d_info, nsubs);
        }
        for (int jsub = 0; jsub < NBO / NB && !fused; ++jsub) {
            const int64_t d = K * (NBO / NB) + jsub;
            int64_t blocks = 0, tiles = 0;
            for (int i = 0; i < n; ++i) {
                if (st.v[i] < 0) continue;
                int64_t rows1 = st.v[i] - (d + 1) * NB;
                if (rows1 < 0) rows1 = 0;
                const int64_t rows2 = r2e.v[i] - r2b.v[i];
                const int64_t rows = rows1 + rows2;
                if (rows > 0) blocks = std::max(blocks, (rows + PANEL_ROWS - 1) / PANEL_ROWS);
                tiles = std::max(tiles, rows1 / NB + rows2 / NB);
            }
            LAUNCH(k_panel, dim3((unsigned)(blocks + 1), 1, (unsigned)n), dim3(256), 0, s, B, st, r2b, r2e, d, d_info); // + inverse workgroup
            const int64_t c_hi = K * (NBO / NB) + (NBO / NB - 1);
            if (jsub < NBO / NB - 1 && tiles > 0)
                LAUNCH(k_upd64, dim3((unsigned)tiles, (unsigned)(c_hi - d), (unsigned)n), dim3(256), 0, s, B,
                                   st, r2b, r2e, d, c_hi);
        }
        const int64_t c_first = k0 + NBO;
        int64_t T = 0;
        double flops = 0;
        for (int i = 0; i < n; ++i) {
            if (st.v[i] < 0) continue;
            int64_t T1 = (st.v[i] - c_first) / TL;
            if (T1 < 0) T1 = 0;
            const int64_t Ti = T1 + (r2e.v[i] - r2b.v[i]) / TL;
            T = std::max(T, Ti);
            flops += (double)(Ti * (Ti + 1) / 2) * (double)TL * (double)TL * (double)NBO * 2.0;
        }
        const bool timed = prof && !prof->dry && T > 0 && 2 * prof->n + 1 < prof->cap;
        if (prof && T > 0) prof->flops += flops;
        if (timed) hipEventRecord(prof->ev[2 * prof->n], s);
        int n_live = 0;
        for (int i = 0; i < n; ++i) n_live += st.v[i] >= 0 ? 1 : 0;
        // small blocks (T <= 4: at most 512 rows): 32x32 tiles, 16x the workgroups.  (Round 3 took this kernel for up to 320
        // pairs of 64 x 64 tiles in a launch; round 4: for every such launch -- k_trail64 is latency-bound at the one to five
        // workgroups a CU these launches have: C3's first level, 16 and 9 chunks of 36 pairs, 45.8 + 31.9 us against ~37 + 23
        // (solve phase 0.859 -> 0.837 ms), C5's, 32 chunks, 2.065 -> 2.02 ms)
        static const int64_t trail32_max = [] {
#ifdef SRK_DEV
            if (const char* e = getenv("SRK_TRAIL32_MAX")) return (int64_t)atoll(e); // development: A/B runs
#endif
            return (int64_t)1 << 40;
        }();
        static const int64_t trail32_t = [] {
#ifdef SRK_DEV
            if (const char* e = getenv("SRK_TRAIL32_T")) return (int64_t)atoll(e);
#endif
            return (int64_t)4;
        }();
        if (T > 0 && T <= trail32_t && (int64_t)n_live * (2 * T) * (2 * T + 1) / 2 <= trail32_max) {
            const int64_t T32 = 4 * T;
            LAUNCH(k_trail32, dim3((unsigned)(T32 * (T32 + 1) / 2), 1, (unsigned)n), dim3(256), 0, s, B, st, r2b, r2e, k0,
                               c_first);
        } else if (T > 0 && T <= 8) { // narrow skyline: 64x64 tiles, 4x the workgroups
            const int64_t T64 = 2 * T;
            LAUNCH(k_trail64, dim3((unsigned)(T64 * (T64 + 1) / 2), 1, (unsigned)n), dim3(256), 0, s, B, st, r2b, r2e, k0,
                               c_first);
        } else if (T > 0) {
            LAUNCH(k_trail, dim3((unsigned)(T * (T + 1) / 2), 1, (unsigned)n), dim3(256), 0, s, B, st, r2b, r2e, k0,
                               c_first);
        }
        if (timed) hipEventRecord(prof->ev[2 * prof->n + 1], s), ++prof->n;
    }
}

// chol_bwd: x = L^-T y over the first `ncols` columns of every item (the border part, if any, has been folded into y
// already).  Launch t serves outer panel nout_i - 1 - t of item i.
static void chol_bwd(hipStream_t s, const CholBatch& B, int n, const CholHostItem* H, SrkSolveProf* prof, int* d_finite)
{
    int64_t nout = 0;
    for (int i = 0; i < n; ++i) nout = std::max(nout, B.it[i].ncols / NBO);
    for (int64_t t = 0; t < nout; ++t) {
        CholStep Kst, cbeg;
        int64_t blocks = 1;
        for (int i = 0; i < SRK_MAX_CHUNKS; ++i) Kst.v[i] = -1, cbeg.v[i] = 0;
        for (int i = 0; i < n; ++i) {
            const int64_t K = B.it[i].ncols / NBO - 1 - t;
            if (K < 0) continue;
            const int64_t* col_begin = H[i].col_begin;
            int64_t cb = col_begin ? col_begin[K * (NBO / NB)] : 0;
            for (int q = 1; q < NBO / NB; ++q)
                if (col_begin) cb = std::min(cb, col_begin[K * (NBO / NB) + q]);
            if (cb > K * NBO) cb = K * NBO;
            const int64_t cols = K * NBO - cb;
            if (cols > 0) blocks = std::max(blocks, (cols + BWD_COLS - 1) / BWD_COLS);
            Kst.v[i] = K;
            cbeg.v[i] = cb;
        }
        LAUNCH(k_bwd256, dim3((unsigned)blocks, 1, (unsigned)n), dim3(256), 0, s, B, Kst, cbeg, d_finite);
    }
}

// w: right-hand side (destroyed).  y: scratch (forward solution).  x: solution.
// row_end[K] (host, one per outer panel, multiple of 128): rows >= row_end[K] have no non-zero in the panel's
// columns and are skipped; NULL = dense.  col_begin[d64] (host, per 64-tile, may be NULL): first column with a
// non-zero in tile row d64.  dinv: scratch, (ld / 64) * 64 * 64 doubles (inverses of the diagonal tiles).
// n_real: rows and columns from there on are padding (identity diagonal, zero right-hand side); 0 = none
void srk_chol_solve(hipStream_t s, int64_t ld, double* A, double* w, double* y, double* x, int* d_info,
                    const int64_t* row_end, const int64_t* col_begin, double* dinv, SrkSolveProf* prof, SrkCholSync* sync,
                    int64_t n_real)
{
    CholBatch B{};
    B.it[0] = CholItem{ A, w, y, x, dinv, ld, ld, ld, ld };
    CholHostItem H;
    H.row_end = row_end;
    H.col_begin = col_begin;
    if (n_real > 0 && n_real < ld) H.n_real = (n_real + NB - 1) / NB * NB;
    chol_factor(s, B, 1, &H, d_info, prof, sync);
    chol_bwd(s, B, 1, &H, prof, d_info); // every variable is written by a k_bwd256 step, which checks it
}

// ================================================================ chunked (bordered block-diagonal) solve
// A banded reduced camera system is a chain of n dependent pivots.  Cut it into P chunks separated by P-1 separators
// of `sepw` variables (sepw >= the bandwidth, so chunks do not touch each other): every chunk
//     [ A_c   B_c^T ]      A_c : the chunk (skyline),  B_c : its couplings to the separator above and below
//     [ B_c    0    ]
// is factorised INDEPENDENTLY, as one item of a batch (the separator rows ride along as border rows of every panel
// and end up holding Y_c = B_c L_c^-T, the border block holds -Y_c Y_c^T); the small separator system
// C - sum_c Y_c Y_c^T is solved next, and the chunks are back-substituted as a batch again.  Exact arithmetic is the
// same as one Cholesky of a re-ordered matrix; the dependency chain is n / P + sepw (P - 1) pivots instead of n.
// No atomics and a fixed summation order: results are bit-reproducible.

// Level set-up in one launch.  blockIdx.z < P: chunk z's matrix, GATHER_ROWS local rows per workgroup (one workgroup
// per row made the launch dispatch-bound: 17 k workgroups at the first level of C3), then the workgroups that transpose
// the coupling block with the separator above.  blockIdx.z == P: the rows of the separator system Cs (block diagonal part
// from S, ws from rhs); only its block tridiagonal band is ever written by the factorisation (and read by a child
// plan's gather), everything outside it stays the zero it was allocated with.
#define GATHER_ROWS 8
__global__ __launch_bounds__(256) void k_level_gather(const double* __restrict__ S, int64_t ld, const double* __restrict__ rhs,
                                                      const int64_t* __restrict__ env_col, const CholBatch B,
                                                      const CholStep first, int64_t sepw, int P,
                                                      const int64_t* __restrict__ sep_start, double* __restrict__ Cs,
                                                      int64_t lds, double* __restrict__ ws, int64_t tile0)
{
    const int z = blockIdx.z;
    if (z == P) {
        for (int rr = 0; rr < GATHER_ROWS; ++rr) {
            const int64_t i = (int64_t)blockIdx.x * GATHER_ROWS + rr;
            if (i >= lds) return;
            const int64_t c = i / sepw, u = i - c * sepw;
            const int64_t g = sep_start[c] + u;
            double* dst = Cs + i * lds;
            const int64_t j0 = c > 0 ? (c - 1) * sepw : 0, j1 = (c + 1) * sepw; // lower part of the band
            for (int64_t j = j0 + threadIdx.x; j < j1; j += 256) {
                int64_t cj = j / sepw, v = j - cj * sepw;
                dst[j] = (cj == c && v <= u) ? S[g * ld + sep_start[c] + v] : 0.0;
            }
            if (threadIdx.x == 0) ws[i] = rhs[g];
        }
        return;
    }
    const int64_t ldc = B.it[z].ld, nc = B.it[z].ncols, a = first.v[z];
    const bool has_top = z > 0, has_bot = z < P - 1;
    if ((int64_t)blockIdx.x >= tile0) {
        // 64 x 64 tile (tr, tc) of the coupling block S[a .. a + jn, a - sepw .. a) with the separator above: read by rows,
        // written transposed into the border rows nc + 64 tc .. of the chunk matrix
        __shared__ double sTile[64][65];
        const int64_t jn = nc < sepw ? nc : sepw, nt_r = jn / 64, tt = (int64_t)blockIdx.x - tile0;
        if (!has_top || tt >= nt_r * (sepw / 64)) return;
        const int64_t tr = tt % nt_r, tc = tt / nt_r;
        const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 16;
        const double* src = S + (a + 64 * tr + r) * ld + (a - sepw + 64 * tc) + c0;
#pragma unroll
        for (int c = 0; c < 16; ++c) sTile[r][c0 + c] = src[c];
        __syncthreads();
        double* out = B.it[z].A + (nc + 64 * tc + r) * ldc + 64 * tr + c0;
#pragma unroll
        for (int c = 0; c < 16; ++c) out[c] = sTile[c0 + c][r];
        return;
    }
    double* __restrict__ wc = B.it[z].w;
    {
        // eight interior rows of one 128-row tile share their skyline segment [c0, c1): all their loads first, then the stores
        // (one row after the other -- a load, its store, the next load -- the top level's 67 MB took 34 us, 2 TB/s)
        static_assert(128 % GATHER_ROWS == 0, "the rows of a workgroup lie in one 128-row tile");
        const int64_t i0 = (int64_t)blockIdx.x * GATHER_ROWS;
        if (i0 + GATHER_ROWS <= nc) {
            int64_t c0 = env_col[(a + i0) / 128] - a;
            if (c0 < 0) c0 = 0;
            const int64_t c1 = 128 * (i0 / 128 + 1);
            for (int64_t j = c0 / 2 + threadIdx.x; j < c1 / 2; j += 256) {
                double2 v[GATHER_ROWS];
#pragma unroll
                for (int rr = 0; rr < GATHER_ROWS; ++rr) v[rr] = reinterpret_cast<const double2*>(S + (a + i0 + rr) * ld + a)[j];
#pragma unroll
                for (int rr = 0; rr < GATHER_ROWS; ++rr) reinterpret_cast<double2*>(B.it[z].A + (i0 + rr) * ldc)[j] = v[rr];
            }
            if (threadIdx.x < GATHER_ROWS) wc[i0 + threadIdx.x] = rhs[a + i0 + threadIdx.x];
            return;
        }
    }
    for (int rr = 0; rr < GATHER_ROWS; ++rr) {
        const int64_t i = (int64_t)blockIdx.x * GATHER_ROWS + rr;
        if (i >= ldc) return;
        double* dst = B.it[z].A + i * ldc;
        if (i < nc) {
            const int64_t g = a + i;
            int64_t c0 = env_col[g / 128] - a;
            if (c0 < 0) c0 = 0;
            const int64_t c1 = 128 * (i / 128 + 1); // end of this row's skyline segment (tile aligned, <= nc)
            const double2* src = reinterpret_cast<const double2*>(S + g * ld + a); // c0, c1: multiples of 128
            double2* d2 = reinterpret_cast<double2*>(dst);
            for (int64_t j = c0 / 2 + threadIdx.x; j < c1 / 2; j += 256) d2[j] = src[j];
            if (threadIdx.x == 0) wc[i] = rhs[g];
            continue;
        }
        // border row: [0, sepw) separator above, [sepw, 2 sepw) separator below.  Columns of the border block right of the
        // row's own 128-tile are never touched (the trailing updates work on the lower tile triangle).
        const int64_t u2 = i - nc;
        const int64_t jend = nc + 128 * (u2 / 128 + 1);
        if (u2 < sepw) {
            // the coupling with the separator above lives in S[a + j][a - sepw + u2], j < sepw (the separator is at least
            // one bandwidth wide): a TRANSPOSED block, copied by the tile workgroups; this row only clears the rest
            const int64_t jn = has_top ? (nc < sepw ? nc : sepw) : 0; // jn, jend, ldc: multiples of 64 -> 16-byte stores
            double2* d2 = reinterpret_cast<double2*>(dst);
            for (int64_t j = jn / 2 + threadIdx.x; j < jend / 2; j += 256) d2[j] = make_double2(0.0, 0.0);
        } else {
            const int64_t g = a + nc + (u2 - sepw);
            const int64_t j0 = nc > 2 * sepw ? nc - 2 * sepw : 0;
            const double2* s2 = reinterpret_cast<const double2*>(S + g * ld + a); // a, j0, nc: even
            double2* d2 = reinterpret_cast<double2*>(dst);
            for (int64_t j = threadIdx.x; j < jend / 2; j += 256)
                d2[j] = (has_bot && 2 * j >= j0 && 2 * j < nc) ? s2[j] : make_double2(0.0, 0.0);
        }
        if (threadIdx.x == 0) wc[i] = 0.0;
    }
}

// Add the chunks' border blocks (-Y Y^T) and border right-hand sides into the separator system: one workgroup per
// separator row.  Separator c lies between chunks c and c + 1: its diagonal block takes chunk c's (below, below)
// block and chunk c + 1's (above, above) block, its coupling to separator c - 1 is chunk c's (below, above) block.
// Fixed summation order, no atomics.
__global__ __launch_bounds__(256) void k_sep_reduce(const CholBatch B, int64_t sepw, double* __restrict__ Cs, int64_t lds,
                                                    double* __restrict__ ws)
{
    const int64_t ru = blockIdx.x;
    const int c = (int)(ru / sepw);
    const int64_t u = ru - (int64_t)c * sepw;
    const int64_t n0 = B.it[c].ncols, l0 = B.it[c].ld, n1 = B.it[c + 1].ncols, l1 = B.it[c + 1].ld;
    const double* below = B.it[c].A + (n0 + sepw + u) * l0 + n0;   // chunk c, border row sepw + u: [above | below]
    const double* above = B.it[c + 1].A + (n1 + u) * l1 + n1;       // chunk c + 1, border row u: [above]
    double* row = Cs + ru * lds;
    for (int64_t v = threadIdx.x; v <= u; v += 256) row[c * sepw + v] = (row[c * sepw + v] + below[sepw + v]) + above[v];
    if (c > 0)
        for (int64_t v = threadIdx.x; v < sepw; v += 256) row[(c - 1) * sepw + v] += below[v];
    if (threadIdx.x == 0) ws[ru] = (ws[ru] + B.it[c].w[n0 + sepw + u]) + B.it[c + 1].w[n1 + u];
}

// border part of a chunk's solution = the separator solution; fold it into y: y_j -= sum_i L[nc + i][j] x[nc + i].
// A workgroup takes 64 columns; its four waves split the 2 sepw border rows and combine through LDS.
// The first column workgroup of chunk c also places the solution of the separator below it into the level's solution
// vector x (and checks it where x is the final solution): with the chunk interiors written there directly by k_bwd256
// (their CholItem::x points into x), no scatter launch is left.
__global__ __launch_bounds__(256) void k_bwd_border(const CholBatch B, int P, int64_t sepw, const double* __restrict__ xs,
                                                    const int64_t* __restrict__ sep_start, double* __restrict__ x,
                                                    int* __restrict__ finite)
{
    __shared__ double sx[2 * SRK_MAX_SEPW];
    __shared__ double sp[4][64];
    const int c = blockIdx.z;
    const int64_t nc = B.it[c].ncols, ldc = B.it[c].ld;
    if ((int64_t)blockIdx.x * 64 >= nc) return;
    const int64_t top_sep = c > 0 ? c - 1 : -1, bot_sep = c < P - 1 ? c : -1;
    if (blockIdx.x == 0 && bot_sep >= 0)
        for (int64_t u = threadIdx.x; u < sepw; u += 256) {
            const double v = xs[bot_sep * sepw + u];
            x[sep_start[bot_sep] + u] = v;
            if (finite && !isfinite(v)) atomicOr(finite, 4);
        }
    for (int64_t u2 = threadIdx.x; u2 < 2 * sepw; u2 += 256) {
        const int64_t su = u2 < sepw ? top_sep : bot_sep;
        sx[u2] = su < 0 ? 0.0 : xs[su * sepw + (u2 < sepw ? u2 : u2 - sepw)];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * 64 + lane; // nc is a multiple of 64
    const int64_t rows = (2 * sepw) / 4, u0 = wave * rows;
    const double* col = B.it[c].A + (nc + u0) * ldc + j;
    double acc = 0;
#pragma unroll 8
    for (int64_t u2 = 0; u2 < rows; ++u2) acc = fma(col[u2 * ldc], sx[u0 + u2], acc);
    sp[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) B.it[c].y[j] -= (sp[0][lane] + sp[1][lane]) + (sp[2][lane] + sp[3][lane]);
}

static void solve_chunked(hipStream_t s, const SrkChunkPlan& pl, int64_t ld, const double* S, const double* rhs, double* x,
                          const int64_t* d_env_col, int* d_info, SrkSolveProf* prof, bool top, SrkCholSync* sync)
{
    const int P = pl.P;
    const int64_t sepw = pl.sepw, lds = pl.lds;
    CholBatch B{}, Bs{};
    CholStep first;
    CholHostItem H[SRK_MAX_CHUNKS], Hs;
    Hs.row_end = pl.s_row_end.data();
    Hs.col_begin = pl.s_col_begin.data();
    int64_t max_ldc = 0, max_nc = 0;
    for (int c = 0; c < SRK_MAX_CHUNKS; ++c) first.v[c] = 0;
    for (int c = 0; c < P; ++c) {
        // a chunk's solution goes straight to its place in the level's solution vector
        B.it[c] = CholItem{ pl.Ac[c], pl.wc[c], pl.yc[c], x + pl.a[c], pl.dinvc[c], pl.ldc[c], pl.n[c], pl.n[c], pl.ldc[c] };
        H[c].row_end = pl.row_end[c].data();
        H[c].col_begin = pl.col_begin[c].data();
        H[c].r2_split = pl.n[c] + sepw;
        H[c].bot_first_col = pl.n[c] - sepw; // the separator below couples with the last sepw (>= bandwidth) columns only
        H[c].has_top = c > 0;
        H[c].has_bot = c < P - 1;
        first.v[c] = pl.a[c];
        max_ldc = std::max(max_ldc, pl.ldc[c]);
        max_nc = std::max(max_nc, pl.n[c]);
    }
    Bs.it[0] = CholItem{ pl.Cs, pl.ws, pl.ys, pl.xs, pl.dinvs, lds, lds, lds, lds };

    // row workgroups (GATHER_ROWS rows each), then the (sepw / 64)^2 transposing tile workgroups of an item
    const int64_t tile0 = (std::max(max_ldc, lds) + GATHER_ROWS - 1) / GATHER_ROWS;
    LAUNCH(k_level_gather, dim3((unsigned)(tile0 + (sepw / 64) * (sepw / 64)), 1, (unsigned)(P + 1)), dim3(256), 0, s, S, ld,
           rhs, d_env_col, B, first, sepw, P, pl.d_sep_start, pl.Cs, lds, pl.ws, tile0);
    chol_factor(s, B, P, H, d_info, prof, sync);
    LAUNCH(k_sep_reduce, dim3((unsigned)lds), dim3(256), 0, s, B, sepw, pl.Cs, lds, pl.ws);
    if (pl.child) { // the separator system is block tridiagonal: chunk it again
        solve_chunked(s, *pl.child, lds, pl.Cs, pl.ws, pl.xs, pl.d_sep_env, d_info, prof, false, sync);
    } else {
        chol_factor(s, Bs, 1, &Hs, d_info, prof, sync);
        chol_bwd(s, Bs, 1, &Hs, prof, nullptr);
    }
    int* d_finite = top ? d_info : nullptr;
    LAUNCH(k_bwd_border, dim3((unsigned)(max_nc / 64), 1, (unsigned)P), dim3(256), 0, s, B, P, sepw, pl.xs, pl.d_sep_start, x,
           d_finite);
    chol_bwd(s, B, P, H, prof, d_finite);
}

void srk_chol_solve_chunked(hipStream_t s, const SrkChunkPlan& pl, int64_t ld, const double* S, const double* rhs,
                            double* x, const int64_t* d_env_col, int* d_info, SrkSolveProf* prof, SrkCholSync* sync)
{
    // every variable of the system is a chunk or a separator variable of the top level: its backward kernels check them all
    solve_chunked(s, pl, ld, S, rhs, x, d_env_col, d_info, prof, true, sync);
}
