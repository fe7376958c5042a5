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
//   inner panel  = 64 columns (4 per outer panel): k_panel factorises the 64x64 diagonal tile (every workgroup
//                  redundantly, in registers + LDS), solves its rows of the panel by substitution and folds the
//                  forward substitution of the right-hand side in; k_upd64 applies the 64-deep update to the
//                  remaining columns of the outer panel (MFMA as well).
// The per-outer-panel row limit `row_end` lets the caller skip the structurally zero part of a banded / skyline
// system (the envelope of a Cholesky factor equals the envelope of the matrix); dense = ld for every panel.
#include "srk_dev.hpp"

#define NB 64
#define NBO SRK_CHOL_NB      // 256, outer panel
#define TL 128               // trailing-update tile
#define KC 16                // trailing-update K chunk
#define KCP (KC + 2)         // padded LDS row (18 doubles: conflict-free ds_read_b64 for the MFMA operand map)
#define LDSP (NB + 2)

typedef double double4_t __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ double fast_rsqrt(double d)
{
    double r = __builtin_amdgcn_rsq(d);
    // Newton for 1/sqrt: r <- r * (1.5 - 0.5 d r^2), twice
    r = r * fma(-0.5 * d * r, r, 1.5);
    r = r * fma(-0.5 * d * r, r, 1.5);
    return r;
}

// Branch-free inner loop: entries above the diagonal are updated too (never read), so the only per-element
// predicate left is "column > pivot column" inside the pivot's own group of four.
__device__ __forceinline__ bool potrf64(double (*sD)[NB + 1], double (*sT)[NB + 2], double* sCol /*[2][64]*/,
                                        double* sDiag /*[64]*/, double* sInv /*[64]*/)
{
    const int t = threadIdx.x, i = t >> 2, q = t & 3;
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = sD[i][4 * m + q];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double* col = sCol + (j & 1) * NB;
        if (q == (j & 3)) col[i] = a[j >> 2];
        __syncthreads();
        double dj = col[j];
        bad |= !(dj > 0.0 && dj < 1.0e300);
        double li = col[i] * fast_rcp(dj);
        {
            const int m0 = j >> 2;
            double f = (q > (j & 3)) ? li : 0.0;
            a[m0] = fma(-f, col[4 * m0 + q], a[m0]);
        }
#pragma unroll
        for (int m = (j >> 2) + 1; m < 16; ++m) a[m] = fma(-li, col[4 * m + q], a[m]);
    }
    // the owner of a diagonal entry still holds its pivot d_i (entries are final once their column is passed)
#pragma unroll
    for (int m = 0; m < 16; ++m)
        if (4 * m + q == i) sDiag[i] = a[m];
    __syncthreads();
    if (t < NB) sInv[t] = fast_rsqrt(sDiag[t]); // 1 / L_tt
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        int c = 4 * m + q;
        double v = a[m] * sInv[c];          // c < i : a_ic / sqrt(d_c) ; c == i : d_i / sqrt(d_i) = L_ii
        v = (c <= i) ? v : 0.0;
        sD[i][c] = v;
        sT[c][i] = v; // transposed copy: column j of L contiguous in t for the row sweeps
    }
    __syncthreads();
    return bad != 0;
}

// ---------------------------------------------------------------- inner panel: potrf + trsm + forward substitution
// d = index of the 64-wide diagonal tile.  Rows (d+1)*64 .. row_end-1 of columns [64 d, 64 d + 64) become L.
// w is the running right-hand side: y_d = L_dd^-1 w_d is published to y, and w_r -= L[r, d-cols] . y_d for the
// rows below (so the forward substitution L y = b costs no extra launches).
__global__ __launch_bounds__(256) void k_panel(double* __restrict__ A, int64_t ld, int64_t d, int64_t row_end,
                                               double* __restrict__ w, double* __restrict__ y,
                                               int* __restrict__ info)
{
    __shared__ double sD[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double sT[NB][NB + 2];
    __shared__ double sCol[2 * NB];
    __shared__ double sDiag[NB];
    __shared__ double sInv[NB];
    __shared__ double sy[NB];
    const int64_t k0 = d * NB;
    double* Ab = A + k0 * ld + k0;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 6, c = e & 63;
        sD[i][c] = (c <= i) ? Ab[(int64_t)i * ld + c] : 0.0;
    }
    if (threadIdx.x < NB) sy[threadIdx.x] = w[k0 + threadIdx.x];
    __syncthreads();
    bool bad = potrf64(sD, sT, sCol, sDiag, sInv);
    if (bad && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(info, 1);
    if (blockIdx.x == 0) {
        for (int e = threadIdx.x; e < NB * NB; e += 256) {
            int i = e >> 6, c = e & 63;
            if (c <= i) Ab[(int64_t)i * ld + c] = sD[i][c];
        }
    }
    // y_d = L_dd^-1 w_d : one wave, lock-step column sweep
    if (threadIdx.x < NB) {
        int i = threadIdx.x;
        double bi = sy[i];
#pragma unroll 8
        for (int j = 0; j < NB; ++j) {
            double yj = __shfl(bi, j, 64) * sInv[j];
            double lij = (i > j) ? sD[i][j] : 0.0;
            bi = (i == j) ? yj : fma(-lij, yj, bi);
        }
        sy[i] = bi;
        if (blockIdx.x == 0) y[k0 + i] = bi;
    }
    __syncthreads();
    // rows of the panel: 64 rows per workgroup, a row is split over the 4 lanes of a quad (lane q owns the columns
    // c = 4m + q, 16 registers) -- 4x the parallelism of a row-per-thread sweep and the same code shape as potrf64
    const int q = threadIdx.x & 3;
    int64_t r = k0 + NB + (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    bool live = r < row_end;
    double* row = A + (live ? r : k0) * ld + k0; // dead rows read the diagonal tile (harmless) and store nothing
    double a[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) a[m] = row[4 * m + q];
    double dot = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        // x_j = a_j / L_jj lives in lane (j & 3) of the quad: broadcast it inside the quad
        double xo = a[j >> 2] * sInv[j];
        double x = __shfl(xo, (threadIdx.x & 60) | (j & 3), 64);
        if (q == (j & 3)) { a[j >> 2] = x; dot = fma(x, sy[j], dot); }
        {
            const int m0 = j >> 2;
            double f = (q > (j & 3)) ? x : 0.0;
            a[m0] = fma(-f, sT[j][4 * m0 + q], a[m0]);
        }
#pragma unroll
        for (int m = (j >> 2) + 1; m < 16; ++m) a[m] = fma(-x, sT[j][4 * m + q], a[m]);
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    if (live) {
#pragma unroll
        for (int m = 0; m < 16; ++m) row[4 * m + q] = a[m];
        if (q == 0) w[r] -= dot;
    }
}

// ---------------------------------------------------------------- 64-deep update inside the outer panel (MFMA)
// A[rt, ct] -= L[rt, d] L[ct, d]^T for row tiles rt > d (rows < row_end) and column tiles d < ct <= c_hi, ct <= rt.
// grid = (row tiles, column tiles).  4 waves, each a 32x32 quadrant = 2x2 accumulator tiles, K = 64.
__global__ __launch_bounds__(256) void k_upd64(double* __restrict__ A, int64_t ld, int64_t d, int64_t c_hi)
{
    __shared__ double sA[NB][LDSP];
    __shared__ double sB[NB][LDSP];
    int64_t rt = d + 1 + blockIdx.x;
    int64_t ct = d + 1 + blockIdx.y;
    if (ct > c_hi || ct > rt) return;
    int64_t k0 = d * NB, r0 = rt * NB, c0 = ct * NB;
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
    __syncthreads();
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int wr = wave >> 1, wc = wave & 1;
    int lr = lane & 15, lk = lane >> 4;
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
    // f64 16x16x4 accumulator map: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                int64_t r = r0 + wr * 32 + m * 16 + lk + 4 * reg;
                int64_t c = c0 + wc * 32 + n * 16 + lr;
                double* pc = A + r * ld + c;
                *pc = *pc - acc[m][n][reg];
            }
}

// ---------------------------------------------------------------- trailing update of an outer panel (MFMA, K = 256)
// C[ti, tj] -= P[ti] P[tj]^T over the 128x128 tile pairs ti >= tj of rows/cols [c_first, row_end), P = the 256
// panel columns starting at k0.  One workgroup per tile pair (linear index -> triangular pair).
__global__ __launch_bounds__(256, 2) void k_trail(double* __restrict__ A, int64_t ld, int64_t k0, int64_t c_first)
{
    __shared__ double sA[2][TL][KCP];
    __shared__ double sB[2][TL][KCP];
    int64_t p = blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= p) ++ti;
    while ((int64_t)ti * (ti + 1) / 2 > p) --ti;
    int tj = (int)(p - (int64_t)ti * (ti + 1) / 2);
    const int64_t r0 = c_first + (int64_t)ti * TL, c0 = c_first + (int64_t)tj * TL;
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

// ---------------------------------------------------------------- backward substitution L^T x = y
// step d (descending): x_d = L_dd^-T y_d (every workgroup, redundantly), then y_j -= sum_i L[64 d + i, j] x_d[i]
// for the columns j in [col_begin, 64 d) (col_begin > 0 only for banded / skyline systems).
__global__ __launch_bounds__(256) void k_bwd_step(const double* __restrict__ A, int64_t ld, int64_t d,
                                                  int64_t col_begin, double* __restrict__ y, double* __restrict__ x)
{
    __shared__ double sL[NB][NB + 1];
    __shared__ double sx[NB];
    __shared__ double sInv[NB];
    const int64_t k0 = d * NB;
    const double* Lb = A + k0 * ld + k0;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 6, c = e & 63;
        sL[i][c] = (c <= i) ? Lb[(int64_t)i * ld + c] : 0.0;
    }
    if (threadIdx.x < NB) sx[threadIdx.x] = y[k0 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < NB) sInv[threadIdx.x] = 1.0 / sL[threadIdx.x][threadIdx.x];
    __syncthreads();
    if (threadIdx.x < NB) {
        int i = threadIdx.x;
        double bi = sx[i];
#pragma unroll 8
        for (int j = NB - 1; j >= 0; --j) {
            double xj = __shfl(bi, j, 64) * sInv[j];
            double lji = (i < j) ? sL[j][i] : 0.0; // (L^T)[i][j] = L[j][i]
            bi = (i == j) ? xj : fma(-lji, xj, bi);
        }
        sx[i] = bi;
        if (blockIdx.x == 0) x[k0 + i] = bi;
    }
    __syncthreads();
    int64_t j = col_begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= k0) return;
    double s = 0;
#pragma unroll 8
    for (int i = 0; i < NB; ++i) s += A[(k0 + i) * ld + j] * sx[i];
    y[j] -= s;
}

// the solution must be all finite (the reference's allFinite check, :1912-1913)
__global__ __launch_bounds__(256) void k_check_finite(int64_t n, const double* __restrict__ x, int* __restrict__ info)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && !isfinite(x[i])) atomicOr(info, 4);
}

// w: right-hand side (destroyed).  y: scratch (forward solution).  x: solution.
// row_end[K] (host, one per outer panel, multiple of 128, > 256 (K+1) or == ld): rows >= row_end[K] have no
// non-zero in the panel's columns and are skipped; NULL = dense.  col_begin[d64] (host, per 64-tile, may be NULL):
// first column with a non-zero in tile row d64.  ev_pairs: optional 2 events per outer panel around k_trail.
void srk_chol_solve(hipStream_t s, int64_t ld, double* A, double* w, double* y, double* x, int* d_info,
                    const int64_t* row_end, const int64_t* col_begin, hipEvent_t* ev_pairs)
{
    const int64_t nout = ld / NBO;
    for (int64_t K = 0; K < nout; ++K) {
        const int64_t k0 = K * NBO;
        int64_t rend = row_end ? row_end[K] : ld;
        if (rend < k0 + NBO) rend = k0 + NBO;
        if (rend > ld) rend = ld;
        for (int jsub = 0; jsub < NBO / NB; ++jsub) {
            int64_t d = K * (NBO / NB) + jsub;
            int64_t rows = rend - (d + 1) * NB;
            int64_t blocks = rows > 0 ? (rows + 63) / 64 : 1;
            hipLaunchKernelGGL(k_panel, dim3((unsigned)blocks), dim3(256), 0, s, A, ld, d, rend, w, y, d_info);
            int64_t c_hi = K * (NBO / NB) + (NBO / NB - 1);
            if (jsub < NBO / NB - 1 && rows > 0) {
                int64_t rtiles = rows / NB;
                int64_t ctiles = c_hi - d;
                hipLaunchKernelGGL(k_upd64, dim3((unsigned)rtiles, (unsigned)ctiles), dim3(256), 0, s, A, ld, d, c_hi);
            }
        }
        int64_t c_first = k0 + NBO;
        int64_t T = (rend - c_first) / TL;
        if (ev_pairs) hipEventRecord(ev_pairs[2 * K], s);
        if (T > 0) {
            int64_t pairs = T * (T + 1) / 2;
            hipLaunchKernelGGL(k_trail, dim3((unsigned)pairs), dim3(256), 0, s, A, ld, k0, c_first);
        }
        if (ev_pairs) hipEventRecord(ev_pairs[2 * K + 1], s);
    }
    const int64_t n64 = ld / NB;
    for (int64_t d = n64 - 1; d >= 0; --d) {
        int64_t cb = col_begin ? col_begin[d] : 0;
        if (cb > d * NB) cb = d * NB;
        int64_t cols = d * NB - cb;
        int64_t blocks = cols > 0 ? (cols + 255) / 256 : 1;
        hipLaunchKernelGGL(k_bwd_step, dim3((unsigned)blocks), dim3(256), 0, s, A, ld, d, cb, y, x);
    }
    hipLaunchKernelGGL(k_check_finite, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, ld, x, d_info);
}
