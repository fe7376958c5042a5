// srk_chol.hip -- dense SPD solve of the reduced camera system on gfx950 (fp64).
//
// Replaces `decomp_lin_sys_left_side.householderQr().solve(rhs)` (bundle-adj-kanatani.cpp:1911).  The reduced
// camera system is symmetric positive definite under the multiplicative LM damping, so a blocked right-looking
// Cholesky is used; its trailing update (the only dense contraction on the path, n^3/3 flops) runs on the fp64
// matrix cores: v_mfma_f64_16x16x4_f64, one 16x16 accumulator tile per MFMA, operands staged through LDS.
// A non-positive or non-finite pivot sets *info (the caller maps it to the reference's "solve failed" path).
//
// Layout: A row-major ld x ld, LOWER triangle authoritative and overwritten by L.  ld % 64 == 0.
#include "srk_dev.hpp"

#define NB SRK_CHOL_NB
#define LDSP (NB + 2) // 66 doubles: rows land on distinct LDS bank groups for the MFMA operand reads

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- diagonal block factorisation
// One workgroup.  Outer-product form with deferred scaling: at step j the pivot d_j = a_jj is final, the trailing
// entries get a_ic -= a_ij a_cj / d_j, and columns are scaled by 1/sqrt(d_j) at the end (one barrier per step).
__global__ __launch_bounds__(256) void k_potrf_diag(double* __restrict__ A, int64_t ld, int64_t k0,
                                                    int* __restrict__ info)
{
    __shared__ double sA[NB][NB + 1];
    __shared__ double sD[NB];
    double* Ab = A + k0 * ld + k0;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 6, c = e & 63;
        sA[i][c] = (c <= i) ? Ab[(int64_t)i * ld + c] : 0.0;
    }
    __syncthreads();
    bool bad = false;
    for (int j = 0; j < NB; ++j) {
        double dj = sA[j][j];
        if (!(dj > 0.0) || !isfinite(dj)) bad = true;
        double inv = 1.0 / dj;
        for (int e = threadIdx.x; e < NB * NB; e += 256) {
            int i = e >> 6, c = e & 63;
            if (c > j && c <= i) sA[i][c] -= sA[i][j] * sA[c][j] * inv;
        }
        if (threadIdx.x == 0) sD[j] = dj;
        __syncthreads();
    }
    if (bad && threadIdx.x == 0) atomicOr(info, 1);
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 6, c = e & 63;
        if (c <= i) {
            double s = sqrt(sD[c]);
            Ab[(int64_t)i * ld + c] = (c == i) ? s : sA[i][c] / s;
        }
    }
}

// ---------------------------------------------------------------- panel triangular solve
// X = A[i, k0:k0+NB] * L_kk^-T for the rows below the diagonal block; one thread per row, the row lives in
// registers, L_kk is broadcast from LDS.
__global__ __launch_bounds__(256) void k_trsm_panel(double* __restrict__ A, int64_t ld, int64_t k0, int64_t n)
{
    __shared__ double sL[NB][NB + 1];
    __shared__ double sInv[NB];
    const double* Lb = A + k0 * ld + k0;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 6, c = e & 63;
        sL[i][c] = (c <= i) ? Lb[(int64_t)i * ld + c] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < NB) sInv[threadIdx.x] = 1.0 / sL[threadIdx.x][threadIdx.x];
    __syncthreads();
    int64_t i = k0 + NB + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double* row = A + i * ld + k0;
    double a[NB];
#pragma unroll
    for (int t = 0; t < NB; t += 2) {
        double2 v = *reinterpret_cast<const double2*>(row + t);
        a[t] = v.x;
        a[t + 1] = v.y;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double x = a[j] * sInv[j];
        a[j] = x;
#pragma unroll
        for (int t = j + 1; t < NB; ++t) a[t] -= x * sL[t][j];
    }
#pragma unroll
    for (int t = 0; t < NB; t += 2) *reinterpret_cast<double2*>(row + t) = make_double2(a[t], a[t + 1]);
}

// ---------------------------------------------------------------- trailing update on the fp64 matrix cores
// C[ti,tj] -= L[ti,k] L[tj,k]^T for the lower tile pairs (ti >= tj) of the trailing matrix.  256 threads = 4 waves,
// each wave owns a 32x32 quadrant = 2x2 MFMA tiles of 16x16; K = NB = 64 -> 16 v_mfma_f64_16x16x4_f64 per tile.
__global__ __launch_bounds__(256) void k_syrk_mfma(double* __restrict__ A, int64_t ld, int64_t k0, int ntiles)
{
    __shared__ double sA[NB][LDSP];
    __shared__ double sB[NB][LDSP];
    // decode the lower-triangular tile pair from the linear block index
    int64_t p = blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= p) ++ti;
    while ((int64_t)ti * (ti + 1) / 2 > p) --ti;
    int tj = (int)(p - (int64_t)ti * (ti + 1) / 2);
    (void)ntiles;
    int64_t r0 = k0 + NB + (int64_t)ti * NB;
    int64_t c0 = k0 + NB + (int64_t)tj * NB;
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

// ---------------------------------------------------------------- triangular solves with the factor
// forward step k: every workgroup solves L_kk y_k = b_k in LDS (redundantly; workgroup 0 publishes y_k), then
// updates its rows below: b_i -= L[i, k-block] y_k.
__global__ __launch_bounds__(256) void k_fwd_step(const double* __restrict__ A, int64_t ld, int64_t k0, int64_t n,
                                                  double* __restrict__ w, double* __restrict__ y)
{
    __shared__ double sL[NB][NB + 1];
    __shared__ double sy[NB];
    const double* Lb = A + k0 * ld + k0;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 6, c = e & 63;
        sL[i][c] = (c <= i) ? Lb[(int64_t)i * ld + c] : 0.0;
    }
    if (threadIdx.x < NB) sy[threadIdx.x] = w[k0 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < NB) { // one wave, lock-step column sweep
        int i = threadIdx.x;
        double bi = sy[i];
        for (int j = 0; j < NB; ++j) {
            double yj = __shfl(bi, j, 64) / sL[j][j];
            if (i == j) bi = yj;
            else if (i > j) bi -= sL[i][j] * yj;
        }
        sy[i] = bi;
        if (blockIdx.x == 0) y[k0 + i] = bi;
    }
    __syncthreads();
    int64_t i = k0 + NB + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* row = A + i * ld + k0;
    double s = 0;
#pragma unroll 8
    for (int t = 0; t < NB; t += 2) {
        double2 v = *reinterpret_cast<const double2*>(row + t);
        s += v.x * sy[t] + v.y * sy[t + 1];
    }
    w[i] -= s; // w (work rhs) is only read at [k0, k0+NB) by this launch and written below it: no race with y
}

// backward step k: solve L_kk^T x_k = y_k, then y_j -= sum_i L[k0+i, j] x_k[i] for the columns j < k0.
__global__ __launch_bounds__(256) void k_bwd_step(const double* __restrict__ A, int64_t ld, int64_t k0,
                                                  double* __restrict__ y, double* __restrict__ x)
{
    __shared__ double sL[NB][NB + 1];
    __shared__ double sx[NB];
    const double* Lb = A + k0 * ld + k0;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        int i = e >> 6, c = e & 63;
        sL[i][c] = (c <= i) ? Lb[(int64_t)i * ld + c] : 0.0;
    }
    if (threadIdx.x < NB) sx[threadIdx.x] = y[k0 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < NB) {
        int i = threadIdx.x;
        double bi = sx[i];
        for (int j = NB - 1; j >= 0; --j) {
            double xj = __shfl(bi, j, 64) / sL[j][j];
            if (i == j) bi = xj;
            else if (i < j) bi -= sL[j][i] * xj; // (L^T)[i][j] = L[j][i]
        }
        sx[i] = bi;
        if (blockIdx.x == 0) x[k0 + i] = bi;
    }
    __syncthreads();
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
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

// w: right-hand side, destroyed.  y: scratch.  x: solution.  ev_pairs: optional 2 * (ld / NB) events recorded around
// every trailing-update launch (no host synchronisation here; the caller reads them after its own sync).
void srk_chol_solve(hipStream_t s, int64_t ld, double* A, double* w, double* y, double* x, int* d_info,
                    hipEvent_t* ev_pairs)
{
    int64_t nblk = ld / NB;
    for (int64_t kb = 0; kb < nblk; ++kb) {
        int64_t k0 = kb * NB;
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), 0, s, A, ld, k0, d_info);
        int64_t rows = ld - k0 - NB;
        if (rows <= 0) break;
        hipLaunchKernelGGL(k_trsm_panel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, A, ld, k0, ld);
        int64_t T = rows / NB;
        int64_t pairs = T * (T + 1) / 2;
        if (ev_pairs) hipEventRecord(ev_pairs[2 * kb], s);
        hipLaunchKernelGGL(k_syrk_mfma, dim3((unsigned)pairs), dim3(256), 0, s, A, ld, k0, (int)T);
        if (ev_pairs) hipEventRecord(ev_pairs[2 * kb + 1], s);
    }
    for (int64_t kb = 0; kb < nblk; ++kb) {
        int64_t k0 = kb * NB;
        int64_t rows = ld - k0 - NB;
        int64_t blocks = rows > 0 ? (rows + 255) / 256 : 1;
        hipLaunchKernelGGL(k_fwd_step, dim3((unsigned)blocks), dim3(256), 0, s, A, ld, k0, ld, w, y);
    }
    for (int64_t kb = nblk - 1; kb >= 0; --kb) {
        int64_t k0 = kb * NB;
        int64_t blocks = k0 > 0 ? (k0 + 255) / 256 : 1;
        hipLaunchKernelGGL(k_bwd_step, dim3((unsigned)blocks), dim3(256), 0, s, A, ld, k0, y, x);
    }
    hipLaunchKernelGGL(k_check_finite, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, ld, x, d_info);
}
