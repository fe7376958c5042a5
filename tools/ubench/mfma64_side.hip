// How fast does a fourth wave of a SIMD run while the other three issue v_mfma_f64_16x16x4 back to back (no other
// instruction in their loop)?  Waves 12 .. 15 of a 1024-thread workgroup run PATTERN for `n` iterations and report their
// own clock64 per iteration, once with the multiplying waves idle and once with them busy (PRIO: s_setprio 2 in the side wave).
//   0: 16 dependent v_add_u32          1: 16 independent v_add_u32        2: ds_read_b64 -> wait -> ds_write_b64 (4x)
// FIRST = 1: the side waves are waves 0 .. 3 of the workgroup (the oldest waves of their SIMDs) instead of 12 .. 15.
//   3: 8 scalar branches (taken)       4: 3 dependent MFMAs (side wave)   5: 16 v_fma_f64 dependent
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int PATTERN, int PRIO, int FIRST>
__global__ __launch_bounds__(1024) void k(double* out, int mf_iters, int side_iters, long long* cyc)
{
    __shared__ double sW[4096];
    __shared__ int stop;
    for (int t = threadIdx.x; t < 4096; t += blockDim.x) sW[t] = 1e-3 * t;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    const int wv0 = threadIdx.x >> 6;
    const int wv = FIRST ? (wv0 + 12) & 15 : wv0; // FIRST: the side waves are waves 0 .. 3, the oldest of their SIMDs
    double s = 0;
    if (wv < 12) {
        double4_t acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (double4_t){ 0, 0, 0, 0 };
        double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
        for (int it = 0; it < mf_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        int x = threadIdx.x, y = 3;
        double f = 1.0, g = 1.000001;
        const int addr = (threadIdx.x & 63) * 8;
        double4_t m = (double4_t){ 0, 0, 0, 0 };
        long long t0 = clock64();
        for (int it = 0; it < side_iters; ++it) {
            if (PATTERN == 0) {
#pragma unroll
                for (int v = 0; v < 16; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
            } else if (PATTERN == 1) {
                int z[4] = { x, x + 1, x + 2, x + 3 };
#pragma unroll
                for (int v = 0; v < 16; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(z[v & 3]) : "v"(y));
                x = z[0] + z[1] + z[2] + z[3];
            } else if (PATTERN == 2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double l;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(l) : "v"(addr) : "memory");
                    asm volatile("ds_write_b64 %0, %1 offset:8192\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(l) : "memory");
                    f += l;
                }
            } else if (PATTERN == 3) {
#pragma unroll
                for (int v = 0; v < 8; ++v) asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" ::: "scc");
            } else if (PATTERN == 4) {
                m = __builtin_amdgcn_mfma_f64_16x16x4f64(f, g, m, 0, 0, 0);
                m = __builtin_amdgcn_mfma_f64_16x16x4f64(f, g, m, 0, 0, 0);
                m = __builtin_amdgcn_mfma_f64_16x16x4f64(f, g, m, 0, 0, 0);
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(f) : "v"(g));
            }
        }
        long long t1 = clock64();
        s = x + f + m[0];
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wv - 12] = t1 - t0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int PATTERN, int PRIO, int FIRST = 0>
void run(const char* what)
{
    const int side = 2000, threads = 1024, blocks = 256;
    double* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * threads * blocks);
    (void)hipMalloc(&cyc, 32);
    double res[2];
    for (int busy = 0; busy < 2; ++busy) {
        // the multiplying waves must outlast the side waves: 24 MFMAs x 64 cycles per SIMD and iteration
        hipLaunchKernelGGL((k<PATTERN, PRIO, FIRST>), dim3(blocks), dim3(threads), 0, 0, out, busy ? 40000 : 0, side, cyc);
        (void)hipDeviceSynchronize();
        long long c[4]; (void)hipMemcpy(c, cyc, 32, hipMemcpyDeviceToHost);
        res[busy] = (double)(c[0] + c[1] + c[2] + c[3]) / 4 / side;
    }
    printf("%-52s prio %d, side waves %s: %8.1f cycles per iteration alone, %8.1f beside three MFMA waves\n", what, PRIO, FIRST ? "first" : "last ", res[0], res[1]);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
    run<0, 0>("16 dependent v_add_u32"); run<0, 1>("16 dependent v_add_u32");
    run<1, 0>("16 v_add_u32, four chains"); run<1, 1>("16 v_add_u32, four chains");
    run<5, 0>("16 dependent v_fma_f64"); run<5, 1>("16 dependent v_fma_f64");
    run<2, 0>("4 x (ds_read, wait, ds_write, wait)"); run<2, 1>("4 x (ds_read, wait, ds_write, wait)");
    run<3, 0>("8 taken scalar branches"); run<3, 1>("8 taken scalar branches");
    run<4, 0>("3 dependent MFMAs in the side wave"); run<4, 1>("3 dependent MFMAs in the side wave");
    run<0, 0, 1>("16 dependent v_add_u32"); run<0, 1, 1>("16 dependent v_add_u32");
    run<5, 0, 1>("16 dependent v_fma_f64"); run<2, 0, 1>("4 x (ds_read, wait, ds_write, wait)"); run<2, 1, 1>("4 x (ds_read, wait, ds_write, wait)");
    run<4, 0, 1>("3 dependent MFMAs in the side wave");
    return 0;
}
