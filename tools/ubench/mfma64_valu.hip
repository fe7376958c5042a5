// Does vector-ALU work steal issue slots from v_mfma_f64_16x16x4?  Three multiplying waves per SIMD issue MFMAs on register
// operands (8 accumulators); after every MFMA they issue NV vector instructions of kind KIND on unrelated registers:
//   0: v_add_u32 (32-bit integer)   1: v_cndmask pair (a double select)   2: v_fma_f64   3: ds_read_b64 (LDS, not VALU)
// MODE 1: the multipliers issue no extra instructions, a fourth wave per SIMD runs a VALU-only loop (helpers of k_schur_mm).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NV, int KIND, int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int iters, long long* cyc)
{
    __shared__ double sW[4096];
    for (int t = threadIdx.x; t < 4096; t += blockDim.x) sW[t] = 1e-3 * t;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    double4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (double4_t){ 0, 0, 0, 0 };
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
    int x0 = threadIdx.x, x1 = threadIdx.x + 1, x2 = 3, x3 = 4;
    double f0 = 1.0, f1 = 2.0, f2 = 3.0, f3 = 4.0;
    double l0 = 0, l1 = 0;
    const int addr = (threadIdx.x & 63) * 8;
    long long t0 = clock64();
    if (wv < 12) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                if (MODE == 0) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        if (KIND == 0) { if (v & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x2)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(x3)); }
                        if (KIND == 1) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x0) : "v"(x2), "v"(x3)); }
                        if (KIND == 2) { if (v & 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f0) : "v"(f2), "v"(f3)); else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f1) : "v"(f2), "v"(f3)); }
                        if (KIND == 3) { if (v & 1) asm volatile("ds_read_b64 %0, %1" : "=v"(l0) : "v"(addr)); else asm volatile("ds_read_b64 %0, %1 offset:512" : "=v"(l1) : "v"(addr)); }
                    }
                }
            }
            if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (KIND == 0) { if (v & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x2)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(x1) : "v"(x3)); }
                    if (KIND == 2) { if (v & 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f0) : "v"(f2), "v"(f3)); else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f1) : "v"(f2), "v"(f3)); }
                }
            }
        }
    }
    long long t1 = clock64();
    double s = x0 + x1 + f0 + f1 + l0 + l1;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NV, int KIND, int MODE>
void run(const char* what)
{
    const int iters = 4000, threads = 1024, blocks = 256;
    double* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * threads * blocks);
    (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, KIND, MODE>), dim3(blocks), dim3(threads), 0, 0, out, 50, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, KIND, MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 24 * 4 * blocks; // MFMAs
    printf("%-44s %d per MFMA: %5.1f TFLOP/s, %5.1f ns per MFMA per SIMD (whole kernel, events)\n", what, NV,
           mf * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * 24));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
    run<0, 0, 0>("no extra instructions");
    run<1, 0, 0>("v_add_u32 in the multiplying waves"); run<2, 0, 0>("v_add_u32 in the multiplying waves"); run<4, 0, 0>("v_add_u32 in the multiplying waves");
    run<2, 1, 0>("v_cndmask_b32 in the multiplying waves"); run<4, 1, 0>("v_cndmask_b32 in the multiplying waves");
    run<1, 2, 0>("v_fma_f64 in the multiplying waves"); run<2, 2, 0>("v_fma_f64 in the multiplying waves");
    run<2, 3, 0>("ds_read_b64 in the multiplying waves"); run<4, 3, 0>("ds_read_b64 in the multiplying waves");
    run<2, 0, 1>("v_add_u32 in a fourth wave (x3 MFMAs)"); run<6, 0, 1>("v_add_u32 in a fourth wave (x3 MFMAs)");
    run<2, 2, 1>("v_fma_f64 in a fourth wave (x3 MFMAs)"); run<6, 2, 1>("v_fma_f64 in a fourth wave (x3 MFMAs)");
    return 0;
}
