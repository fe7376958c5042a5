// fp64 MFMA issue rate: v_mfma_f64_16x16x4 on register operands, NACC independent accumulators per wave, 1..4 waves per
// SIMD (256 CUs x 4 SIMDs).  Prints TFLOP/s and cycles per MFMA per SIMD at the clock the chip holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, int iters, long long* cyc)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){ 0, 0, 0, 0 };
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int waves_per_simd)
{
    const int iters = 20000, threads = 64 * 4 * waves_per_simd, blocks = 256;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * threads * blocks);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)iters * NACC * waves_per_simd * 4 * blocks; // MFMAs
    printf("NACC %d, %d waves/SIMD: %.1f TFLOP/s, %.1f shader cycles per MFMA per SIMD (clock64), %.2f GHz\n", NACC, waves_per_simd,
           mf * 2048 / (ms * 1e-3) / 1e12, (double)c / ((double)iters * NACC * waves_per_simd), c / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w = 1; w <= 4; ++w) run<8>(w);
    run<4>(1); run<4>(3); run<2>(4); run<1>(4);
    return 0;
}
