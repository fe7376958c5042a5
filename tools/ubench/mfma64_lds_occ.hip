// mfma64_lds.hip at ONE workgroup per CU (96 KB of padding LDS), i.e. three multiplying waves per SIMD as in k_schur_mm, and
// with PF = 0 / 1: the next K step's operands requested before / after this K step's MFMAs (sched barriers pin the order).
// Question: is an LDS-fed fp64 MFMA stream at three waves per SIMD latency-bound, and does explicit prefetch fix it?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int RD, int PF, int PAD>
__global__ __launch_bounds__(768) void k(double* out, int iters)
{
    __shared__ double sW[12 * 208], sY[12 * 208];
    __shared__ double pad[PAD];
    for (int t = threadIdx.x; t < 12 * 208; t += blockDim.x) { sW[t] = 1e-3 * t; sY[t] = 2e-3 * t; }
    if (iters < 0) pad[threadIdx.x] = 1;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lbase = (lane >> 4) * 208 + (lane & 15);
    double4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (double4_t){ 0, 0, 0, 0 };
    double a[2][RD], b[2][RD];
    auto load = [&](int set, int ks) {
        int lb = lbase;
        asm volatile("" : "+v"(lb));
#pragma unroll
        for (int s = 0; s < RD; ++s) {
            a[set][s] = sW[ks * 4 * 208 + lb + 16 * ((wv + s) % 13)];
            b[set][s] = sY[ks * 4 * 208 + lb + 16 * ((wv + 2 * s + 1) % 13)];
        }
    };
    auto mac = [&](int set) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[set][s % RD], b[set][(s / (8 / RD)) % RD], acc[s], 0, 0, 0);
    };
    if (PF) load(0, 0);
    for (int it = 0; it < iters; ++it) {
        if (PF) {
            load(1, 1); __builtin_amdgcn_sched_barrier(0); mac(0); __builtin_amdgcn_sched_barrier(0);
            load(0, 2); __builtin_amdgcn_sched_barrier(0); mac(1); __builtin_amdgcn_sched_barrier(0);
            load(1, 0); __builtin_amdgcn_sched_barrier(0); mac(0); __builtin_amdgcn_sched_barrier(0);
            load(0, 1); __builtin_amdgcn_sched_barrier(0); mac(1); __builtin_amdgcn_sched_barrier(0);
            load(1, 2); __builtin_amdgcn_sched_barrier(0); mac(0); __builtin_amdgcn_sched_barrier(0);
            load(0, 0); __builtin_amdgcn_sched_barrier(0); mac(1); __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) { load(0, ks % 3); mac(0); }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (iters < 0 ? pad[0] : 0.0);
}
template <int RD, int PF, int PAD>
void run()
{
    const int iters = 2000, threads = 768, blocks = 256 * (PAD > 5000 ? 1 : 2);
    double* out;
    (void)hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<RD, PF, PAD>), dim3(blocks), dim3(threads), 0, 0, out, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<RD, PF, PAD>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 48 * 12 * blocks; // MFMAs
    const double waves_per_simd = PAD > 5000 ? 3 : 6;
    printf("%s waves/SIMD, %d operand pairs per 8 MFMAs, prefetch %d: %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", PAD > 5000 ? "3" : "6", RD, PF,
           mf * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * 48 * waves_per_simd));
    (void)hipFree(out);
}
int main()
{
    run<2, 0, 12000>(); run<2, 1, 12000>(); run<4, 0, 12000>(); run<4, 1, 12000>(); run<8, 0, 12000>(); run<8, 1, 12000>();
    run<2, 0, 16>(); run<2, 1, 16>(); run<4, 0, 16>(); run<4, 1, 16>();
    return 0;
}
