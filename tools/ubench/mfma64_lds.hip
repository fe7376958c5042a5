// fp64 MFMA rate with LDS-fed operands, as in k_schur_mm: per K step a wave reads RD operand pairs (ds_read_b64 each)
// for 8 MFMAs; RD = 8: every MFMA its own A and B (the kernel); RD = 4 / 2: operands shared by 2 / 4 MFMAs.
// 3 multiplying waves per SIMD (768 threads), no barriers inside the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int RD>
__global__ __launch_bounds__(768) void k(double* out, int iters)
{
    __shared__ double sW[12 * 208], sY[12 * 208];
    for (int t = threadIdx.x; t < 12 * 208; t += blockDim.x) { sW[t] = 1e-3 * t; sY[t] = 2e-3 * t; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lbase = (lane >> 4) * 208 + (lane & 15);
    double4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (double4_t){ 0, 0, 0, 0 };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            double a[RD], b[RD];
            int lb = lbase;
            asm volatile("" : "+v"(lb));
#pragma unroll
            for (int s = 0; s < RD; ++s) {
                a[s] = sW[ks * 4 * 208 + lb + 16 * ((wv + s) % 13)];
                b[s] = sY[ks * 4 * 208 + lb + 16 * ((wv + 2 * s + 1) % 13)];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s % RD], b[(s / (8 / RD)) % RD], acc[s], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int RD>
void run()
{
    const int iters = 4000, threads = 768, blocks = 256;
    double* out;
    (void)hipMalloc(&out, sizeof(double) * threads * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<RD>, dim3(blocks), dim3(threads), 0, 0, out, 50);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<RD>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 24 * 12 * blocks; // MFMAs
    printf("LDS-fed, %d operand pairs read per 8 MFMAs: %.1f TFLOP/s (%.1f ns per MFMA per SIMD)\n", RD,
           mf * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / ((double)iters * 24 * 3));
    (void)hipFree(out);
}
int main() { run<8>(); run<4>(); run<2>(); return 0; }
