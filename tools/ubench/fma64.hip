// micro-benchmark: achievable fp64 VALU FMA rate on gfx950 (the bound of the Schur accumulation) as a function of
// waves per SIMD and independent accumulators per lane, plus the same loop with its operands read from LDS (b128).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NACC>
__global__ __launch_bounds__(512) void k_fma(double* out, int iters, double a, double b)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(x, y, acc[i]);
        asm volatile("" : "+v"(x), "+v"(y)); // keep the loop from collapsing
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 5 x 10 tile fed from LDS like k_schur_grouped: per step 3 reads for 5 w + 5 b128 reads for 10 y, 50 FMAs
__global__ __launch_bounds__(512) void k_fma_lds(double* out, int iters)
{
    __shared__ __attribute__((aligned(16))) double sW[4096];
    __shared__ __attribute__((aligned(16))) double sY[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sW[i] = 1e-3 * i, sY[i] = 1.0 - 1e-4 * i;
    __syncthreads();
    double acc[5][10];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int c = 0; c < 10; ++c) acc[i][c] = 0;
    const int offW = ((threadIdx.x >> 1) % 20) * 36 + 6 * (threadIdx.x & 1), offY = ((threadIdx.x >> 1) % 20) * 30;
    for (int it = 0; it < iters; ++it) {
        const double* wp = sW + offW + 12 * (it % 3) + 720 * (it & 3);
        const double2* yp = reinterpret_cast<const double2*>(sY + offY + 10 * (it % 3) + 600 * (it & 3));
        const double2 w01 = *reinterpret_cast<const double2*>(wp), w23 = *reinterpret_cast<const double2*>(wp + 2);
        const double wr[5] = { w01.x, w01.y, w23.x, w23.y, wp[4] };
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
    double s = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int c = 0; c < 10; ++c) s += acc[i][c];
    out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double* out;
    hipMalloc(&out, 8 * 4096 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](auto f, const char* name, double flops) {
        for (int i = 0; i < 2; ++i) f();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("%-34s %8.1f us  %6.1f TFLOP/s\n", name, ms * 1e3, flops / ms / 1e9);
    };
    const int iters = 2000;
    for (int wgs : { 256, 512, 1024 }) {   // x 512 threads: 2, 4, 8 waves per SIMD when all are resident
        for (int threads : { 256, 512 }) {
            char name[64];
            snprintf(name, sizeof name, "fma acc=50 grid=%d x %d", wgs, threads);
            time([&] { hipLaunchKernelGGL(k_fma<50>, dim3(wgs), dim3(threads), 0, 0, out, iters, 1.0000001, 0.5); }, name,
                 2.0 * 50 * iters * (double)wgs * threads);
        }
    }
    time([&] { hipLaunchKernelGGL(k_fma<16>, dim3(1024), dim3(512), 0, 0, out, iters, 1.0000001, 0.5); }, "fma acc=16 grid=1024 x 512",
         2.0 * 16 * iters * 1024.0 * 512);
    for (int wgs : { 256, 512 })
        for (int threads : { 256, 512 }) {
            char name[64];
            snprintf(name, sizeof name, "lds-fed 5x10 grid=%d x %d", wgs, threads);
            time([&] { hipLaunchKernelGGL(k_fma_lds, dim3(wgs), dim3(threads), 0, 0, out, iters); }, name,
                 2.0 * 50 * iters * (double)wgs * threads);
        }
    return 0;
}
