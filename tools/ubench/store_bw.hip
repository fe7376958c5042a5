// micro-benchmark: streaming-store bandwidth of the W layouts (SoA 8 B/lane x 30 planes vs 16 B/lane x 15 planes
// vs AoS 240 B/obs through per-lane strided stores)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k_soa8(double* W, int64_t O, int64_t Os)
{
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    double v = (double)o;
    double* wp = W + o;
#pragma unroll
    for (int k = 0; k < 30; ++k) { *wp = v + k; wp += Os; }
}
__global__ __launch_bounds__(256) void k_soa16(double2* W, int64_t O, int64_t Os)
{
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    double v = (double)o;
    double2* wp = W + o;
#pragma unroll
    for (int k = 0; k < 15; ++k) { *wp = make_double2(v + k, v - k); wp += Os; }
}
__global__ __launch_bounds__(256) void k_soa8_nt(double* W, int64_t O, int64_t Os)
{
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    double v = (double)o;
    double* wp = W + o;
#pragma unroll
    for (int k = 0; k < 30; ++k) { __builtin_nontemporal_store(v + k, wp); wp += Os; }
}
// blocked by 64 observations: [O / 64][30][64] -- a wave's 30 stores of one step form ONE contiguous 15 KB burst
__global__ __launch_bounds__(256) void k_blk8(double* W, int64_t O, int64_t Os)
{
    int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    double v = (double)o;
    double* wp = W + (o >> 6) * 1920 + (o & 63);
#pragma unroll
    for (int k = 0; k < 30; ++k) { *wp = v + k; wp += 64; }
}
// the derivative kernel's walk: a wave takes a task of ~100 landmarks x 20 observations and writes 60 of them per step
// (off 64-alignment), 2048 waves in flight; SoA planes against the blocked layout
template <int BLK>
__global__ __launch_bounds__(256, 2) void k_walk(double* W, int64_t O, int64_t Os, int per_wave)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t o0 = wave * per_wave;
    for (int i = 0; i + 60 <= per_wave; i += 60) {
        const int64_t o = o0 + i + lane;
        if (lane < 60 && o < O) {
            double v = (double)o;
            if (BLK) {
                double* wp = W + (o >> 6) * 1920 + (o & 63);
#pragma unroll
                for (int k = 0; k < 30; ++k) { *wp = v + k; wp += 64; }
            } else {
                double* wp = W + o;
#pragma unroll
                for (int k = 0; k < 30; ++k) { *wp = v + k; wp += Os; }
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_copy16(const double2* a, double2* b, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
int main()
{
    int64_t O = 2000000, Os = O;
    double* W; hipMalloc(&W, 8 * 30 * Os + 1024);
    double* W2; hipMalloc(&W2, 8 * 30 * Os + 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto f, const char* name, double bytes) {
        for (int i = 0; i < 3; ++i) f();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-12s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
    };
    unsigned blocks = (unsigned)((O + 255) / 256);
    time([&] { hipLaunchKernelGGL(k_soa8, dim3(blocks), dim3(256), 0, 0, W, O, Os); }, "soa 8B", 240.0 * O);
    time([&] { hipLaunchKernelGGL(k_soa16, dim3(blocks), dim3(256), 0, 0, (double2*)W, O, Os); }, "soa 16B", 240.0 * O);
    time([&] { hipLaunchKernelGGL(k_soa8_nt, dim3(blocks), dim3(256), 0, 0, W, O, Os); }, "soa 8B nt", 240.0 * O);
    time([&] { hipLaunchKernelGGL(k_blk8, dim3(blocks), dim3(256), 0, 0, W, O, Os); }, "blocked 8B", 240.0 * O);
    {
        const int per_wave = 960; // 2M observations over ~2084 waves
        unsigned wb = (unsigned)((O / per_wave + 3) / 4);
        time([&] { hipLaunchKernelGGL(k_walk<0>, dim3(wb), dim3(256), 0, 0, W, O, Os, per_wave); }, "walk soa", 240.0 * (O / per_wave) * per_wave);
        time([&] { hipLaunchKernelGGL(k_walk<1>, dim3(wb), dim3(256), 0, 0, W, O, Os, per_wave); }, "walk blocked", 240.0 * (O / per_wave) * per_wave);
    }
    int64_t n = 15 * O;
    time([&] { hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const double2*)W, (double2*)W2, n); }, "copy 16B", 480.0 * O);
    return 0;
}
