// micro-benchmark: relative error of v_rcp_f64 / v_rsq_f64 seeds and after 1 and 2 Newton steps (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r0 = __builtin_amdgcn_rcp(d);
    double r1 = fma(fma(-d, r0, 1.0), r0, r0);
    double r2 = fma(fma(-d, r1, 1.0), r1, r1);
    double s0 = __builtin_amdgcn_rsq(d);
    double s1 = s0 * fma(-0.5 * d * s0, s0, 1.5);
    double s2 = s1 * fma(-0.5 * d * s1, s1, 1.5);
    out[6 * i] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2;
    out[6 * i + 3] = s0; out[6 * i + 4] = s1; out[6 * i + 5] = s2;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(6 * n);
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        double u = (double)(st >> 11) / 9007199254740992.0;
        x[i] = std::exp((u - 0.5) * 60.0); // 1e-13 .. 1e13
    }
    double *dx, *dout;
    hipMalloc(&dx, 8 * n); hipMalloc(&dout, 48 * n);
    hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 48 * n, hipMemcpyDeviceToHost);
    double e[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < n; ++i) {
        long double rr = 1.0L / x[i], ss = 1.0L / sqrtl((long double)x[i]);
        for (int k2 = 0; k2 < 3; ++k2) {
            e[k2] = fmax(e[k2], (double)fabsl((o[6 * i + k2] - rr) / rr));
            e[3 + k2] = fmax(e[3 + k2], (double)fabsl((o[6 * i + 3 + k2] - ss) / ss));
        }
    }
    printf("rcp: seed %.3e  +1 Newton %.3e  +2 Newton %.3e   (eps = %.3e)\n", e[0], e[1], e[2], 2.22e-16);
    printf("rsq: seed %.3e  +1 Newton %.3e  +2 Newton %.3e\n", e[3], e[4], e[5]);
    return 0;
}
