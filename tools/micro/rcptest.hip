// Accuracy of v_rcp_f64 on gfx950 and of 1 / 2 Newton steps on top of it (the reciprocal of the scan kernels, fast_rcp in
// device_math.h): max |1 - x r| over 2^24 arguments, measured with an FMA residual.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* e0, double* e1, double* e2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    e0[i] = fabs(fma(-v, r, 1.0));
    r = fma(fma(-v, r, 1.0), r, r);
    e1[i] = fabs(fma(-v, r, 1.0));
    r = fma(fma(-v, r, 1.0), r, r);
    e2[i] = fabs(fma(-v, r, 1.0));
}
int main() {
    const int n = 1 << 24;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = (1.0 + u) * std::ldexp(1.0, (int)(s % 121) - 60);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, 8 * n); hipMalloc(&d0, 8 * n); hipMalloc(&d1, 8 * n); hipMalloc(&d2, 8 * n);
    hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, 8 * n, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, 8 * n, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, 8 * n, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m0 = std::fmax(m0, a[i]); m1 = std::fmax(m1, b[i]); m2 = std::fmax(m2, c[i]); }
    printf("max |1 - x r|: v_rcp_f64 %.3e (2^%.1f), + 1 Newton step %.3e (2^%.1f), + 2 steps %.3e (2^%.1f)\n", m0, std::log2(m0), m1,
           std::log2(m1), m2, std::log2(m2));
    return 0;
}
