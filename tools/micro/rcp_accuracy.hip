// accuracy of v_rcp_f64 on gfx950 and of one / two Newton refinements (decides how many steps rcp_nr in obca_model.h needs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, double *r3, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double d = x[i], r = __builtin_amdgcn_rcp(d);
    r0[i] = r; { const double e = fma(-d, r, 1.0), s_ = fma(e, e, e); r3[i] = fma(r, s_, r); }      /* r (1 + e + e^2): four dependent operations */
    r = fma(r, fma(-d, r, 1.0), r); r1[i] = r; r = fma(r, fma(-d, r, 1.0), r); r2[i] = r;
}
int main() {
    const int n = 1 << 22; std::vector<double> x(n), a(n), b(n), c(n), e(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = ldexp(1.0 + u, (int)(s % 80) - 40); }
    double *dx, *d0, *d1, *d2, *d3; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, d3, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(e.data(), d3, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; i++) { long double t = 1.0L / (long double)x[i]; e0 = fmax(e0, (double)fabsl((a[i] - t) / t)); e1 = fmax(e1, (double)fabsl((b[i] - t) / t)); e2 = fmax(e2, (double)fabsl((c[i] - t) / t)); e3 = fmax(e3, (double)fabsl((e[i] - t) / t)); }
    printf("max relative error: v_rcp_f64 %.3e (2^%.1f)  +1 Newton %.3e (%.2f ulp)  +2 Newton %.3e (%.2f ulp)  r(1+e+e^2) %.3e (%.2f ulp)\n", e0, log2(e0), e1, e1 / 1.11e-16, e2, e2 / 1.11e-16, e3, e3 / 1.11e-16);
    return 0;
}
