// Probe: accuracy of v_rsq_f64 and of 1 vs 2 Goldschmidt refinement steps (decides the Step-1 inner loop).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* o0, double* o1, double* o2, double* o1b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double y0 = __builtin_amdgcn_rsq(v);
    double g = v * y0, h = 0.5 * y0;
    double e = fma(-g, h, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    o0[i] = y0; o1[i] = 2 * h;
    // variant b: one Goldschmidt + residual-corrected sqrt:  d = x - g^2 ; g' = g + d*h
    double d = fma(-g, g, v);
    o1b[i] = fma(d, h, g);
    e = fma(-g, h, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    o2[i] = 2 * h;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n);
    for (int i = 0; i < n; i++) x[i] = std::exp((drand48() - 0.5) * 40.0);
    double *dx, *d0, *d1, *d2, *d1b;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d1b, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, d1b, n);
    std::vector<double> r0(n), r1(n), r2(n), r1b(n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1b.data(), d1b, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e1b = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / sqrtl((long double)x[i]);
        e0 = fmax(e0, fabs((double)((r0[i] - t) / t))); e1 = fmax(e1, fabs((double)((r1[i] - t) / t))); e2 = fmax(e2, fabs((double)((r2[i] - t) / t)));
        const long double ts = sqrtl((long double)x[i]);
        e1b = fmax(e1b, fabs((double)((r1b[i] - ts) / ts)));
    }
    printf("max rel err: seed %.3e  1 step (rsqrt) %.3e  2 steps %.3e  1 step + residual (sqrt) %.3e\n", e0, e1, e2, e1b);
    return 0;
}
