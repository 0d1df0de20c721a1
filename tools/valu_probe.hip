// Issue-rate probe for the fp64 instructions of the Step-1 inner loop (gfx950): cycles per wave-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/bin/valu_probe && tools/bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP> __device__ __forceinline__ double op(double x, double y, int k) {
    if (OP == 0) return fma(x, y, 0.5);
    if (OP == 1) return __builtin_amdgcn_rsq(x);
    if (OP == 2) return __builtin_amdgcn_ldexp(x, k);
    if (OP == 3) return x * y;
    if (OP == 4) return x + y;
    if (OP == 5) return __builtin_amdgcn_rcp(x);
    if (OP == 6) return (double)__builtin_amdgcn_rsqf((float)x);   // cvt + rsq_f32 + cvt
    if (OP == 7) return __builtin_amdgcn_fract(x);
    return x;
}
template <int OP> __global__ __launch_bounds__(256) void probe(double* out, int iters, double seed, int k, long long* cyc) {
    double v[8];
#pragma unroll
    for (int a = 0; a < 8; a++) v[a] = seed + a * 0.125 + threadIdx.x * 1e-3;
    const double y = seed * 0.999;
    __syncthreads();
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = op<OP>(v[a], y, k);
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int a = 0; a < 8; a++) s += v[a];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}

typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void probe32(float* out, int iters, float seed, long long* cyc) {
    float2v v[8];
#pragma unroll
    for (int a = 0; a < 8; a++) { v[a].x = seed + a * 0.125f + threadIdx.x * 1e-3f; v[a].y = v[a].x * 1.01f; }
    const float2v y = {seed * 0.999f, seed * 0.998f};
    const float2v hf = {0.5f, 0.25f};
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 8; a++) {
            if (OP == 0) v[a].x = fmaf(v[a].x, y.x, 0.5f);
            if (OP == 1) v[a].x = __builtin_amdgcn_rsqf(v[a].x);
            if (OP == 2) v[a].x = __builtin_amdgcn_exp2f(v[a].x);
            if (OP == 3) v[a] = __builtin_elementwise_fma(v[a], y, hf);   // v_pk_fma_f32
            if (OP == 4) v[a] = v[a] * y;                 // v_pk_mul_f32
            if (OP == 5) v[a] = v[a] + y;                 // v_pk_add_f32
            if (OP == 6) v[a].x = __builtin_amdgcn_sqrtf(v[a].x);
            if (OP == 7) v[a].x = __builtin_amdgcn_rcpf(v[a].x);
        }
    }
    const long long t1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int a = 0; a < 8; a++) s += v[a].x + v[a].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = 0; cyc[1] = t1 - t0; }
}
template <int OP> int run32(const char* name, float seed) {
    const int blocks = 256 * 8, iters = 20000;
    float* out; long long* cyc;
    CHK(hipMalloc(&out, blocks * 256 * sizeof(float)));
    CHK(hipMalloc(&cyc, 16));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe32<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, seed, cyc);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe32<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, seed, cyc);
    hipEventRecord(b);
    CHK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winstr = (double)blocks * 4 / 1024.0 * iters * 8;
    printf("%-28s %8.3f ms  %7.2f ns per wave-instr per SIMD\n", name, ms, ms * 1e6 / winstr);
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}
template <int OP> int run(const char* name, double seed, int k) {
    const int blocks = 256 * 8, iters = 20000;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    double* out; long long* cyc;
    CHK(hipMalloc(&out, blocks * 256 * sizeof(double)));
    CHK(hipMalloc(&cyc, 16));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, seed, k, cyc);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, seed, k, cyc);
    hipEventRecord(b);
    CHK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[2]; CHK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
    // wave-instructions per SIMD: blocks*4 waves / (256 CUs * 4 SIMDs) * iters * 8
    const double winstr = (double)blocks * 4 / 1024.0 * iters * 8;
    printf("%-28s %8.3f ms  %7.2f ns per wave-instr per SIMD   (shader clock64 delta %lld, wall_clock64 delta %lld [100 MHz])\n", name, ms,
           ms * 1e6 / winstr, h[0], h[1]);
    hipFree(out); hipFree(cyc);
    return 0;
}
int main() {
    run<0>("v_fma_f64", 1.0001, 0);
    run<3>("v_mul_f64", 1.0000001, 0);
    run<4>("v_add_f64", 1.0e-9, 0);
    run<1>("v_rsq_f64", 1.37, 0);
    run<5>("v_rcp_f64", 1.37, 0);
    run<2>("v_ldexp_f64", 1.37, 0);
    run<7>("v_fract_f64", 1.37, 0);
    run<6>("cvt+v_rsq_f32+cvt", 1.37, 0);
    run32<0>("v_fma_f32", 1.0001f);
    run32<1>("v_rsq_f32", 1.37f);
    run32<2>("v_exp_f32", -0.37f);
    run32<6>("v_sqrt_f32", 1.37f);
    run32<7>("v_rcp_f32", 1.37f);
    run32<3>("v_pk_fma_f32", 1.0001f);
    run32<4>("v_pk_mul_f32", 1.0000001f);
    run32<5>("v_pk_add_f32", 1e-9f);
    return 0;
}
