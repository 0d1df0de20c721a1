// issue-cost micro-benchmark of the far loop's instruction kinds on gfx950: 2 waves per SIMD (like the Step-1 kernel), 8 independent chains per wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    f2 a[8]; double d[8]; float e[8];
    for (int i = 0; i < 8; i++) { a[i] = f2{seed + i + threadIdx.x, seed * 2 + i}; d[i] = seed + i + threadIdx.x; e[i] = seed + i; }
    const f2 c1 = {1.0001f, 0.9999f}, c2 = {1e-3f, 2e-3f};
    const double dc1 = 1.0001, dc2 = 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (KIND == 0) {   // 8 pk_fma
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
                REP8(X)
#undef X
            } else if (KIND == 1) {   // 8 v_exp_f32
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i].x));
                REP8(X)
#undef X
            } else if (KIND == 2) {   // 8 v_rsq_f32
#define X(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i].x));
                REP8(X)
#undef X
            } else if (KIND == 3) {   // 8 pk_fma + 8 exp interleaved (on different registers)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_exp_f32 %1, %1" : "+v"(a[i]), "+v"(e[i]) : "v"(c1), "v"(c2));
                // (d reused as a float holder: only its low half is touched)
                REP8(X)
#undef X
            } else if (KIND == 4) {   // 8 v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(c1.x), "v"(c2.x));
                REP8(X)
#undef X
            } else if (KIND == 5) {   // 8 v_fma_f64
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dc1), "v"(dc2));
                REP8(X)
#undef X
            } else if (KIND == 6) {   // 8 v_rsq_f64
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
                REP8(X)
#undef X
            } else if (KIND == 7) {   // 8 v_mul_f64
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc1));
                REP8(X)
#undef X
            } else if (KIND == 8) {   // 8 v_pk_mul_f32
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
                REP8(X)
#undef X
            } else if (KIND == 9) {   // 8 v_pk_add_f32
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
                REP8(X)
#undef X
            } else if (KIND == 10) {   // 8 fma_f64 + 8 exp_f32 interleaved
#define X(i) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_exp_f32 %1, %1" : "+v"(d[i]), "+v"(a[i].x) : "v"(dc1), "v"(dc2));
                REP8(X)
#undef X
            } else if (KIND == 11) {   // 8 v_add_f64
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc2));
                REP8(X)
#undef X
            } else if (KIND == 12) {   // 8 v_lshl_add_u32
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i].x) : "v"(c2.x));
                REP8(X)
#undef X
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i].x + a[i].y + (float)d[i] + e[i];
    if (s == 12345.678f) out[0] = s;
}
template <int KIND> void run(const char* name, float* out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, 100, 1.f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, iters, 1.f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    // per SIMD: 2 waves x iters x 32 (x2 for the interleaved kinds) instructions
    const double n = 2.0 * iters * 32;
    printf("%-34s %.3f ms   %.2f ns per wave-instruction-slot per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
    float* out; (void)hipMalloc(&out, 4);
    run<4>("v_fma_f32", out); run<0>("v_pk_fma_f32", out); run<8>("v_pk_mul_f32", out); run<9>("v_pk_add_f32", out); run<1>("v_exp_f32", out); run<2>("v_rsq_f32", out);
    run<3>("pk_fma + exp (per pair of instr)", out); run<5>("v_fma_f64", out); run<7>("v_mul_f64", out); run<11>("v_add_f64", out); run<6>("v_rsq_f64", out);
    run<10>("fma_f64 + exp_f32 (per pair)", out); run<12>("v_lshl_add_u32", out);
    return 0;
}
