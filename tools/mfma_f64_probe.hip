// Throughput of v_mfma_f64_16x16x4_f64 on gfx950 (round-2 probe for the Step-1 accumulate-on-matrix-cores experiment):
// (a) independent accumulators back to back, (b) interleaved with independent fp64 VALU work, (c) VALU result feeding the MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = seed + threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6, v[4] = {a, a + 1, a + 2, a + 3};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (MODE >= 1) {   // ~22 dependent-free fp64 VALU ops per MFMA (what Step 1 does per pair)
#pragma unroll
                for (int u = 0; u < 22; u++) v[e] = __builtin_fma(v[e], 1.0000001, 1e-9);
            }
            const double g = MODE == 2 ? v[e] : a;
            if (MODE != 3) acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(g, b, acc[e], 0, 0, 0);
            else { acc[e][0] = __builtin_fma(v[e], b, acc[e][0]); acc[e][1] = __builtin_fma(v[e], a, acc[e][1]); acc[e][2] = __builtin_fma(v[e], b + a, acc[e][2]); }
        }
    }
    double s = v[0] + v[1] + v[2] + v[3];
    for (int e = 0; e < 4; e++) s += acc[e][0] + acc[e][1] + acc[e][2] + acc[e][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double* out, int waves_per_simd) {
    const int iters = 20000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 4 * waves_per_simd;   // MFMAs (or FMA triples) per SIMD
    printf("%-44s waves/SIMD %d: %.3f ms  -> %.1f ns per MFMA-slot per SIMD (%.0f cycles at 2.2 GHz)\n", name, waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.2);
}
int main() {
    double* out; hipMalloc(&out, 256 * 8 * 256 * 8);
    for (int w : {1, 2, 4}) {
        run<0>("mfma only", out, w);
        run<1>("mfma + 22 independent VALU fma per mfma", out, w);
        run<2>("mfma fed by the VALU chain", out, w);
        run<3>("VALU only: 22 fma + 3 accumulate fma", out, w);
    }
    return 0;
}
