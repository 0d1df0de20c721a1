// micro-benchmark: dense row-major mat-vec variants (one workgroup per row)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kBlock = 256;
__device__ __forceinline__ double block_sum(double s, double* lds) {
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    double r = 0.;
    if (threadIdx.x == 0) for (int a = 0; a < kBlock / 64; a++) r += lds[a];
    return r;
}
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT, int ROWS>
__global__ __launch_bounds__(kBlock) void mv(int m, int ld, const double* __restrict__ G, const double* __restrict__ w, double* __restrict__ u) {
    __shared__ double lds[8];
    for (int q = 0; q < ROWS; q++) {
        const int row = blockIdx.x * ROWS + q;
        if (row >= m) break;
        const double* g = G + (size_t)row * ld;
        double s = 0.;
        for (int c0 = 0; c0 < m; c0 += U * kBlock * 4) {
            d2 gv[U][2];
#pragma unroll
            for (int a = 0; a < U; a++) {
                const int c = c0 + (threadIdx.x + a * kBlock) * 4;
                if (c < m) {
                    if (NT) { gv[a][0] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(g + c)); gv[a][1] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(g + c + 2)); }
                    else { gv[a][0] = *reinterpret_cast<const d2*>(g + c); gv[a][1] = *reinterpret_cast<const d2*>(g + c + 2); }
                }
            }
#pragma unroll
            for (int a = 0; a < U; a++) {
                const int c = c0 + (threadIdx.x + a * kBlock) * 4;
                if (c < m) {
                    double t = gv[a][0].x * w[c];
                    if (c + 1 < m) t += gv[a][0].y * w[c + 1];
                    if (c + 2 < m) t += gv[a][1].x * w[c + 2];
                    if (c + 3 < m) t += gv[a][1].y * w[c + 3];
                    s += t;
                }
            }
        }
        s = block_sum(s, lds);
        if (threadIdx.x == 0) u[row] = s;
        if (ROWS > 1) __syncthreads();
    }
}
// one wave per row, 4 rows per workgroup, no barrier
template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void mv_wave(int m, int ld, const double* __restrict__ G, const double* __restrict__ w, double* __restrict__ u) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= m) return;
    const double* g = G + (size_t)row * ld;
    double s = 0.;
    for (int c0 = 0; c0 < m; c0 += U * 64 * 4) {
        d2 gv[U][2];
#pragma unroll
        for (int a = 0; a < U; a++) {
            const int c = c0 + (lane + a * 64) * 4;
            if (c < m) {
                if (NT) { gv[a][0] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(g + c)); gv[a][1] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(g + c + 2)); }
                else { gv[a][0] = *reinterpret_cast<const d2*>(g + c); gv[a][1] = *reinterpret_cast<const d2*>(g + c + 2); }
            }
        }
#pragma unroll
        for (int a = 0; a < U; a++) {
            const int c = c0 + (lane + a * 64) * 4;
            if (c < m) {
                double t = gv[a][0].x * w[c];
                if (c + 1 < m) t += gv[a][0].y * w[c + 1];
                if (c + 2 < m) t += gv[a][1].x * w[c + 2];
                if (c + 3 < m) t += gv[a][1].y * w[c + 3];
                s += t;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) u[row] = s;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main() {
    for (int m : {2842, 7748, 12612}) {
        const int ld = (m + 63) / 64 * 64;
        double *G, *w, *u;
        CK(hipMalloc(&G, (size_t)ld * ld * 8)); CK(hipMalloc(&w, ld * 8)); CK(hipMalloc(&u, ld * 8));
        CK(hipMemset(G, 0, (size_t)ld * ld * 8)); CK(hipMemset(w, 0, ld * 8));
        const double gb = (double)m * ld * 8 / 1e9;
#define RUN(name, ...) { float ms = timeit([&] { __VA_ARGS__; }); printf("m=%5d %-28s %.4f ms  %.2f TB/s\n", m, name, ms, gb / ms); }
        RUN("wg/row U=1", hipLaunchKernelGGL((mv<1, false, 1>), dim3(m), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wg/row U=4", hipLaunchKernelGGL((mv<4, false, 1>), dim3(m), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wg/row U=8", hipLaunchKernelGGL((mv<8, false, 1>), dim3(m), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wg/row U=16", hipLaunchKernelGGL((mv<16, false, 1>), dim3(m), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wg/row U=8 nt", hipLaunchKernelGGL((mv<8, true, 1>), dim3(m), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wg/row U=16 nt", hipLaunchKernelGGL((mv<16, true, 1>), dim3(m), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wg/4rows U=8", hipLaunchKernelGGL((mv<8, false, 4>), dim3((m + 3) / 4), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wave/row U=8", hipLaunchKernelGGL((mv_wave<8, false>), dim3((m + 3) / 4), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wave/row U=16", hipLaunchKernelGGL((mv_wave<16, false>), dim3((m + 3) / 4), dim3(kBlock), 0, 0, m, ld, G, w, u));
        RUN("wave/row U=16 nt", hipLaunchKernelGGL((mv_wave<16, true>), dim3((m + 3) / 4), dim3(kBlock), 0, 0, m, ld, G, w, u));
        hipFree(G); hipFree(w); hipFree(u);
    }
    return 0;
}
