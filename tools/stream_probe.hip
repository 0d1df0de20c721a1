// Practical HBM ceilings of plain streaming access patterns on MI355X (round-2 probe; not part of the product).
// Patterns: copy (1R1W), axpy-like 3R1W (cg_x_update2), 2R1W (update_p), 4R2W (update_xr), read-only sum -- each at several grid sizes,
// unroll depths and with / without non-temporal hints.  Arrays are 512^3 doubles (1.07 GB): far beyond the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double2 V;  // 16 bytes per lane

template <bool NT> __device__ __forceinline__ V ld(const V* p) {
    if (NT) { V v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(V* p, V v) {
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}

struct Bufs { V *a, *b, *c, *d, *o1, *o2; };
// NR reads, NW writes per element; UN independent elements per thread per trip
template <int NR, int NW, int UN, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void stream_kernel(size_t nvec, const V* a, const V* b, const V* c, const V* d, V* o1, V* o2, double s) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t v0 = (size_t)blockIdx.x * 256 + threadIdx.x; v0 < nvec; v0 += stride * UN) {
        V r[UN][4];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const size_t v = v0 + u * stride;
            if (v < nvec) {
                if (NR > 0) r[u][0] = ld<NTL>(a + v);
                if (NR > 1) r[u][1] = ld<NTL>(b + v);
                if (NR > 2) r[u][2] = ld<NTL>(c + v);
                if (NR > 3) r[u][3] = ld<NTL>(d + v);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const size_t v = v0 + u * stride;
            if (v < nvec) {
                V x = r[u][0];
                if (NR > 1) { x.x += s * r[u][1].x; x.y += s * r[u][1].y; }
                if (NR > 2) { x.x += s * r[u][2].x; x.y += s * r[u][2].y; }
                V y = x;
                if (NR > 3) { y.x = r[u][2].x + s * r[u][3].x; y.y = r[u][2].y + s * r[u][3].y; }
                if (NW > 0) st<NTS>(o1 + v, x);
                if (NW > 1) st<NTS>(o2 + v, y);
            }
        }
    }
}
// block-contiguous: workgroup b owns the C consecutive 4 KB tiles [b*C, (b+1)*C); all loads of a trip (UN tiles) issued before use
template <int NR, int NW, int UN, bool NT>
__global__ __launch_bounds__(256) void chunk_kernel(size_t nvec, int C, const V* a, const V* b, const V* c, const V* d, V* o1, V* o2, double s, double* partials) {
    const size_t base = (size_t)blockIdx.x * C * 256 + threadIdx.x;
    double acc = 0;
    for (int t0 = 0; t0 < C; t0 += UN) {
        V r[UN][4];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const size_t v = base + (size_t)(t0 + u) * 256;
            if (t0 + u < C && v < nvec) {
                if (NR > 0) r[u][0] = ld<NT>(a + v);
                if (NR > 1) r[u][1] = ld<NT>(b + v);
                if (NR > 2) r[u][2] = ld<NT>(c + v);
                if (NR > 3) r[u][3] = ld<NT>(d + v);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const size_t v = base + (size_t)(t0 + u) * 256;
            if (t0 + u < C && v < nvec) {
                V x = r[u][0];
                if (NR > 1) { x.x += s * r[u][1].x; x.y += s * r[u][1].y; }
                if (NR > 2) { x.x += s * r[u][2].x; x.y += s * r[u][2].y; }
                V y = x;
                if (NR > 3) { y.x = r[u][2].x + s * r[u][3].x; y.y = r[u][2].y + s * r[u][3].y; }
                acc += x.x * x.x + x.y * x.y;
                if (NW > 0) st<NT>(o1 + v, x);
                if (NW > 1) st<NT>(o2 + v, y);
            }
        }
    }
    if (NW == 0 || partials) {   // block partial like the solver's reductions
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0 && partials) atomicAdd(&partials[blockIdx.x & 1023], acc);
    }
}
template <int NR, int NW, int UN, bool NT>
void runc(const char* name, const Bufs& B, size_t nvec, int C, double* partials) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    V* o1 = NW > 0 ? B.a : nullptr; V* o2 = NW > 1 ? B.c : nullptr;
    const int grid = (int)((nvec + (size_t)C * 256 - 1) / ((size_t)C * 256));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((chunk_kernel<NR, NW, UN, NT>), dim3(grid), dim3(256), 0, 0, nvec, C, B.a, B.b, B.c, B.d, o1, o2, 1e-9, partials);
    CHK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((chunk_kernel<NR, NW, UN, NT>), dim3(grid), dim3(256), 0, 0, nvec, C, B.a, B.b, B.c, B.d, o1, o2, 1e-9, partials);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)nvec * 16 * (NR + NW);
    printf("chunk %-9s R%dW%d C %3d unroll %d nt %d grid %6d : %.3f ms  %.0f GB/s\n", name, NR, NW, C, UN, (int)NT, grid, ms, bytes / ms / 1e6);
}
__global__ __launch_bounds__(256) void sum_kernel(size_t nvec, const V* a, double* out) {
    double acc = 0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) { V x = a[v]; acc += x.x + x.y; }
    if (acc == 1.2345e300) out[0] = acc;
}

template <int NR, int NW, int UN, bool NTL, bool NTS>
void run(const char* name, const Bufs& B, size_t nvec, int grid) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    // in-place style like the solver: outputs alias the first inputs when NW > 0
    V* o1 = NW > 0 ? B.a : nullptr; V* o2 = NW > 1 ? B.c : nullptr;
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((stream_kernel<NR, NW, UN, NTL, NTS>), dim3(grid), dim3(256), 0, 0, nvec, B.a, B.b, B.c, B.d, o1, o2, 1e-9);
    CHK(hipEventRecord(e0));
    const int reps = 10;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((stream_kernel<NR, NW, UN, NTL, NTS>), dim3(grid), dim3(256), 0, 0, nvec, B.a, B.b, B.c, B.d, o1, o2, 1e-9);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)nvec * 16 * (NR + NW);
    printf("%-10s R%dW%d unroll %d ntl %d nts %d grid %6d : %.3f ms  %.0f GB/s\n", name, NR, NW, UN, (int)NTL, (int)NTS, grid, ms, bytes / ms / 1e6);
}
int main() {
    const size_t N = (size_t)512 * 512 * 512, nvec = N / 2;
    Bufs B; for (V** p : {&B.a, &B.b, &B.c, &B.d, &B.o1, &B.o2}) { CHK(hipMalloc((void**)p, N * 8)); CHK(hipMemset(*p, 0, N * 8)); }
    for (int grid : {1024, 2048, 4096, 8192, 32768, 262144}) {
        run<1, 1, 1, false, false>("copy", B, nvec, grid);
        run<1, 1, 4, false, false>("copy", B, nvec, grid);
    }
    const int g = 2048;
    run<1, 1, 2, false, false>("copy", B, nvec, g); run<1, 1, 2, true, false>("copy", B, nvec, g); run<1, 1, 2, false, true>("copy", B, nvec, g); run<1, 1, 2, true, true>("copy", B, nvec, g);
    for (int grid : {2048, 8192}) {
        run<2, 1, 1, false, false>("update_p", B, nvec, grid); run<2, 1, 2, false, false>("update_p", B, nvec, grid); run<2, 1, 2, true, true>("update_p", B, nvec, grid);
        run<3, 1, 1, false, false>("x_upd2", B, nvec, grid); run<3, 1, 2, false, false>("x_upd2", B, nvec, grid); run<3, 1, 2, true, true>("x_upd2", B, nvec, grid); run<3, 1, 2, false, true>("x_upd2", B, nvec, grid);
        run<4, 2, 1, false, false>("update_xr", B, nvec, grid); run<4, 2, 2, false, false>("update_xr", B, nvec, grid); run<4, 2, 2, true, true>("update_xr", B, nvec, grid);
        run<3, 0, 1, false, false>("read3", B, nvec, grid); run<3, 0, 4, false, false>("read3", B, nvec, grid);
    }
    {   hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1)); double* out; CHK(hipMalloc((void**)&out, 8));
        for (int grid : {2048, 8192}) { hipLaunchKernelGGL(sum_kernel, dim3(grid), dim3(256), 0, 0, nvec, B.a, out);
            CHK(hipEventRecord(e0)); for (int w = 0; w < 10; w++) hipLaunchKernelGGL(sum_kernel, dim3(grid), dim3(256), 0, 0, nvec, B.a, out);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
            printf("sum        R1W0 grid %d : %.3f ms  %.0f GB/s\n", grid, ms, (double)nvec * 16 / ms / 1e6); } }
    {   double* part; CHK(hipMalloc((void**)&part, 1024 * 8)); CHK(hipMemset(part, 0, 1024 * 8));
        for (int C : {1, 2, 4, 8, 16, 32, 64}) {
            runc<1, 1, 1, false>("copy", B, nvec, C, nullptr);
            if (C >= 2) runc<1, 1, 2, false>("copy", B, nvec, C, nullptr);
            if (C >= 4) runc<1, 1, 4, false>("copy", B, nvec, C, nullptr);
        }
        for (int C : {1, 4, 16, 64}) {
            runc<2, 1, 1, false>("update_p", B, nvec, C, nullptr);
            runc<3, 1, 1, false>("x_upd2", B, nvec, C, nullptr);
            runc<4, 2, 1, false>("update_xr", B, nvec, C, part);
            runc<1, 0, 1, false>("norm2", B, nvec, C, part);
            if (C >= 4) { runc<3, 1, 4, false>("x_upd2", B, nvec, C, nullptr); runc<4, 2, 4, false>("update_xr", B, nvec, C, part); runc<1, 0, 4, false>("norm2", B, nvec, C, part);
                          runc<3, 1, 4, true>("x_upd2", B, nvec, C, nullptr); runc<4, 2, 4, true>("update_xr", B, nvec, C, part); }
        }
    }
    return 0;
}
