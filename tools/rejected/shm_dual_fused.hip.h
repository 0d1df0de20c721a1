// REJECTED (round 2, measured): one-launch form of the dual iteration's m-vector half with grid-wide barriers.
//   bunny_small 256^3 (m = 2842), per iteration, same box:   seven kernels 49 us (+ 4 us scatter, + launch gaps)  |  this kernel 63 us
//   64^3: 38 us | 75 us.   With agent-scope fences in the barrier instead of sc1 atomics: 750-1150 us (L2 write-back / invalidate per wave).
// Each of the five phases is a dependent chain across XCDs (sc1 store -> counter -> poll -> sc1 load -> LDS stage -> block reductions): ~12 us
// per phase whatever the grid size (16 ... 256 workgroups gave the same time), no better than a kernel boundary.  Kept for reference only; not
// compiled into the library.  What shipped instead: fewer, ordinary launches (see solve_dual).
//
// The m-vector half of one dual-CG iteration (signed_heat_grid_solver.cpp:101-107 replaced by CG on S = A K^+ A^T, see the block comment
// above dual_init_mu_kernel) in ONE launch.  Round 1 issued seven kernels for it -- gather A z, dual_update, G^-1, B, G^-1,
// dual_direction, scatter A^T p -- each a few microseconds of work behind 5-8 us of launch and drain latency (256^3: 62 us of the
// 176 us iteration).  Here a grid of one 1024-thread workgroup per CU walks the same five phases separated by four grid-wide barriers:
//
//   A  Sp = A z                       rows over all threads (8 entries each)
//   --------------------------------- barrier
//   B  alpha = r.z / p.Pm(Sp)         every workgroup forms the two dots itself, in the same order: the same bits everywhere, no broadcast
//      t1 = G^-1 (r - alpha Pm Sp)    a wave per row; the updated residual is formed on the fly, and written once (by slices) into the
//      mu += alpha p, r' = ...        OTHER residual buffer so that no workgroup reads a half-updated vector
//   --------------------------------- barrier
//   C  t2 = B t1                      sparse (A K A^T, 27 entries per row), a thread per row
//   --------------------------------- barrier
//   D  z = G^-1 t2                    a wave per row
//   --------------------------------- barrier
//   E  beta = r'.Pm(z) / r.z          dots again per workgroup;  p' = Pm z + beta p by slices into the other direction buffer,
//      w = A^T p'                     node-major scatter into the grid array the x sweep of K^+ reads next, p' formed on the fly
//
// Pm = removal of the mean (S is singular along the constant).  G^-1 is the single-precision copy of the dense inverse (the preconditioner
// only has to be a fixed SPD operator), 4 m^2 bytes per application, read through L2 / MALL.
//
// Grid barrier: one arrival counter in global memory, monotone over the launches of a solve (epoch passed as an argument).  The eight XCDs
// have private L2s, so what one workgroup writes for the others must not sit in (or be read from) a non-coherent line.  An agent-scope
// release / acquire fence pair per barrier does that by writing back and invalidating the L2 -- measured 0.2-0.3 ms per barrier with 4096
// waves, twenty times the kernels this launch replaces.  Instead every vector that crosses a barrier (Sp, t1, t2, z, r') is written and
// read with relaxed agent-scope atomics (global_load / global_store with sc1: coherent at the device level, no cache maintenance), each
// workgroup staging the vector it needs once in LDS; the barrier itself is then a relaxed counter behind the workgroup barrier's
// s_waitcnt vmcnt(0).  Everything else (G^-1, the CSR lists, r, p of the previous launch) is ordinary cached data.  The launch is
// an ordinary one sized to the resident capacity; should the workgroups not all become resident (another process's kernels filling the
// device) the spin gives up after kSpinLimitNs, raises sc[SC_FUSED_ERR], and the host redoes the solve with the seven-kernel form.
#pragma once
#include "shm_kernels.hip.h"

namespace shm {

constexpr int SC_FUSED_ERR = 13;                 // != 0: a grid barrier of dual_mspace_kernel timed out
constexpr long long kSpinLimitNs = 40000000LL;   // 40 ms at the 100 MHz s_memrealtime clock (below)

struct DualFusedArgs {
    int m, ld;                 // constraint rows; leading dimension of Ginv32
    const int* row_ptr;        // A by rows (gather)
    const uint32_t* ent_node;
    const double* ent_coef;
    int n_touched;             // A^T by nodes (scatter)
    const uint32_t* node_id;
    const int* node_ptr;
    const int* ent_row;
    const double* nent_coef;
    const float* Ginv32;
    const int* Bptr;
    const int* Bcol;
    const double* Bval;
    double* mu;
    double* Sp;
    double* t1;
    double* t2;
    double* z;
    const double* r_in;
    double* r_out;
    const double* p_in;
    double* p_out;
    double* sc;
    unsigned* bar;
};

__device__ __forceinline__ double xload(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xstore(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target, double* sc) {
    __syncthreads();   // every wave's stores have been acknowledged (s_waitcnt vmcnt(0) before s_barrier)
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (((long long)__builtin_amdgcn_s_memrealtime() - t0) * 10 > kSpinLimitNs) {   // 100 MHz counter: 10 ns per tick
                sc[SC_FUSED_ERR] = 1.;
                break;
            }
        }
    }
    __syncthreads();
}

// dot of row `g` (single precision, ld-padded: every float4 load is in bounds) with the vector v staged in LDS (entries >= m are zero up to
// the next multiple of 4)
__device__ __forceinline__ double wave_row_dot(const float* __restrict__ g, int m, const double* v) {
    const int lane = threadIdx.x & 63;
    double s = 0.;
    constexpr int kU = 4;   // float4 loads in flight per lane
    for (int c0 = lane * 4; c0 < m; c0 += 64 * 4 * kU) {
        float4 gv[kU];
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int c = c0 + u * 256;
            if (c < m) gv[u] = *reinterpret_cast<const float4*>(g + c);
        }
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const int c = c0 + u * 256;
            if (c < m) s += (double)gv[u].x * v[c] + (double)gv[u].y * v[c + 1] + (double)gv[u].z * v[c + 2] + (double)gv[u].w * v[c + 3];
        }
    }
    return wave_sum(s);   // valid in lane 0
}

// dynamic LDS: ((m + 3) & ~3) doubles
template <typename T>
__global__ __launch_bounds__(kDualBlock) void dual_mspace_kernel(DualFusedArgs a, const T* __restrict__ zarr, T* __restrict__ parr, unsigned epoch0) {
    __shared__ double lds[17];
    extern __shared__ double vec[];
    const int m = a.m, m4 = (m + 3) & ~3;
    const int gid = blockIdx.x * kDualBlock + threadIdx.x, gsz = gridDim.x * kDualBlock;
    const int wid = blockIdx.x * (kDualBlock / kWave) + (threadIdx.x >> 6), nwv = gridDim.x * (kDualBlock / kWave);
    const int lane = threadIdx.x & 63;
    unsigned target = epoch0;
    // ---- A: Sp = A z
    for (int row = gid; row < m; row += gsz) {
        double s = 0.;
        for (int e = a.row_ptr[row]; e < a.row_ptr[row + 1]; e++) s += a.ent_coef[e] * (double)zarr[a.ent_node[e]];
        xstore(a.Sp + row, s);
    }
    grid_barrier(a.bar, target += gridDim.x, a.sc);
    // ---- B: alpha; r' = r - alpha Pm Sp (staged in LDS, written once by slices); t1 = G^-1 r'; mu += alpha p
    const double rz_cur = a.sc[SC_RZ];
    double alpha;
    {
        double s = 0.;
        for (int j = threadIdx.x; j < m4; j += kDualBlock) {
            const double v = j < m ? xload(a.Sp + j) : 0.;
            vec[j] = v;
            s += v;
        }
        const double meanSp = block_sum_1024(s, lds) / (double)m;
        double pSp = 0.;
        for (int j = threadIdx.x; j < m; j += kDualBlock) pSp += a.p_in[j] * (vec[j] - meanSp);
        pSp = block_sum_1024(pSp, lds);
        alpha = rz_cur == 0. ? 0. : rz_cur / pSp;   // r.z == 0: already solved (m == 1, or an exact start) -- no 0/0
        double rr = 0.;
        for (int j = threadIdx.x; j < m; j += kDualBlock) {   // a thread rewrites only the entries it staged itself
            const double v = a.r_in[j] - alpha * (vec[j] - meanSp);
            vec[j] = v;
            rr += v * v;
        }
        rr = block_sum_1024(rr, lds);   // (its barriers also publish vec to the whole workgroup)
        if (blockIdx.x == 0 && threadIdx.x == 0) a.sc[SC_RR] = rr;
    }
    for (int row = wid; row < m; row += nwv) {
        const double s = wave_row_dot(a.Ginv32 + (size_t)row * a.ld, m, vec);
        if (lane == 0) xstore(a.t1 + row, s);
    }
    for (int j = gid; j < m; j += gsz) {
        a.mu[j] += alpha * a.p_in[j];
        xstore(a.r_out + j, vec[j]);
    }
    grid_barrier(a.bar, target += gridDim.x, a.sc);
    // ---- C: t2 = B t1
    for (int row = gid; row < m; row += gsz) {
        double s = 0.;
        for (int e = a.Bptr[row]; e < a.Bptr[row + 1]; e++) s += a.Bval[e] * xload(a.t1 + a.Bcol[e]);
        xstore(a.t2 + row, s);
    }
    grid_barrier(a.bar, target += gridDim.x, a.sc);
    // ---- D: z = G^-1 t2
    for (int j = threadIdx.x; j < m4; j += kDualBlock) vec[j] = j < m ? xload(a.t2 + j) : 0.;
    __syncthreads();
    for (int row = wid; row < m; row += nwv) {
        const double s = wave_row_dot(a.Ginv32 + (size_t)row * a.ld, m, vec);
        if (lane == 0) xstore(a.z + row, s);
    }
    grid_barrier(a.bar, target += gridDim.x, a.sc);
    // ---- E: beta; p' = Pm z + beta p; w = A^T p'
    double beta, meanZ;
    {
        double s = 0.;
        for (int j = threadIdx.x; j < m; j += kDualBlock) {
            const double v = xload(a.z + j);
            vec[j] = v;
            s += v;
        }
        meanZ = block_sum_1024(s, lds) / (double)m;
        double rz = 0.;
        for (int j = threadIdx.x; j < m; j += kDualBlock) rz += xload(a.r_out + j) * (vec[j] - meanZ);
        rz = block_sum_1024(rz, lds);
        beta = rz_cur == 0. ? 0. : rz / rz_cur;
        if (blockIdx.x == 0 && threadIdx.x == 0) a.sc[SC_RZ] = rz;   // every workgroup read the old value in phase B, two barriers ago
    }
    for (int j = gid; j < m; j += gsz) a.p_out[j] = (vec[j] - meanZ) + beta * a.p_in[j];
    for (int t = gid; t < a.n_touched; t += gsz) {
        double s = 0.;
        for (int e = a.node_ptr[t]; e < a.node_ptr[t + 1]; e++) {
            const int row = a.ent_row[e];
            s += a.nent_coef[e] * ((vec[row] - meanZ) + beta * a.p_in[row]);
        }
        parr[a.node_id[t]] = (T)s;
    }
}

}  // namespace shm
