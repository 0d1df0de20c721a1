// Fused sweeps of the projected 7-point-stencil CG (the "stencil-PCG" of the north star; replaces the Eigen assembly + sparse LU of
// signed_heat_grid_solver.cpp:101-108 together with the projection kernels).  The canonical loop moves 11 N T bytes per iteration
// in four N-sized kernels (q = Kp | x += a p | r += a q | p = -r + b p); here q = K p is never stored and x is updated every other
// iteration, which leaves 8 N T in three kernels:
//
//   cg_fused_kernel<DIR>   p' = -z + beta p  (z = r, or P M^-1 r with the DCT preconditioner), partial p'.(K p')      3 N T
//                          reads z and p with their one-node halo, writes p' into the OTHER direction buffer (the halo values of p'
//                          are recomputed from the halo of z and p, so p cannot be updated in place)
//   cg_fused_kernel<RES>   r += alpha K p', partial ||r||^2; K p' recomputed from p' (halo re-read)                    3 N T
//   cg_x_update2_kernel    x += alpha_{k-1} p_{k-1} + alpha_k p_k  on odd k (both directions are still in the two buffers)  4 N T / 2
//
// Both stencil kernels share one z-march.  A workgroup of WX * WY waves owns whole x rows (WX waves side by side cover a row, so
// there is no halo in x), WY * RY consecutive rows and a chunk of ZC planes; a lane keeps VEC consecutive x nodes of RY rows of the
// planes k-1, k, k+1 in registers.  x neighbours come from the adjacent lanes (wave shuffles); the values at wave boundaries --
// first / last lane in x, first / last row in y -- are exchanged through LDS once per plane (one barrier per plane, double
// buffered); only the two rows bordering the workgroup's row block and the two planes bordering its chunk are read a second time (a
// factor 1 + 2 / (WY RY) in y, 1 + 2 / ZC in z of loads) -- and those re-reads are L2 hits: the PMC-measured HBM traffic of both sweeps
// is 3.0 N T, the algorithmic figure (profiles/r02_pmc_traffic.json).  The raw loads of plane k+2 are in flight while plane k is computed
// (software prefetch in registers).  Shipped shape: RY = 2 rows per lane (103-117 VGPRs), 8 waves per workgroup (two 512-thread workgroups per CU, out of
// phase) where a row needs one or two waves, 16 waves where it needs four or more (512^3 fp64, 1024^3 fp32: 8 rows per workgroup instead of 4),
// chunks of ~64 planes, ~4 workgroups per CU in the grid; measured alternatives in DESIGN.md 4c / 6.
// K = -L with the reference's Neumann convention (laplacian(): :278-334): an out-of-grid neighbour is the node itself.
#pragma once
#include "shm_kernels.hip.h"

namespace shm {

struct FusedParams {
    int n, nzl, k0;  // grid side, owned planes of this slab, global index of its first owned plane
    int zc;          // planes per workgroup
    int yblocks;     // workgroups along y (z chunks of this launch: gridDim.x / yblocks)
    int zc_first, zc_stride;  // z chunk of a workgroup = zc_first + (logical block / yblocks) * zc_stride: a launch covers all chunks (0, 1), the
                              // interior ones (1, 1) or the first and the last (0, zchunks - 1) -- the latter two overlap the halo exchange
    double inv_h2;
};

enum CgFusedMode : int { CGF_DIR = 0, CGF_RES = 1 };
enum : int { SC_ALPHA_A = 11, SC_ALPHA_B = 12 };  // device scalars (shm_kernels.hip.h: enum Scalar): alpha of even / odd iterations

// streaming (non-temporal) vector access for the arrays a sweep touches exactly once (r in RES, x and the directions in x_update2, the
// p' store of DIR): reuse distance of gigabytes, nothing worth keeping in the caches (tools/stream_probe.hip on 512^3 arrays: +4-8 % on
// 3R1W / 4R2W; x_update2 5.45 -> 6.04 TB/s).  The z / p loads of the stencil sweeps stay plain: their bordering rows and planes are
// re-read by the neighbouring workgroups and should come out of L2 (measured: non-temporal there costs 3 %)
template <typename T, int VEC> struct ExtVec { typedef T type __attribute__((ext_vector_type(VEC))); };
template <typename T> struct ExtVec<T, 1> { typedef T type; };
template <typename T, int VEC>
__device__ __forceinline__ void load_vec_nt(const T* p, T (&v)[VEC]) {
    using V = typename ExtVec<T, VEC>::type;
#ifdef SHM_FUSED_NO_NT
    const V t = *reinterpret_cast<const V*>(p);
#else
    const V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#endif
    if constexpr (VEC == 1) v[0] = t;
    else {
#pragma unroll
        for (int e = 0; e < VEC; e++) v[e] = t[e];
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec_nt(T* p, const T (&v)[VEC]) {
    using V = typename ExtVec<T, VEC>::type;
    V t;
    if constexpr (VEC == 1) t = v[0];
    else {
#pragma unroll
        for (int e = 0; e < VEC; e++) t[e] = v[e];
    }
#ifdef SHM_FUSED_NO_NT
    *reinterpret_cast<V*>(p) = t;
#else
    __builtin_nontemporal_store(t, reinterpret_cast<V*>(p));
#endif
}

template <typename T, int VEC, int RY, int WX, int WY, int MODE>
__global__ __launch_bounds__(WX* WY * 64) void cg_fused_kernel(FusedParams F, double* __restrict__ sc, int slot_old, int slot_new,
                                                               const double* __restrict__ red0, const double* __restrict__ pq, int init, int use_uw,
                                                               int alpha_slot,
                                                               const T* __restrict__ zsrc /* DIR: z */, const T* __restrict__ pin /* DIR: p ; RES: p' */,
                                                               T* __restrict__ pout /* DIR: p' */, T* __restrict__ rio /* RES: r, in place */,
                                                               double* __restrict__ partials,
                                                               const double* __restrict__ pq_partials /* RES, pq_np > 0: the block partials of p'.Kp' the DIR sweep left */,
                                                               int pq_np) {
    constexpr int NW = WX * WY;
    constexpr int RYB = WY * RY;
    // rows exchanged between waves, per parity: slot w = first / last row of wave w; slots NW + wx = the two rows bordering the
    // workgroup's row block (made by its first / last row of waves from their extra loads)
    __shared__ T ylds[2][NW + WX][2][64 * VEC];
    __shared__ T xlds[2][NW][RY][2];
    __shared__ double red[NW];

    // the wave index is made provably wave-uniform (readfirstlane), so that every row base address below is scalar (SGPR pair) and the
    // only per-lane address register is the x offset: global_load ... v_off, s[base:base+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wx = wave % WX, wy = wave / WX;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int yb = (int)(lb % (unsigned)F.yblocks), zci = F.zc_first + (int)(lb / (unsigned)F.yblocks) * F.zc_stride;
    const int n = F.n;
    const size_t plane = (size_t)n * n;
    const int i = (wx * 64 + lane) * VEC;
    const int jb = yb * RYB, j0 = jb + wy * RY;
    const bool ax = i < n;
    const int kk0 = zci * F.zc, kk1 = min(kk0 + F.zc, F.nzl);
    const T ih2 = (T)F.inv_h2;

    // ---- scalars (device resident, as in update_xr_kernel / update_p_kernel)
    T beta = (T)0, alpha = (T)0;
    if (MODE == CGF_DIR) {
        const double rho_new = *red0 - (use_uw ? sc[SC_UW] : 0.);
        const double rho_old = sc[slot_old];
        beta = (T)((init || rho_old == 0.) ? 0. : rho_new / rho_old);
        // every workgroup has read sc[slot_old] before any workgroup of the NEXT kernel runs; the new value goes to the other slot
        if (blockIdx.x == 0 && tid == 0) {
            sc[slot_new] = rho_new;
            if (init) sc[SC_RHO0] = rho_new;
        }
    } else {
        const double rho_cur = sc[slot_old];
        // p'.Kp': either finished by finalize_sum_kernel (*pq), or -- one GPU, round 5 -- summed here by every workgroup from the DIR sweep's block partials, in
        // finalize_sum_kernel's order (one short dependent launch less per iteration; the partials live in an array of their own: this sweep writes `partials`)
        double pqv;
        if (pq_np > 0) {
            __shared__ double pq_s;
            double s = 0.;
            for (int a = tid; a < pq_np; a += NW * 64) s += pq_partials[a];
            s = block_sum(s, red);
            if (tid == 0) pq_s = s;
            __syncthreads();
            pqv = pq_s;
            __syncthreads();   // (red is used again at the end of the sweep)
        } else {
            pqv = *pq;
        }
        const double a = rho_cur == 0. ? 0. : rho_cur / pqv;  // rho == 0: already solved (no 0/0)
        alpha = (T)a;
        if (blockIdx.x == 0 && tid == 0) sc[alpha_slot] = a;  // consumed by cg_x_update2_kernel
    }

    auto exists = [&](int q) { const int kg = F.k0 + q; return kg >= 0 && kg < n; };
    auto row_off = [&](int q, int j) { return (size_t)(q + 1) * plane + (size_t)j * n; };  // wave-uniform part; + i per lane
    const bool lo_halo = (wy == 0) && (jb > 0);               // this wave also produces row jb - 1
    const bool hi_halo = (wy == WY - 1) && (jb + RYB < n);    // ... row jb + RYB
    // with WY > 1 a wave borders the row block on at most one side: one set of registers serves either row
    const bool one_halo = (WY > 1) && (lo_halo || hi_halo);
    const int jh1 = lo_halo ? jb - 1 : jb + RYB;

    // raw registers of the plane being fetched
    T rz[RY][VEC], rp[RY][VEC], rzh[VEC], rph[VEC], rzl[WY == 1 ? VEC : 1], rpl[WY == 1 ? VEC : 1];
    auto issue = [&](int q) {
        if (!ax) return;
        const bool halo = q >= kk0 && q < kk1;  // the plane will be a centre plane: its bordering rows are needed too
#pragma unroll
        for (int r = 0; r < RY; r++) {
            if (j0 + r < n) {
                if (MODE == CGF_DIR) load_vec<T, VEC>(zsrc + row_off(q, j0 + r) + i, rz[r]);
                if (MODE == CGF_RES || !init) load_vec<T, VEC>(pin + row_off(q, j0 + r) + i, rp[r]);
            }
        }
        if constexpr (WY > 1) {
            if (halo && one_halo) {
                if (MODE == CGF_DIR) load_vec<T, VEC>(zsrc + row_off(q, jh1) + i, rzh);
                if (MODE == CGF_RES || !init) load_vec<T, VEC>(pin + row_off(q, jh1) + i, rph);
            }
        } else {
            if (halo && lo_halo) {
                if (MODE == CGF_DIR) load_vec<T, VEC>(zsrc + row_off(q, jb - 1) + i, rzl);
                if (MODE == CGF_RES || !init) load_vec<T, VEC>(pin + row_off(q, jb - 1) + i, rpl);
            }
            if (halo && hi_halo) {
                if (MODE == CGF_DIR) load_vec<T, VEC>(zsrc + row_off(q, jb + RYB) + i, rzh);
                if (MODE == CGF_RES || !init) load_vec<T, VEC>(pin + row_off(q, jb + RYB) + i, rph);
            }
        }
    };
    auto make = [&](const T (&z)[VEC], const T (&p)[VEC], T (&out)[VEC]) {
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            if (MODE == CGF_RES) out[e] = p[e];
            else out[e] = init ? -z[e] : (-z[e] + beta * p[e]);
        }
    };
    // raw -> values of plane q, stores p' where this workgroup owns it; a centre plane's first / last rows, its wave-boundary x values
    // and the rows bordering the row block go to LDS (parity par) for the neighbouring waves
    auto finish = [&](int q, int par, T (&out)[RY][VEC]) {
        if (!ax) return;
        const bool centre = q >= kk0 && q < kk1;
#pragma unroll
        for (int r = 0; r < RY; r++) {
            if (j0 + r < n) {
                make(rz[r], rp[r], out[r]);
                // owned planes of the chunk, plus the slab's ghost planes when a neighbouring slab exists (q = -1 / nzl reach this
                // point only then): the next DIR sweep reads them as the halo of p
                if (MODE == CGF_DIR && (centre || q < 0 || q >= F.nzl)) store_vec_nt<T, VEC>(pout + row_off(q, j0 + r) + i, out[r]);
            }
        }
        if (!centre) return;
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            ylds[par][wave][0][lane * VEC + e] = out[0][e];
            ylds[par][wave][1][lane * VEC + e] = out[RY - 1][e];
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RY; r++) xlds[par][wave][r][0] = out[r][0];
        }
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < RY; r++) xlds[par][wave][r][1] = out[r][VEC - 1];
        }
        T h[VEC];
        if constexpr (WY > 1) {
            if (one_halo) {
                make(rzh, rph, h);
#pragma unroll
                for (int e = 0; e < VEC; e++) ylds[par][NW + wx][lo_halo ? 0 : 1][lane * VEC + e] = h[e];
            }
        } else {
            if (lo_halo) {
                make(rzl, rpl, h);
#pragma unroll
                for (int e = 0; e < VEC; e++) ylds[par][NW + wx][0][lane * VEC + e] = h[e];
            }
            if (hi_halo) {
                make(rzh, rph, h);
#pragma unroll
                for (int e = 0; e < VEC; e++) ylds[par][NW + wx][1][lane * VEC + e] = h[e];
            }
        }
    };

    T prv[RY][VEC], cur[RY][VEC], nxt[RY][VEC], rr[RY][VEC], rrn[RY][VEC];
    auto copy_plane = [&](const T (&a)[RY][VEC], T (&b)[RY][VEC]) {
#pragma unroll
        for (int r = 0; r < RY; r++)
#pragma unroll
            for (int e = 0; e < VEC; e++) b[r][e] = a[r][e];
    };
    auto issue_r = [&](int q, T (&dst)[RY][VEC]) {  // RES: the residual rows of plane q
        if (MODE != CGF_RES || !ax) return;
#pragma unroll
        for (int r = 0; r < RY; r++)
            if (j0 + r < n) load_vec_nt<T, VEC>(rio + row_off(q, j0 + r) + i, dst[r]);
    };

    // ---- prologue: planes kk0-1 (prv), kk0 (cur), raw loads of kk0+1 in flight
    const bool has_prv = exists(kk0 - 1);
    if (has_prv) {
        issue(kk0 - 1);
        finish(kk0 - 1, 0, prv);
    }
    issue(kk0);
    finish(kk0, 0, cur);
    issue_r(kk0, rr);
    if (!has_prv) copy_plane(cur, prv);
    if (exists(kk0 + 1)) issue(kk0 + 1);
    __syncthreads();

    double acc = 0.;
    for (int kk = kk0; kk < kk1; kk++) {
        const int par = (kk - kk0) & 1;
        if (exists(kk + 1)) finish(kk + 1, par ^ 1, nxt);
        else copy_plane(cur, nxt);
        if (kk + 2 <= kk1 && exists(kk + 2)) issue(kk + 2);
        if (kk + 1 < kk1) issue_r(kk + 1, rrn);

        // ---- centre plane kk
        T ylo[VEC], yhi[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            ylo[e] = (wy > 0) ? ylds[par][wave - WX][1][lane * VEC + e] : (lo_halo ? ylds[par][NW + wx][0][lane * VEC + e] : cur[0][e]);
            yhi[e] = (wy < WY - 1) ? ylds[par][wave + WX][0][lane * VEC + e] : (hi_halo ? ylds[par][NW + wx][1][lane * VEC + e] : cur[RY - 1][e]);
        }
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const int j = j0 + r;
            T left = __shfl_up(cur[r][VEC - 1], 1, kWave), right = __shfl_down(cur[r][0], 1, kWave);
            if (lane == 0) left = (wx > 0) ? xlds[par][wave - 1][r][1] : cur[r][0];
            if (lane == 63 && wx < WX - 1) right = xlds[par][wave + 1][r][0];
            if (lane == 63 && wx == WX - 1) right = cur[r][VEC - 1];
            if (i + VEC >= n) right = cur[r][VEC - 1];
            if (ax && j < n) {
                T outv[VEC];
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    const T c = cur[r][e];
                    const T xm = (e == 0) ? left : cur[r][e - 1];
                    const T xp = (e == VEC - 1) ? right : cur[r][e + 1];
                    T up = (r < RY - 1) ? cur[(r < RY - 1) ? r + 1 : r][e] : yhi[e];
                    T dn = (r > 0) ? cur[(r > 0) ? r - 1 : r][e] : ylo[e];
                    if (j == n - 1) up = c;
                    if (j == 0) dn = c;
                    const T s = (xp + up + nxt[r][e] + xm + dn + prv[r][e]) - (T)6 * c;
                    const T qv = -s * ih2;
                    if (MODE == CGF_DIR) {
                        acc += (double)c * (double)qv;
                    } else {
                        const T v = rr[r][e] + alpha * qv;
                        outv[e] = v;
                        acc += (double)v * (double)v;
                    }
                }
                if (MODE == CGF_RES) store_vec_nt<T, VEC>(rio + row_off(kk, j) + i, outv);
            }
        }
        copy_plane(cur, prv);
        copy_plane(nxt, cur);
        if (MODE == CGF_RES) copy_plane(rrn, rr);
        __syncthreads();
    }

    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.;
#pragma unroll
        for (int a = 0; a < NW; a++) s += red[a];
        partials[zci * F.yblocks + yb] = s;
    }
}

// x += a0 pa + a1 pb  (a0 / a1 = alpha of the even / odd iteration; either may be switched off).  4 N T bytes.
// Launch shape (tools/stream_probe.hip on 512^3 arrays): a workgroup owns kXuTiles consecutive 4 KB tiles and the grid covers the array
// once, in dispatch order -- the concurrently active workgroups then sweep one narrow window of each array (5.9 TB/s for this 3R1W
// pattern with non-temporal access) instead of striding over all of it as a persistent grid-stride loop does (5.1 TB/s).
constexpr int kXuTiles = 4;
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void cg_x_update2_kernel(size_t nvec, size_t off, const double* __restrict__ sc, int use_a, int use_b,
                                                              const T* __restrict__ pa, const T* __restrict__ pb, T* __restrict__ x) {
    const T a0 = use_a ? (T)sc[SC_ALPHA_A] : (T)0, a1 = use_b ? (T)sc[SC_ALPHA_B] : (T)0;
    const size_t v0 = (size_t)blockIdx.x * (kXuTiles * kBlock) + threadIdx.x;
    T xv[kXuTiles][VEC], av[kXuTiles][VEC], bv[kXuTiles][VEC];
#pragma unroll
    for (int u = 0; u < kXuTiles; u++) {
        const size_t v = v0 + (size_t)u * kBlock;
        if (v < nvec) {
            const size_t c = off + v * VEC;
            load_vec_nt<T, VEC>(x + c, xv[u]);
            if (use_a) load_vec_nt<T, VEC>(pa + c, av[u]);
            if (use_b) load_vec_nt<T, VEC>(pb + c, bv[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < kXuTiles; u++) {
        const size_t v = v0 + (size_t)u * kBlock;
        if (v < nvec) {
#pragma unroll
            for (int e = 0; e < VEC; e++) {
                if (use_a) xv[u][e] += a0 * av[u][e];
                if (use_b) xv[u][e] += a1 * bv[u][e];
            }
            store_vec_nt<T, VEC>(x + off + v * VEC, xv[u]);
        }
    }
}

}  // namespace shm
