// Hand-written gfx950 (CDNA4, wave64) kernels of the grid Signed Heat Method hot path.
// Every kernel cites the reference lines (into nzfeng/signed-heat-3d) whose arithmetic it reproduces.
//
// Data layout in HBM (per z-slab): every N-vector is stored as (nzl+2) planes of n*n values -- one
// ghost plane below, nzl owned planes, one ghost plane above -- x fastest (idx = i + j*n + kk*n*n),
// so that the 7-point stencil and the z-differences index ghosts uniformly.  The reference's AoS
// Y[3*idx+p] (signed_heat_grid_solver.cpp:58) is stored as three planar arrays (3N > 2^31 at 1024^3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "shm_mc_table.h"

namespace shm {

constexpr int kWave = 64;       // CDNA wavefront
constexpr int kBlock = 256;     // 4 waves = one per SIMD
constexpr int kMaxPartials = 8192;

// ---- wave / block reductions (double accumulators everywhere) -----------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// Sum over a 256-thread block; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* lds /* >= 4 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double s = 0.;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x * blockDim.y + 63) >> 6;
        for (int a = 0; a < nw; a++) s += lds[a];
    }
    return s;
}

// XCD-aware block remap (8 XCDs, block b is dispatched to XCD b%8): give every XCD one contiguous
// range of logical blocks so that the z-neighbour planes a block re-reads sit in its own L2.
// Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// =================================================================================================
// Step 1 + 2: X(x) = sum_s (N_s A_s) exp(-lambda |x-b_s|)/|x-b_s| ;  Y = X/|X|
//   reference: signed_heat_grid_solver.cpp:48-65 (mesh), :157-174 (points); yukawaPotential
//   signed_heat_3d.cpp:45-49; node position :510-514.
// Lanes = grid nodes (NPT per thread for ILP), sources staged through LDS in tiles of kSrcTile and
// broadcast-read by every lane.  Compute-bound (sqrt+exp+div per pair); HBM traffic is 3 stores/node.
// For T=float the exponent is offset per node by a lower bound d0 of every source distance so that
// exp() cannot underflow to 0 for all sources (SURVEY trap #4); the factor cancels in X/|X|.
// =================================================================================================
constexpr int kSrcTile = 256;

// exp(-lambda r)/r evaluated without a division and without the library sqrt: y = rsqrt(r^2) (hardware seed +
// two Goldschmidt steps), r = r^2 y, 1/r = y.  fp64 exp: range reduction by ln2 (two-term Cody-Waite) + degree-13
// Taylor polynomial on |f| <= ln2/2 (truncation 4e-18) + ldexp; the argument is always <= 0 and > -745.
template <typename T> struct YukawaMath;
template <> struct YukawaMath<double> {
    // e^{-lambda r}/r from x = |d|^2, division-free.  25 VALU + 3 integer instructions per (node, source) pair.
    //  - r and 1/r: v_rsq_f64 seed y0 (relative error 2^-24), then ONE third-order step: with t = x y0 ~ r and
    //    e = 1 - x y0^2,  (1 - e)^(-1/2) = 1 + e/2 + 3e^2/8 + O(e^3)  (remainder 5e^3/16 < 1e-21), so r = t (1 + q), 1/r = y0 (1 + q).
    //  - exp: u = r c with c = -lambda 2048/ln2 is the exponent in units of ln2/2048; k = round(u) by the 1.5*2^52 trick (the integer
    //    lands in the low mantissa bits), f = fma(r, c, -k) is the remainder with ONE rounding (|f| <= 1/2), and
    //    e^{-lambda r} = 2^(k >> 11) * T[k & 2047] * P3(f),  T[j] = 2^(j/2048) from a 16 KB LDS table (the LDS pipe is otherwise idle
    //    in this kernel), P3 the degree-3 Taylor polynomial of 2^(f/2048) (truncation 3.4e-17).  The only error the reference's
    //    std::exp(-lambda * r) does not share is the rounding of c, relative 1.1e-16 * lambda r in the result -- the same size as
    //    the reference's own rounding of the product lambda * r.
    static constexpr int kExpTabBits = 11;
    static __device__ __forceinline__ double yukawa(double x, double c, const double* __restrict__ tab /* LDS: 2^(j/2048) */) {
        const double y0 = __builtin_amdgcn_rsq(x);
        const double t = x * y0;
        const double e = fma(-t, y0, 1.0);
        const double q = fma(e, 0.375, 0.5) * e;
        const double r = fma(t, q, t);
        const double rinv = fma(y0, q, y0);
        const double tm = fma(r, c, 6755399441055744.0);   // 1.5 * 2^52
        const double kf = tm - 6755399441055744.0;
        const int ki = (int)(unsigned)__double_as_longlong(tm);
        const double f = fma(r, c, -kf);
        double p = 6.461528672932366e-12;                   // (ln2/2048)^3 / 6
        p = fma(p, f, 5.727446245172041e-08);               // (ln2/2048)^2 / 2
        p = fma(p, f, 3.384507717577858e-04);               // ln2/2048
        p = fma(p, f, 1.0);
        return __builtin_amdgcn_ldexp(tab[ki & ((1 << kExpTabBits) - 1)] * p * rinv, ki >> kExpTabBits);   // r = 0 -> NaN (0 * inf)
    }
    // table-free variant (range reduction by ln2 + degree-13 polynomial): the tile epilogue and tests
    static __device__ __forceinline__ double exp_neg(double x) {
        const double kf = rint(x * 1.4426950408889634074);
        double f = fma(kf, -6.93147180369123816490e-01, x);
        f = fma(kf, -1.90821492927058770002e-10, f);
        double p = 1.6059043836821613e-10;                 // 1/13!
        p = fma(p, f, 2.08767569878681e-09);               // 1/12!
        p = fma(p, f, 2.505210838544172e-08);              // 1/11!
        p = fma(p, f, 2.755731922398589e-07);              // 1/10!
        p = fma(p, f, 2.7557319223985893e-06);             // 1/9!
        p = fma(p, f, 2.48015873015873e-05);               // 1/8!
        p = fma(p, f, 1.984126984126984e-04);              // 1/7!
        p = fma(p, f, 1.3888888888888889e-03);             // 1/6!
        p = fma(p, f, 8.333333333333333e-03);              // 1/5!
        p = fma(p, f, 4.1666666666666664e-02);             // 1/4!
        p = fma(p, f, 1.6666666666666666e-01);             // 1/3!
        p = fma(p, f, 0.5);
        p = fma(p, f, 1.0);
        p = fma(p, f, 1.0);
        return __builtin_amdgcn_ldexp(p, (int)kf);
    }
};
template <> struct YukawaMath<float> {
    static __device__ __forceinline__ void rsqrt_and_sqrt(float x, float& rinv, float& r) {
        const float y0 = __builtin_amdgcn_rsqf(x);  // 1 ulp
        r = x * y0;
        rinv = y0;
    }
    static __device__ __forceinline__ float exp_neg(float x) { return __expf(x); }
};
template <typename T> __device__ __forceinline__ T t_rcp(T x);
template <> __device__ __forceinline__ double t_rcp<double>(double x) { return __builtin_amdgcn_rcp(x); }
template <> __device__ __forceinline__ float t_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }

struct ConvParams {
    int n;              // nodes per side
    int kk_begin;       // first local plane (0 = low ghost) to evaluate
    int kk_end;         // one past the last local plane
    int k0;             // global k of local plane 1 (first owned plane)
    int tiles_x, tiles_y;  // 8x8x8-node tiles per axis (tiles_z = gridDim.x / (tiles_x*tiles_y))
    double bbox_min[3];
    double cell;
    double lambda;
    double cexp;        // -lambda * 2048 / ln 2 (fp64 path: exponent in table units)
    int S;              // padded to whole clusters
    int n_clusters;
    float far_gap;      // fp64 path: a cluster is "far" when d_lo(tile, cluster) - r_hi(tile) > far_gap
    float tier_log;     // tiered fp64 path (shm_conv_tiered.hip.h): a source is "far" for a sub-tile when all its terms are below e^-tier_log of the sub-tile's dominant terms
    float skip_base;    // conv_normalize_kernel only: ln(S / eps) -- a cluster (largest weight A_c) is skipped when lambda gap > skip_base + ln(A_c / A_near); the tiered kernels drop by accumulated bound (drop_* below)
    float inv_lambda;
    int exact_offset;   // fp32 path only: 1 = per-node nearest-source distance (coarse grids: lambda * tile diameter too large)
    int n_tiles;        // total tiles; workgroups stride over them (fewer workgroups than slots leave room for the set-up stream)
    double wscale;         // tiered fp64 path: power of two that brings the largest source weight into (0.5, 1] -- applied to the fp32 copies of the weights only
    double pad_pos[3];     // tiered fp64 path: where the zero-weight padding entries of a compacted source list sit (bbox_min - n cell: a grid side from every node)
    float far_redo_ratio;  // tiered fp64 path: a block whose packed-fp32 sums exceed this fraction of |X| at any node evaluates its far sources again in fp64
                           // (= budget on Y / calibrated relative error of a packed-fp32 term; 3e38: never)
    int far_rule;             // tiered fp64 path: 1 = the differential far rule beside the box rule (shm_conv_tiered.hip.h; per launch from Solver::far_rule_now, per layer of the queue order from Solver::far_rule_plan)
    const unsigned char* unit_rule;        // tiered fp64 path: far rule per layer of the queue order (nullptr: far_rule everywhere) -- 1 sample, 0 box, 2 as the sample decides
    unsigned long long* sample_ctr;        // ... [0] far pairs, [1] pairs evaluated again, [2] blocks done of the sample (zeroed before the launch)
    int sample_blocks;                     // ... blocks of the sample
    const int* layer_order;   // tiered path: z-layer of blocks that stands at position p of the work queues' order (nullptr: p itself) -- see launch_conv
    // tiered path, round 6: the drop rule by ACCUMULATED bound (shm_conv_tiered.hip.h).  b_s = (|w_s| / |w_*|) e^{-lambda gap_s} r_hi / d_s bounds a source's terms against the
    // terms of the block's reference source s* at every node of the block.  A source is a candidate when b_s <= 2^drop_ltau and is dropped while the running sum of the
    // block's dropped bounds stays <= drop_eps_soft; one with b_s <= 2^drop_ltau_hard = drop_tau_hard (the S-times-worst-case rule of rounds 3-5 on an eighth of the
    // budget) is dropped whatever the sum.  drop_eps_soft = 0: nothing is dropped.
    float drop_eps_soft, drop_ltau, drop_ltau_hard, drop_tau_hard;
    float tier_u0;            // fp64 solve: exponent beyond which the a-posteriori test prices a packed-fp32 term at eps_far u / u0 (0: flat)
    int tier_flush;           // fp64 solve: packed-fp32 sources between two flushes of their sums into the fp64 accumulators (0: never)
    float drop_check;         // fp64 solve: 1 / eps_far -- what a unit of dropped bound weighs against the packed-fp32 tier's L1 sums in the a-posteriori test
};

typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int kConvTile = 8;      // 8x8x8 nodes per workgroup, 2 per lane
// sources per cluster (Morton-sorted on the host): 64 in the fp64 solve, 32 in the fp32 solve.  Smaller clusters have tighter bounding spheres -- 10 % fewer
// (tile, source) pairs survive the culling at SprayBottle 1024^3 (5.38 -> 5.21 s) -- but twice the per-cluster bookkeeping, which the fp64 kernel on the
// bunny (nothing to cull) pays with +1.2 % (42.0 -> 42.5 ms); the culled configurations are the fp32 ones
template <typename T> constexpr int conv_cluster() { return sizeof(T) == 8 ? 64 : 32; }
template <typename T> constexpr int conv_chunk() { return (sizeof(T) == 8 ? kSrcTile : 4 * kSrcTile) / conv_cluster<T>(); }  // clusters per LDS fill (256 / 1024 sources)
constexpr int kConvClusterRec = 6;  // floats per cluster record: bounding sphere (centre, radius), ln of its largest source weight, ln of the sum of its source weights (round 6)

// Workgroup = one compact 8x8x8 tile of nodes (2 per lane); sources arrive as Morton-sorted clusters of 64 with bounding
// spheres, staged through LDS 8 clusters at a time and broadcast-read by every lane.
//   T = double: near clusters use the reference's fp64 arithmetic (no exponent offset).  A cluster whose every term is below
//     e^-25 of the tile's dominant term (lower distance bound d_lo minus the tile's nearest-source upper bound r_hi exceeds
//     far_gap) is evaluated in fp32 with the tile offset d0 and folded in as (double)f * exp(-lambda d0): its rounding error
//     stays below 2e-16 of the sum.  The classification is uniform over the workgroup (no divergence).
//   T = float: everything in fp32 with d0 = max(0, min_s |c_tile - b_s| - r_tile), a lower bound of every (node, source)
//     distance that exceeds the nearest one by at most the tile diameter, so exp(-lambda (r - d0)) of the nearest source never
//     underflows (SURVEY trap #4); the common factor exp(lambda d0) cancels in X/|X|.  Coarse grids (lambda * tile
//     diameter > 30) take one extra sweep for exact per-node offsets.
// Two sources in flight per lane, also at 8 nodes per lane: 162 registers = three waves per SIMD.  (Round 2 ran four in flight -- 228 registers, two waves --
// which was 4 % faster while the tiles were strided statically; with the tile queue the third wave is worth more than the deeper unroll: rocker 512^3
// 383.5 against 386.6 ms, SprayBottle 1024^3 4163 against 4238 ms; three in flight, 175 registers: 419 / -- .)
#ifndef SHM_CONV32_UNROLL
#define SHM_CONV32_UNROLL 2
#endif
template <int NPT> struct ConvSrcUnroll { static constexpr int value = NPT == 8 ? SHM_CONV32_UNROLL : 2; };   // sources in flight per lane in the packed fp32 loop
template <typename T, int NPT>
__global__ __launch_bounds__(kBlock) void conv_normalize_kernel(ConvParams P, const T* __restrict__ src /* [S][6]: pos xyz, wn xyz */,
                                                                const float* __restrict__ src32, const float* __restrict__ clusters,
                                                                const double* __restrict__ exp_tab_g /* [2048]: 2^(j/2048) */,
                                                                T* __restrict__ Y0, T* __restrict__ Y1, T* __restrict__ Y2,
                                                                unsigned long long* __restrict__ counters /* [0] fp64, [1] fp32 (node, source) pairs evaluated; may be null */,
                                                                unsigned* __restrict__ next_tile /* work-queue head, zeroed before the launch; null: tiles strided statically */) {
    // tile = 8 x 8 x (4 NPT) nodes; a lane owns the z-column (i, j, k0 + w + 4 e), e < NPT (w = its wave): the NPT nodes share
    // dx^2 + dy^2 of every source
    constexpr int kTileZ = (kBlock / 64) * NPT;
    constexpr bool kMixed = sizeof(T) == 8;
    // fp32, 8 nodes per lane: the tile is culled as two halves of 16 planes (nodes e < 4 / e >= 4 of every lane), each with its own bounding sphere,
    // nearest-source bound and exponent offset -- exactly the quantities of two NPT = 4 tiles, so the skipped set and every node's sum are those of the
    // NPT = 4 kernel bit for bit, while the clusters both halves need (most of them where the kernel spans the object) share dx^2 + dy^2 and the LDS
    // reads over 8 nodes instead of 4 (rocker 512^3: 484 -> 450 ms)
    constexpr int kHalves = (!kMixed && NPT == 8) ? 2 : 1;
    constexpr int kNH = NPT / kHalves;            // nodes of a lane per half
    constexpr int kHalfPlanes = kTileZ / kHalves;
    // sources per LDS fill: 256 (4 clusters) in fp64, where the fill shares the LDS with the exponential table and the fp32 copy; 1024 (16 clusters,
    // 24 KB) in fp32, whose LDS is otherwise empty -- a quarter of the fills and barriers (rocker 512^3: 54 -> 14 per tile)
    constexpr int kConvCluster = conv_cluster<T>();
    constexpr int kChunk = conv_chunk<T>();
    constexpr int kFill = kChunk * kConvCluster;
    __shared__ T tile[kFill * 6];
    __shared__ float tile32[kMixed ? kFill * 6 : 1];
    __shared__ float red[kBlock / kWave], redw[kBlock / kWave];
    constexpr int kTab = kMixed ? (1 << YukawaMath<double>::kExpTabBits) : 1;
    __shared__ double exp_tab[kTab];
    if (kMixed)
        for (int a = threadIdx.x; a < kTab; a += kBlock) exp_tab[a] = exp_tab_g[a];
    const int n = P.n;
    const size_t plane = (size_t)n * n;
    unsigned cnt_half64 = 0, cnt_half32 = 0;   // (cluster, half tile) pairs evaluated in fp64 / fp32 by this workgroup (uniform: scalar registers)
    // Tiles are handed out by a queue head (one atomic per tile): with culling their costs differ severalfold, and on a thin z-slab (a multi-GPU rank)
    // a static stride leaves the workgroups 10-20 % apart at the end of the launch.
    __shared__ int queue_bt;
    auto next_bt = [&](int cur) {
        if (!next_tile) return cur + (int)gridDim.x;
        __syncthreads();   // (also: everybody has read the previous value)
        if (threadIdx.x == 0) queue_bt = (int)atomicAdd(next_tile, 1u);
        __syncthreads();
        return queue_bt;
    };
    for (int bt = next_tile ? next_bt(0) : (int)blockIdx.x; bt < P.n_tiles; bt = next_bt(bt)) {
    __syncthreads();  // LDS reuse between consecutive tiles of this workgroup
    const int tz = bt / (P.tiles_x * P.tiles_y), trem = bt - tz * (P.tiles_x * P.tiles_y);
    const int ty = trem / P.tiles_x, tx = trem - ty * P.tiles_x;
    const int i0 = tx * kConvTile, j0 = ty * kConvTile, kk0 = P.kk_begin + tz * kTileZ;

    T pz[NPT], ax[NPT], ay[NPT], az[NPT];
    float qz[NPT], fx[NPT], fy[NPT], fz[NPT];
    bool live[NPT];
    size_t vidx[NPT];
    const int li = i0 + (threadIdx.x & 7), lj = j0 + ((threadIdx.x >> 3) & 7);
    const int ci = min(li, n - 1), cj = min(lj, n - 1);
    // indicesToNodePosition: (i,j,k)*cellSize + bboxMin, evaluated in double like the reference
    const double xd = ci * P.cell + P.bbox_min[0], yd = cj * P.cell + P.bbox_min[1];
    const T px = (T)xd, py = (T)yd;
    const float qx = (float)xd, qy = (float)yd;
#pragma unroll
    for (int e = 0; e < NPT; e++) {
        int kk = kk0 + (int)(threadIdx.x >> 6) + e * (kBlock / 64);
        live[e] = li < n && lj < n && kk < P.kk_end;
        kk = min(kk, P.kk_end - 1);
        vidx[e] = (size_t)kk * plane + (size_t)cj * n + ci;
        const int k = P.k0 + kk - 1;
        const double z = k * P.cell + P.bbox_min[2];
        pz[e] = (T)z;
        qz[e] = (float)z;
        ax[e] = ay[e] = az[e] = (T)0;
        fx[e] = fy[e] = fz[e] = 0.f;
    }
    // tile centre / circumscribed radius, then the workgroup-wide minimum distance from the centre to the sources
    constexpr double kHalfZ = 0.5 * (kHalfPlanes - 1);
    const float cx = (float)((i0 + 3.5) * P.cell + P.bbox_min[0]), cy = (float)((j0 + 3.5) * P.cell + P.bbox_min[1]);
    const float rt = (float)(sqrt(3.5 * 3.5 * 2 + kHalfZ * kHalfZ) * P.cell) * 1.000001f;
    float cz[kHalves], ln_anear[kHalves], r_hi[kHalves], d0t[kHalves];
#pragma unroll
    for (int h = 0; h < kHalves; h++) {
        cz[h] = (float)((P.k0 + kk0 + h * kHalfPlanes - 1 + kHalfZ) * P.cell + P.bbox_min[2]);
        float dmin = 3.0e38f, wnear = 0.f;   // nearest source and |A N|^2 of it (ties -- the zero-weight padding repeats a source -- go to the larger weight)
        for (int s = threadIdx.x; s < P.S; s += kBlock) {
            const float dx = cx - (float)src[(size_t)s * 6], dy = cy - (float)src[(size_t)s * 6 + 1], dz = cz[h] - (float)src[(size_t)s * 6 + 2];
            const float wx = (float)src[(size_t)s * 6 + 3], wy = (float)src[(size_t)s * 6 + 4], wz = (float)src[(size_t)s * 6 + 5];
            const float d2 = dx * dx + dy * dy + dz * dz, w2 = wx * wx + wy * wy + wz * wz;
            if (d2 < dmin || (d2 == dmin && w2 > wnear)) {
                dmin = d2;
                wnear = w2;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(dmin, off, kWave), ow = __shfl_xor(wnear, off, kWave);
            if (od < dmin || (od == dmin && ow > wnear)) {
                dmin = od;
                wnear = ow;
            }
        }
        if (h > 0) __syncthreads();   // the previous half's partials have been read
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6] = dmin;
            redw[threadIdx.x >> 6] = wnear;
        }
        __syncthreads();
        dmin = red[0];
        wnear = redw[0];
#pragma unroll
        for (int a = 1; a < kBlock / kWave; a++)
            if (red[a] < dmin || (red[a] == dmin && redw[a] > wnear)) {
                dmin = red[a];
                wnear = redw[a];
            }
        dmin = sqrtf(dmin);
        ln_anear[h] = 0.5f * __logf(fmaxf(wnear, 1e-37f)) - 1e-5f;   // rounded down
        r_hi[h] = dmin * 1.000001f + rt;                   // every node of the (half) tile has a source at most this far
        d0t[h] = fmaxf(0.f, dmin * 0.999999f - rt);        // no source is closer than this to any node of the (half) tile
    }

    T d0[NPT];
#pragma unroll
    for (int e = 0; e < NPT; e++) d0[e] = kMixed ? (T)0 : (T)d0t[e / kNH];
    if (!kMixed && P.exact_offset) {
        // coarse grid, fp32 solve: one extra sweep over the sources for the exact nearest distance of every node
        float m2[NPT];
#pragma unroll
        for (int e = 0; e < NPT; e++) m2[e] = 3.0e38f;
        for (int s0 = 0; s0 < P.S; s0 += kFill) {
            const int cnt = min(kFill, P.S - s0);
            __syncthreads();
            for (int a = threadIdx.x; a < cnt * 6; a += kBlock) tile[a] = src[(size_t)s0 * 6 + a];
            __syncthreads();
            for (int s = 0; s < cnt; s++) {
                const float dx = qx - (float)tile[6 * s], dy = qy - (float)tile[6 * s + 1];
                const float dxy2 = dx * dx + dy * dy;
#pragma unroll
                for (int e = 0; e < NPT; e++) {
                    const float dz = qz[e] - (float)tile[6 * s + 2];
                    m2[e] = fminf(m2[e], dxy2 + dz * dz);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < NPT; e++) d0[e] = (T)(sqrtf(m2[e]) * 0.99999f);
    }
    const float lamf = (float)P.lambda;
    const float cexp32 = (float)(-P.lambda * 1.4426950408889634);   // fp32 path: exponent in powers of two
    float coff[NPT];
#pragma unroll
    for (int e = 0; e < NPT; e++) coff[e] = (float)(P.lambda * 1.4426950408889634 * (double)d0[e]);
    for (int c0 = 0; c0 < P.n_clusters; c0 += kChunk) {
        const int ncl = min(kChunk, P.n_clusters - c0);
        const int cnt = ncl * kConvCluster;
        // per cluster: lower bound of every (node of the tile, source of the cluster) distance minus the upper bound of the tile's distance
        // to its nearest source -- from the bounding spheres (uniform addresses: scalar loads), before anything is staged: a chunk whose
        // clusters are all negligible costs neither the LDS fill nor its barriers.  All branches below are workgroup-uniform.
        auto record_gap = [&](const float* rec, int h) {
            const float gdx = cx - rec[0], gdy = cy - rec[1], gdz = cz[h] - rec[2];
            return sqrtf(gdx * gdx + gdy * gdy + gdz * gdz) * 0.999999f - rt - rec[3] - r_hi[h];
        };
        if constexpr (!kMixed) {   // the whole fill first: its bounding sphere contains every cluster's and its weight bounds theirs, so passing this test implies every
            // cluster of the fill passes its own (same skipped set, one test instead of kChunk -- two thirds of the fills at SprayBottle 1024^3)
            const float* rec = clusters + (size_t)(P.n_clusters + c0 / kChunk) * kConvClusterRec;
            bool all_skipped = true;
#pragma unroll
            for (int h = 0; h < kHalves; h++) all_skipped = all_skipped && record_gap(rec, h) > (P.skip_base + rec[4] - ln_anear[h]) * P.inv_lambda;
            if (all_skipped) continue;
        }
        // bit c: cluster c0 + c is skipped for that half (all its terms together are below the rounding unit of the half's dominant term) / far
        unsigned skipmask[kHalves];
#pragma unroll
        for (int h = 0; h < kHalves; h++) skipmask[h] = 0;
        bool any = false;
        for (int c = 0; c < ncl; c++) {
            const float* rec = clusters + (size_t)(c0 + c) * kConvClusterRec;
#pragma unroll
            for (int h = 0; h < kHalves; h++) {
                const float gap = record_gap(rec, h);
                const bool sk = gap > (P.skip_base + rec[4] - ln_anear[h]) * P.inv_lambda;
                skipmask[h] |= sk ? (1u << c) : 0u;
                any = any || !sk;
            }
        }
        if (!any) continue;
        __syncthreads();
        for (int a = threadIdx.x; a < cnt * 6; a += kBlock) {
            tile[a] = src[(size_t)c0 * kConvCluster * 6 + a];
            if (kMixed) tile32[a] = src32[(size_t)c0 * kConvCluster * 6 + a];
        }
        __syncthreads();
#pragma unroll 1
        for (int c = 0; c < ncl; c++) {
            const bool sk0 = (skipmask[0] >> c) & 1u, sk1 = (skipmask[kHalves - 1] >> c) & 1u;
            if (sk0 && sk1) continue;
            // (fp64: classified here, from the scalar record, not from a mask built above -- with the mask the compiler schedules the fp64 loop
            // below into 165 instead of 241 registers and Step 1 takes 44.0 instead of 40.8 ms at 256^3)
            const bool far = kMixed && record_gap(clusters + (size_t)(c0 + c) * kConvClusterRec, 0) > P.far_gap;
            if (kMixed && !far) cnt_half64 += 1;
            else cnt_half32 += (kHalves == 1 ? 1u : (unsigned)!sk0 + (unsigned)!sk1);
            if (far) {
#pragma unroll 2
                for (int s = c * kConvCluster; s < (c + 1) * kConvCluster; s++) {
                    const float sz = tile32[6 * s + 2];
                    const float wx = tile32[6 * s + 3], wy = tile32[6 * s + 4], wz = tile32[6 * s + 5];
                    const float dx = qx - tile32[6 * s], dy = qy - tile32[6 * s + 1];
                    const float dxy2 = dx * dx + dy * dy;
#pragma unroll
                    for (int e = 0; e < NPT; e++) {
                        const float dz = qz[e] - sz;
                        float r, rinv;
                        YukawaMath<float>::rsqrt_and_sqrt(dxy2 + dz * dz, rinv, r);
                        const float g = YukawaMath<float>::exp_neg(-lamf * (r - d0t[0])) * rinv;
                        fx[e] += wx * g; fy[e] += wy * g; fz[e] += wz * g;
                    }
                }
            } else if constexpr (kMixed) {
#pragma unroll 2
                for (int s = c * kConvCluster; s < (c + 1) * kConvCluster; s++) {
                    const T sz = tile[6 * s + 2];
                    const T wx = tile[6 * s + 3], wy = tile[6 * s + 4], wz = tile[6 * s + 5];
                    const T dx = px - tile[6 * s], dy = py - tile[6 * s + 1];
                    const T dxy2 = dx * dx + dy * dy;
#pragma unroll
                    for (int e = 0; e < NPT; e++) {
                        const T dz = pz[e] - sz;
                        const T g = YukawaMath<double>::yukawa(dxy2 + dz * dz, P.cexp, exp_tab);   // r = 0 -> NaN, like exp(0)/0 = inf -> NaN after normalise
                        ax[e] += wx * g; ay[e] += wy * g; az[e] += wz * g;
                    }
                }
            } else {
                // fp32: the lane's nodes two at a time in packed registers (v_pk_add/mul/fma_f32: two values per issue slot); per pair
                // of nodes and source: 8 packed + 2 v_rsq_f32 + 2 v_exp_f32.  exp(-lambda (r - d0)) = 2^(r c + c0), c = -lambda log2(e).
                static_assert(NPT % 2 == 0, "packed fp32 path handles the lane's nodes in pairs");
                auto sweep = [&](auto e_begin, auto e_end) {   // nodes [e_begin, e_end) of every lane against cluster c
                    constexpr int E0 = decltype(e_begin)::value, E1 = decltype(e_end)::value;
#pragma unroll ConvSrcUnroll<NPT>::value   // sources in flight per lane (see ConvSrcUnroll)
                    for (int s = c * kConvCluster; s < (c + 1) * kConvCluster; s++) {
                        const float sz = tile[6 * s + 2];
                        const float wx = tile[6 * s + 3], wy = tile[6 * s + 4], wz = tile[6 * s + 5];
                        const float dx = px - tile[6 * s], dy = py - tile[6 * s + 1];
                        const float dxy2 = dx * dx + dy * dy;
#pragma unroll
                        for (int e = E0; e < E1; e += 2) {
                            const float2v z2 = {pz[e], pz[e + 1]}, c02 = {coff[e], coff[e + 1]};
                            const float2v dz = z2 - sz;
                            const float2v d2 = __builtin_elementwise_fma(dz, dz, float2v{dxy2, dxy2});
                            const float2v rinv = {__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};   // r = 0 -> inf -> NaN below
                            const float2v r = d2 * rinv;
                            const float2v arg = __builtin_elementwise_fma(r, float2v{cexp32, cexp32}, c02);
                            const float2v ex = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                            const float2v g = ex * rinv;
                            float2v a;
                            a = __builtin_elementwise_fma(float2v{wx, wx}, g, float2v{ax[e], ax[e + 1]}); ax[e] = a.x; ax[e + 1] = a.y;
                            a = __builtin_elementwise_fma(float2v{wy, wy}, g, float2v{ay[e], ay[e + 1]}); ay[e] = a.x; ay[e + 1] = a.y;
                            a = __builtin_elementwise_fma(float2v{wz, wz}, g, float2v{az[e], az[e + 1]}); az[e] = a.x; az[e + 1] = a.y;
                        }
                    }
                };
                using std::integral_constant;
                if constexpr (kHalves == 1) sweep(integral_constant<int, 0>{}, integral_constant<int, NPT>{});
                else if (!sk0 && !sk1) sweep(integral_constant<int, 0>{}, integral_constant<int, NPT>{});
                else if (!sk0) sweep(integral_constant<int, 0>{}, integral_constant<int, kNH>{});
                else sweep(integral_constant<int, kNH>{}, integral_constant<int, NPT>{});
            }
        }
    }
    if (kMixed) {
        const double e0 = YukawaMath<double>::exp_neg(-P.lambda * (double)d0t[0]);
#pragma unroll
        for (int e = 0; e < NPT; e++) {
            ax[e] += (T)((double)fx[e] * e0);
            ay[e] += (T)((double)fy[e] * e0);
            az[e] += (T)((double)fz[e] * e0);
        }
    }
#pragma unroll
    for (int e = 0; e < NPT; e++) {
        if (!live[e]) continue;
        T x0 = ax[e], x1 = ay[e], x2 = az[e];
        if (sizeof(T) == 4) {  // fp32 only: pre-scale so that the squares cannot underflow (|X| itself may be ~1e-30)
            const T mx = fmaxf(fmaxf(fabsf((float)x0), fabsf((float)x1)), fabsf((float)x2));
            x0 /= mx; x1 /= mx; x2 /= mx;  // mx = 0 -> NaN, the reference's 0/0
        }
        const T nrm = t_sqrt<T>(x0 * x0 + x1 * x1 + x2 * x2);
        Y0[vidx[e]] = x0 / nrm;  // 0/0 -> NaN exactly like X /= X.norm() (:61)
        Y1[vidx[e]] = x1 / nrm;
        Y2[vidx[e]] = x2 / nrm;
    }
    }  // tile loop
    if (counters && threadIdx.x == 0) {
        constexpr unsigned long long kPairsPerHalf = (unsigned long long)kConvCluster * (kBlock * kNH);
        if (cnt_half64) atomicAdd(&counters[0], cnt_half64 * kPairsPerHalf);
        if (cnt_half32) atomicAdd(&counters[1], cnt_half32 * kPairsPerHalf);
    }
}

// =================================================================================================
// divYt = D^T Y  (gradient(): signed_heat_grid_solver.cpp:336-402, product :71, scrub :72-74)
// Closed form of the transpose of the forward-difference operator with its mirrored last row:
//   b += ( [a>=1] Ya[a-1] + [a==n-1] Ya[a] - [a<n-1] Ya[a] - [a==n-2] Ya[a+1] ) / h   per axis a.
// One streaming pass: reads Y0 (row), Y1 (rows j+-1), Y2 (planes k+-1, ghosts), writes b.  4NT bytes.
// =================================================================================================
struct GridParams {
    int n;      // nodes per side
    int nzl;    // owned planes of this slab
    int k0;     // global k of the first owned plane
    double inv_h;
    double inv_h2;
};

// Launch: one workgroup per (row segment of 256 nodes, row j, owned plane kk) -- blockIdx = (x chunk, j, kk): no index divisions, every
// load a contiguous row segment (the neighbours in y / z are the same segment of the adjacent row / plane: L2 hits), workgroups in dispatch
// order sweep the arrays front to back.  (Round 1's flat grid-stride loop spent its time on two 64-bit divisions per node: 0.18 ms at
// 256^3 = 3.0 TB/s.)
template <typename T>
__global__ __launch_bounds__(kBlock) void divergence_kernel(GridParams G, const T* __restrict__ Y0, const T* __restrict__ Y1,
                                                            const T* __restrict__ Y2, T* __restrict__ b, int scrub) {
    const int n = G.n;
    const size_t plane = (size_t)n * n;
    const int i = blockIdx.x * kBlock + threadIdx.x, j = blockIdx.y, kk = blockIdx.z;
    if (i >= n) return;
    const int k = G.k0 + kk;
    const size_t c = (size_t)(kk + 1) * plane + (size_t)j * n + i;  // skip the low ghost plane
    const T ih = (T)G.inv_h;
    T acc = (T)0;
    // x axis
    if (i >= 1) acc += ih * Y0[c - 1];
    if (i == n - 1) acc += ih * Y0[c];
    if (i < n - 1) acc -= ih * Y0[c];
    if (i == n - 2) acc -= ih * Y0[c + 1];
    // y axis
    if (j >= 1) acc += ih * Y1[c - n];
    if (j == n - 1) acc += ih * Y1[c];
    if (j < n - 1) acc -= ih * Y1[c];
    if (j == n - 2) acc -= ih * Y1[c + n];
    // z axis
    if (k >= 1) acc += ih * Y2[c - plane];
    if (k == n - 1) acc += ih * Y2[c];
    if (k < n - 1) acc -= ih * Y2[c];
    if (k == n - 2) acc -= ih * Y2[c + plane];
    if (scrub && !isfinite(acc)) acc = (T)0;
    b[c] = acc;
}

// =================================================================================================
// q = K p with K = -L, L the reference's 7-point Laplacian (laplacian(): :278-334): an out-of-grid
// neighbour is replaced by the node itself, everything divided by h^2.  Fused with the partial
// p.q reduction (double accumulators, one partial per block -> deterministic two-stage sum).
// Launch: blockDim = (TX, RY); logical block = RY consecutive (j,kk) rows; thread handles VEC
// consecutive x nodes per step.  Blocks are XCD-remapped so each XCD sweeps a contiguous z-range.
// Algorithmic HBM bytes: read p once, write q once = 2NT.
// =================================================================================================
template <typename T, int VEC> struct VecT;
template <> struct VecT<double, 2> { using type = double2; };
template <> struct VecT<float, 4> { using type = float4; };
template <> struct VecT<double, 1> { using type = double; };
template <> struct VecT<float, 1> { using type = float; };

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* p, T (&v)[VEC]) {
    using V = typename VecT<T, VEC>::type;
    const V t = *reinterpret_cast<const V*>(p);
    const T* tp = reinterpret_cast<const T*>(&t);
#pragma unroll
    for (int e = 0; e < VEC; e++) v[e] = tp[e];
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const T (&v)[VEC]) {
    using V = typename VecT<T, VEC>::type;
    V t;
    T* tp = reinterpret_cast<T*>(&t);
#pragma unroll
    for (int e = 0; e < VEC; e++) tp[e] = v[e];
    *reinterpret_cast<V*>(p) = t;
}

// The same operator as a register-blocked z-march (round 4): a lane owns VEC consecutive x nodes of one row and walks ZC planes, keeping the Y2 values of the
// plane below in registers; the x neighbour comes from the adjacent lane (wave shuffle; the first lane of a wave loads it), the row above (Y1[c - n]) is the
// "own" row of the workgroup (or lane group) next door: an L2 hit when both run on the same XCD, hence the XCD-contiguous block order.  Vector loads / stores,
// Y2 read once instead of twice, no per-node index arithmetic.  The one-node-per-thread kernel above read 5.0 NT from HBM for 4 NT algorithmic (rocprofv3 PMC,
// profiles/r03_pmc_traffic.json) and ran at 0.47 of the HBM peak at 512^3.
// blockDim = kBlock = LX * RB: LX lanes side by side along x (a power of two, >= ceil(n / VEC) unless the row needs several chunks), RB rows per workgroup.
// Logical blocks: x chunk fastest, then row group, then z chunk.
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void divergence_march_kernel(GridParams G, int LX, int xchunks, int rowgroups, int ZC, const T* __restrict__ Y0,
                                                                  const T* __restrict__ Y1, const T* __restrict__ Y2, T* __restrict__ b, int scrub,
                                                                  double* __restrict__ sum_partials = nullptr /* [gridDim.x]: sum of the workgroup's b (the dual solver's 1^T b) */) {
    __shared__ double red[8];
    double mysum = 0.;
    const int n = G.n;
    const size_t plane = (size_t)n * n;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int xc = (int)(lb % (unsigned)xchunks), rg = (int)((lb / (unsigned)xchunks) % (unsigned)rowgroups), zc = (int)(lb / ((unsigned)xchunks * (unsigned)rowgroups));
    const int lx = (int)threadIdx.x % LX, ry = (int)threadIdx.x / LX, RB = kBlock / LX;
    const int i0 = (xc * LX + lx) * VEC, j = rg * RB + ry;
    const int kk_lo = zc * ZC, kk_hi = min(G.nzl, kk_lo + ZC);
    const bool active = i0 < n && j < n;     // (n % VEC == 0: a vector never straddles the end of a row)
    const T ih = (T)G.inv_h;
    const bool need_left = i0 >= 1 && ((threadIdx.x & 63) == 0 || lx == 0);   // the lane to the left belongs to another wave / chunk: load the value
    T y2m[VEC];
    size_t c = (size_t)(kk_lo + 1) * plane + (size_t)(active ? j : 0) * n + (active ? i0 : 0);
    if (active) load_vec<T, VEC>(Y2 + c - plane, y2m);   // the plane below the first one (the low ghost plane for kk = 0; unused when k = 0)
    for (int kk = kk_lo; kk < kk_hi; kk++, c += plane) {
        const int k = G.k0 + kk;
        T y0[VEC], y1[VEC], y1m[VEC], y2[VEC], acc[VEC];
        if (active) {
            load_vec<T, VEC>(Y0 + c, y0);   // (non-temporal loads of Y0 / Y2 and stores of b: measured no faster, 0.85 against 0.83 ms at 512^3 fp64)
            load_vec<T, VEC>(Y1 + c, y1);
            load_vec<T, VEC>(Y2 + c, y2);
            if (j >= 1) load_vec<T, VEC>(Y1 + c - n, y1m);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; e++) y0[e] = y1[e] = y2[e] = (T)0;
        }
        T left = __shfl_up(y0[VEC - 1], 1, kWave);   // every lane of the wave takes part
        if (!active) continue;
        if (need_left) left = Y0[c - 1];
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            const int i = i0 + e;
            T a = (T)0;
            // x axis
            if (i >= 1) a += ih * (e == 0 ? left : y0[e > 0 ? e - 1 : 0]);
            if (i == n - 1) a += ih * y0[e];
            if (i < n - 1) a -= ih * y0[e];
            if (i == n - 2) a -= ih * (e + 1 < VEC ? y0[e + 1 < VEC ? e + 1 : e] : Y0[c + e + 1]);
            // y axis
            if (j >= 1) a += ih * y1m[e];
            if (j == n - 1) a += ih * y1[e];
            if (j < n - 1) a -= ih * y1[e];
            if (j == n - 2) a -= ih * Y1[c + e + n];
            // z axis
            if (k >= 1) a += ih * y2m[e];
            if (k == n - 1) a += ih * y2[e];
            if (k < n - 1) a -= ih * y2[e];
            if (k == n - 2) a -= ih * Y2[c + e + plane];
            if (scrub && !isfinite(a)) a = (T)0;
            acc[e] = a;
            mysum += (double)a;
            y2m[e] = y2[e];
        }
        store_vec<T, VEC>(b + c, acc);
    }
    if (sum_partials) {   // (wave-uniform: every lane of the workgroup arrives)
        mysum = block_sum(mysum, red);
        if (threadIdx.x == 0) sum_partials[blockIdx.x] = mysum;
    }
}

// Register-blocked z-march: a lane owns VEC consecutive x nodes of RY consecutive rows and walks ZC planes keeping the
// planes k-1, k, k+1 in registers, so every p value is loaded once per (RY x ZC) block plus its halo ((RY+2)/RY in y,
// (ZC+2)/ZC in z) instead of 5 times; the x neighbours come from the adjacent lanes (wave shuffles), only the two edge lanes
// of a wave touch memory for them.  blockDim = (TX, TYB), logical blocks ordered x-chunk fastest, then y, then z-chunk, and
// XCD-remapped so that each XCD sweeps a contiguous range of z-chunks.
constexpr int kStRY = 4;   // rows per lane

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void stencil_dot_kernel(GridParams G, int xchunks, int yblocks, int ZC /* planes per workgroup */, const T* __restrict__ p, T* __restrict__ q,
                                                             double* __restrict__ partials) {
    __shared__ double red[8];
    constexpr int RY = kStRY;
    const int n = G.n;
    const size_t plane = (size_t)n * n;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int xc = (int)(lb % (unsigned)xchunks), yb = (int)((lb / (unsigned)xchunks) % (unsigned)yblocks), zc = (int)(lb / (unsigned)(xchunks * yblocks));
    const int i = (xc * (int)blockDim.x + (int)threadIdx.x) * VEC;
    const int j0 = (yb * (int)blockDim.y + (int)threadIdx.y) * RY;
    const int kk0 = zc * ZC, kk1 = min(kk0 + ZC, G.nzl);
    const T ih2 = (T)G.inv_h2;
    const int lane = (threadIdx.y * blockDim.x + threadIdx.x) & 63;
    double acc = 0.;
    const bool active = i < n && j0 < n;
    // register planes: cur has the two y-halo rows (index 0 and RY+1)
    T prv[RY][VEC], cur[RY + 2][VEC], nxt[RY][VEC];
    auto row_ptr = [&](int kk, int j) { return p + (size_t)(kk + 1) * plane + (size_t)j * n + i; };
    auto load_row = [&](int kk, int j, T (&dst)[VEC]) {   // clamped row index = "an out-of-grid neighbour is the node itself" in y
        const int jc = min(max(j, 0), n - 1);
        if (active) load_vec<T, VEC>(row_ptr(kk, jc), dst);
    };
    if (active) {
        const int kg0 = G.k0 + kk0;
#pragma unroll
        for (int r = 0; r < RY + 2; r++) load_row(kk0, j0 - 1 + r, cur[r]);
#pragma unroll
        for (int r = 0; r < RY; r++) {
            if (kg0 > 0) load_row(kk0 - 1, j0 + r, prv[r]);
            else {
#pragma unroll
                for (int e = 0; e < VEC; e++) prv[r][e] = cur[r + 1][e];
            }
        }
    }
    for (int kk = kk0; kk < kk1; kk++) {
        const int kg = G.k0 + kk;
        if (active) {
#pragma unroll
            for (int r = 0; r < RY; r++) {
                if (kg < n - 1) load_row(kk + 1, j0 + r, nxt[r]);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; e++) nxt[r][e] = cur[r + 1][e];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RY; r++) {
            const int j = j0 + r;
            // x neighbours of the lane's first / last element: from the adjacent lanes, or from memory at the wave's edges
            T left = __shfl_up(cur[r + 1][VEC - 1], 1, kWave), right = __shfl_down(cur[r + 1][0], 1, kWave);
            if (active && j < n) {
                if (threadIdx.x == 0 || lane == 0) left = (i > 0) ? row_ptr(kk, j)[-1] : cur[r + 1][0];
                if (threadIdx.x == blockDim.x - 1 || lane == 63 || i + VEC >= n) right = (i + VEC < n) ? row_ptr(kk, j)[VEC] : cur[r + 1][VEC - 1];
                T out[VEC];
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    const T c = cur[r + 1][e];
                    const T xm = (e == 0) ? left : cur[r + 1][e - 1];
                    const T xp = (e == VEC - 1) ? right : cur[r + 1][e + 1];
                    const T s = (xp + cur[r + 2][e] + nxt[r][e] + xm + cur[r][e] + prv[r][e]) - (T)6 * c;
                    out[e] = -s * ih2;
                    acc += (double)c * (double)out[e];
                }
                store_vec<T, VEC>(q + (size_t)(kk + 1) * plane + (size_t)j * n + i, out);
            }
        }
        if (active && kk + 1 < kk1) {   // rotate the planes and fetch the y-halo rows of the new current plane
#pragma unroll
            for (int r = 0; r < RY; r++)
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    prv[r][e] = cur[r + 1][e];
                    cur[r + 1][e] = nxt[r][e];
                }
            load_row(kk + 1, j0 - 1, cur[0]);
            load_row(kk + 1, j0 + RY, cur[RY + 1]);
        }
    }
    // block reduction over blockDim.x*blockDim.y == kBlock threads
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partials[lb] = red[0] + red[1] + red[2] + red[3];
}

// L*u (not -L) for the operator-identity tests: same stencil, no reduction.
template <typename T>
__global__ __launch_bounds__(kBlock) void laplacian_kernel(GridParams G, const T* __restrict__ u, T* __restrict__ out) {
    const int n = G.n;
    const size_t plane = (size_t)n * n;
    const size_t nown = (size_t)G.nzl * plane;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nown; v += (size_t)gridDim.x * kBlock) {
        const int kk = (int)(v / plane);
        const int rem = (int)(v - (size_t)kk * plane);
        const int j = rem / n, i = rem - j * n;
        const int k = G.k0 + kk;
        const size_t c = v + plane;
        const T uc = u[c];
        const T xp = (i < n - 1) ? u[c + 1] : uc, xm = (i > 0) ? u[c - 1] : uc;
        const T yp = (j < n - 1) ? u[c + n] : uc, ym = (j > 0) ? u[c - n] : uc;
        const T zp = (k < n - 1) ? u[c + plane] : uc, zm = (k > 0) ? u[c - plane] : uc;
        out[c] = ((xp + yp + zp + xm + ym + zm) - (T)6 * uc) * (T)G.inv_h2;
    }
}

// =================================================================================================
// Device-resident CG scalars (no host round trip inside the loop)
// =================================================================================================
enum Scalar : int {
    SC_RHO0 = 0,   // ||P b||^2
    SC_RHO_A = 1,  // rho of even iterations
    SC_RHO_B = 2,  // rho of odd iterations
    SC_UW = 3,     // u.w of the current projection
    SC_SHIFT = 4,  // area-weighted mean of phi along the sources
    SC_AREA = 5,   // sum of source areas
    SC_RR = 6,     // ||P r||^2 of the current residual (convergence test)
    SC_RR0 = 7,    // ||P b||^2
    SC_RZ = 8,     // dual solver: r.z of the current iteration
    SC_SUMB = 9,   // dual solver: sum of the right-hand side b (compatibility condition sum(mu) = sum(b))
    SC_AXSUM = 10, // dual solver: sum_i (A x0)_i  (x_kkt = x0 - mean(A x0): the constant that the shift absorbs)
    SC_COUNT = 16
};
// red[] layout (the all-reduced vector): red[0] = scalar partial sum, red[1..m] = w = A r'

// Sum `np` block partials into *dst (single block; fixed order -> deterministic).
static __global__ __launch_bounds__(kBlock) void finalize_sum_kernel(const double* __restrict__ partials, int np, double* __restrict__ dst) {
    __shared__ double red[8];
    double s = 0.;
    for (int a = threadIdx.x; a < np; a += kBlock) s += partials[a];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *dst = s;
}

// x += alpha p ; r += alpha q ; partial ||r||^2     (alpha = rho / (p.Kp))
// 6NT bytes (read x,p,r,q; write x,r).
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void update_xr_kernel(size_t nvec /* owned/VEC */, size_t off /* ghost plane offset */,
                                                           const double* __restrict__ sc, int rho_slot, const double* __restrict__ pq,
                                                           T* __restrict__ x, const T* __restrict__ p, T* __restrict__ r,
                                                           const T* __restrict__ q, double* __restrict__ partials) {
    __shared__ double red[8];
    const double rho_cur = sc[rho_slot];
    const double alpha_d = rho_cur == 0. ? 0. : rho_cur / *pq;  // rho == 0: already solved, keep x (no 0/0)
    const T alpha = (T)alpha_d;
    double acc = 0.;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += (size_t)gridDim.x * kBlock) {
        const size_t c = off + v * VEC;
        T xv[VEC], pv[VEC], rv[VEC], qv[VEC];
        load_vec<T, VEC>(x + c, xv);
        load_vec<T, VEC>(p + c, pv);
        load_vec<T, VEC>(r + c, rv);
        load_vec<T, VEC>(q + c, qv);
#pragma unroll
        for (int e = 0; e < VEC; e++) {
            xv[e] += alpha * pv[e];
            rv[e] += alpha * qv[e];
            acc += (double)rv[e] * (double)rv[e];
        }
        store_vec<T, VEC>(x + c, xv);
        store_vec<T, VEC>(r + c, rv);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// partial ||v||^2 over the owned nodes (initial residual)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void norm2_kernel(size_t nvec, size_t off, const T* __restrict__ v_, double* __restrict__ partials) {
    __shared__ double red[8];
    double acc = 0.;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += (size_t)gridDim.x * kBlock) {
        T a[VEC];
        load_vec<T, VEC>(v_ + off + v * VEC, a);
#pragma unroll
        for (int e = 0; e < VEC; e++) acc += (double)a[e] * (double)a[e];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// p = -z + beta p,  beta = rho_new / rho_old.  Plain projected CG: z = r and rho_new = red[0] - u.w
// (||r' - A^T u||^2 = ||r'||^2 - u.w because A A^T u = w).  Preconditioned: z = P M^-1 r and rho_new = red[0] = r.z'
// (z' = M^-1 r before its projection; r.A^T u = 0 because A r = 0).  init!=0: p = -z.  Block 0 publishes rho_new.  3NT bytes.
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void update_p_kernel(size_t nvec, size_t off, double* __restrict__ sc, int rho_old_slot, int rho_new_slot,
                                                          const double* __restrict__ red0, int init, int use_uw, const T* __restrict__ z,
                                                          T* __restrict__ p) {
    const double rho_new = *red0 - (use_uw ? sc[SC_UW] : 0.);
    const double rho_old = sc[rho_old_slot];
    const double beta_d = (init || rho_old == 0.) ? 0. : rho_new / rho_old;
    const T beta = (T)beta_d;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += (size_t)gridDim.x * kBlock) {
        const size_t c = off + v * VEC;
        T zv[VEC], pv[VEC];
        load_vec<T, VEC>(z + c, zv);
        if (!init) load_vec<T, VEC>(p + c, pv);
#pragma unroll
        for (int e = 0; e < VEC; e++) pv[e] = init ? -zv[e] : (-zv[e] + beta * pv[e]);
        store_vec<T, VEC>(p + c, pv);
    }
    // every block has read sc[rho_old_slot] before any block of the NEXT kernel runs; the new value goes
    // to the other slot, so there is no intra-kernel race.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[rho_new_slot] = rho_new;
        if (init) sc[SC_RHO0] = rho_new;
    }
}

// =================================================================================================
// Constraint operator A (trilinearCoefficients :433-464; rows :80-98) restricted to the nodes a slab owns.
// =================================================================================================
// red[1+row] = sum_e coef * v[node]  over the slab-owned entries of every row (CSR over all m rows);
// block 0 additionally folds the scalar partials into red[0].
template <typename T>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(int m, const int* __restrict__ row_ptr, const uint32_t* __restrict__ ent_node,
                                                             const double* __restrict__ ent_coef, const T* __restrict__ v,
                                                             const double* __restrict__ partials, int np, double* __restrict__ red) {
    __shared__ double lds[8];
    if (blockIdx.x == 0) {
        double s = 0.;
        for (int a = threadIdx.x; a < np; a += kBlock) s += partials[a];
        s = block_sum(s, lds);
        if (threadIdx.x == 0) red[0] = s;
        return;
    }
    // Eight lanes per row, one entry each (a row has at most the eight corners of its cell), summed in entry order by every lane of the group.  (Rounds 1-5 walked the
    // entries in one thread: eight dependent index -> value load pairs in a row, 10 us on an idle device and 50 us beside a sweep that saturates the HBM -- the
    // projection of the stencil CG runs beside the x update and had become the longer of the two, profiles/r05_projection.txt.)
    const int gid = (blockIdx.x - 1) * kBlock + threadIdx.x, row = gid >> 3, k = gid & 7;
    double term = 0.;
    if (row < m) {
        const int e = row_ptr[row] + k;
        if (e < row_ptr[row + 1]) term = ent_coef[e] * (double)v[ent_node[e]];
    }
    const int base = (int)(threadIdx.x & 63) & ~7;
    double s = 0.;
#pragma unroll
    for (int a = 0; a < 8; a++) s += __shfl(term, base + a, 64);
    if (row < m && k == 0) red[1 + row] = s;
}

// u = Ginv * w  (Ginv = (A A^T)^-1, dense, row-major, leading dimension ld, a multiple of 4).  One workgroup per row, four
// consecutive columns per lane and load (16 or 32 bytes), U loads of the row in flight per lane at once (the kernel is pure latency
// otherwise: a row is 11-100 KB): U = 4 covers rows up to 4096 columns in one go; longer rows (round 6: rocker at 512^3, m = 12 612,
// where one load in flight per lane read the 1.27 GB of S at 3.1 TB/s) take chunks of 8192 columns with U = 8 (launch_ginv_matvec).
// TM = float: the dual solver's preconditioner reads a single-precision copy (any fixed symmetric positive definite operator will do
// there); products and sums stay in double.
template <typename TM, int U>
__global__ __launch_bounds__(kBlock) void ginv_matvec_kernel(int m, int ld, const TM* __restrict__ Ginv, const double* __restrict__ w,
                                                             double* __restrict__ u) {
    __shared__ double lds[8];
    const int row = blockIdx.x;
    const TM* g = Ginv + (size_t)row * ld;
    auto load4 = [&](int c, TM (&gv)[4]) {
        if (sizeof(TM) == 4) *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g + c);
        else {
            *reinterpret_cast<double2*>(gv) = *reinterpret_cast<const double2*>(g + c);
            *reinterpret_cast<double2*>(gv + 2) = *reinterpret_cast<const double2*>(g + c + 2);
        }
    };
    auto dot4 = [&](int c, const TM (&gv)[4]) {
        double s = (double)gv[0] * w[c];
        if (c + 1 < m) s += (double)gv[1] * w[c + 1];
        if (c + 2 < m) s += (double)gv[2] * w[c + 2];
        if (c + 3 < m) s += (double)gv[3] * w[c + 3];
        return s;
    };
    double s = 0.;
    for (int c0 = 0; c0 < m; c0 += U * kBlock * 4) {
        TM gv[U][4];
#pragma unroll
        for (int a = 0; a < U; a++) {
            const int c = c0 + (threadIdx.x + a * kBlock) * 4;
            if (c < m) load4(c, gv[a]);
        }
#pragma unroll
        for (int a = 0; a < U; a++) {
            const int c = c0 + (threadIdx.x + a * kBlock) * 4;
            if (c < m) s += dot4(c, gv[a]);
        }
    }
    s = block_sum(s, lds);
    if (threadIdx.x == 0) u[row] = s;
}
template <typename TM>
static inline void launch_ginv_matvec(hipStream_t st, int rows, int m, int ld, const TM* G, const double* w, double* u) {
    if (m <= 4 * kBlock * 4) hipLaunchKernelGGL((ginv_matvec_kernel<TM, 4>), dim3(rows), dim3(kBlock), 0, st, m, ld, G, w, u);
    else hipLaunchKernelGGL((ginv_matvec_kernel<TM, 8>), dim3(rows), dim3(kBlock), 0, st, m, ld, G, w, u);
}

// v[node] -= sum_e coef * u[row]  (node-major lists: no atomics, deterministic); block 0 also
// computes u.w into sc[SC_UW].
template <typename T>
__global__ __launch_bounds__(kBlock) void scatter_nodes_kernel(int nnodes, const uint32_t* __restrict__ node_id, const int* __restrict__ node_ptr,
                                                               const int* __restrict__ ent_row, const double* __restrict__ ent_coef,
                                                               const double* __restrict__ u, const double* __restrict__ w, int m,
                                                               double* __restrict__ sc, int save_rr /*0 no, 1 SC_RR, 2 SC_RR and SC_RR0*/,
                                                               T* __restrict__ v, int nred, double* scratch /* [nred] */, unsigned* ticket /* zero between launches */) {
    __shared__ double lds[8];
    if ((int)blockIdx.x < nred) {
        // u.w over the m rows: `nred` workgroups take contiguous chunks (one workgroup walking 12 612 rows -- rocker at 512^3 -- took 16 us of a 0.13 ms
        // projection), leave their partial sums in `scratch`, and the last one to arrive adds them in index order: the same bits whichever that is
        const int chunk = (m + nred - 1) / nred, a0 = (int)blockIdx.x * chunk, a1 = min(m, a0 + chunk);
        double s = 0.;
        for (int a = a0 + (int)threadIdx.x; a < a1; a += kBlock) s += u[a] * w[a];
        s = block_sum(s, lds);
        __shared__ unsigned last;
        if (threadIdx.x == 0) {
            scratch[blockIdx.x] = s;
            __threadfence();
            last = atomicAdd(ticket, 1u) == (unsigned)nred - 1u;
        }
        __syncthreads();
        if (!last) return;
        if (threadIdx.x == 0) {
            __threadfence();
            double tot = 0.;
            for (int b = 0; b < nred; b++) tot += __builtin_nontemporal_load(scratch + b);
            *ticket = 0u;
            sc[SC_UW] = tot;
            if (save_rr) sc[SC_RR] = w[-1] - tot;  // w = red+1: red[0] holds ||v||^2 before the projection
            if (save_rr == 2) sc[SC_RR0] = w[-1] - tot;
        }
        return;
    }
    const int t = ((int)blockIdx.x - nred) * kBlock + threadIdx.x;
    if (t >= nnodes) return;
    double s = 0.;
    for (int e = node_ptr[t]; e < node_ptr[t + 1]; e++) s += ent_coef[e] * u[ent_row[e]];
    v[node_id[t]] = (T)((double)v[node_id[t]] - s);
}

// =================================================================================================
// Shift (evaluateAverageAlongSourceGeometry :466-496 -> evaluateFunction :405-431): per source and per
// z-plane of its cell the bilinear value (x lerp then y lerp) times area*(1-tz | tz); the owner of the
// plane evaluates it, so slabs sum to the reference's nested lerp.  phi = -x here.
// =================================================================================================
struct ShiftItem {
    uint32_t node;  // local index (ghost layout) of the (i,j) corner in the plane
    float pad;
    double tx, ty, weight;  // weight = area * (1-tz) or area * tz
};

template <typename T>
__global__ __launch_bounds__(kBlock) void shift_partial_kernel(int nitems, const ShiftItem* __restrict__ items, int n, const T* __restrict__ x,
                                                               double* __restrict__ partials) {
    __shared__ double red[8];
    double acc = 0.;
    for (int a = blockIdx.x * kBlock + threadIdx.x; a < nitems; a += gridDim.x * kBlock) {
        const ShiftItem it = items[a];
        const double v00 = -(double)x[it.node], v10 = -(double)x[it.node + 1];
        const double v01 = -(double)x[it.node + n], v11 = -(double)x[it.node + n + 1];
        const double a0 = v00 * (1. - it.tx) + v10 * it.tx;
        const double a1 = v01 * (1. - it.tx) + v11 * it.tx;
        acc += it.weight * (a0 * (1. - it.ty) + a1 * it.ty);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// phi = -x - shift   (:108, :111), written as double for the caller regardless of T.
template <typename T>
__global__ __launch_bounds__(kBlock) void write_phi_kernel(size_t nown, size_t off, const T* __restrict__ x, const double* __restrict__ red0,
                                                           double area_sum, double* __restrict__ sc, T* __restrict__ phi) {
    const double shift = *red0 / area_sum;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nown; v += (size_t)gridDim.x * kBlock)
        phi[off + v] = (T)(-(double)x[off + v] - shift);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[SC_SHIFT] = shift;
}

// =================================================================================================
// fastIntegration (integrateGreedily, signed_heat_grid_solver.cpp:224-275).  The reference's FIFO BFS from node
// (0,0,0) visits the grid level by level (i+j+k) and, inside a level, in descending lexicographic (i,j,k) order, so the
// first visitor ("parent") of a node is  (i,j,k-1) if k>0, else (i,j-1,0) if j>0, else (i-1,0,0)  -- verified against the
// order-dependent host BFS in tests.  The BFS therefore equals three families of independent prefix scans:
//   row (.,0,0) along x  ->  plane k=0 along y  ->  every (i,j) column along z,
// with the reference's step  phi[q] = phi[p] + dot(normalize(Y_p + Y_q), pos(q) - pos(p))  (:245-251).
// The kernels store x = -phi (the CG convention) so that the shift / write-out kernels are shared.
// =================================================================================================
template <typename T>
__device__ __forceinline__ double bfs_step(const T* __restrict__ Y0, const T* __restrict__ Y1, const T* __restrict__ Y2, size_t p, size_t q,
                                           double e0, double e1, double e2) {
    double a0 = (double)Y0[q] + (double)Y0[p], a1 = (double)Y1[q] + (double)Y1[p], a2 = (double)Y2[q] + (double)Y2[p];
    const double nrm = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
    a0 /= nrm; a1 /= nrm; a2 /= nrm;
    return a0 * e0 + a1 * e1 + a2 * e2;
}

// plane k = 0 of the slab that owns it: thread 0 scans the row j = 0 along x, then every thread scans its column along y.
template <typename T>
__global__ __launch_bounds__(1024) void bfs_plane0_kernel(int n, double cell, double bx, double by, const T* __restrict__ Y0,
                                                          const T* __restrict__ Y1, const T* __restrict__ Y2, T* __restrict__ x) {
    const size_t off = (size_t)n * n;  // owned plane 0 sits after the low ghost plane
    if (threadIdx.x == 0) {
        double phi = 0.;
        x[off] = (T)0;
        for (int i = 1; i < n; i++) {
            const double e0 = (i * cell + bx) - ((i - 1) * cell + bx);
            phi = phi + bfs_step<T>(Y0, Y1, Y2, off + i - 1, off + i, e0, 0., 0.);
            x[off + i] = (T)(-phi);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double phi = -(double)x[off + i];
        for (int j = 1; j < n; j++) {
            const double e1 = (j * cell + by) - ((j - 1) * cell + by);
            phi = phi + bfs_step<T>(Y0, Y1, Y2, off + (size_t)(j - 1) * n + i, off + (size_t)j * n + i, 0., e1, 0.);
            x[off + (size_t)j * n + i] = (T)(-phi);
        }
    }
}

// columns along z for the owned planes; the plane below (low ghost, or owned plane 0 when the slab starts at k = 0)
// already holds x = -phi.
template <typename T>
__global__ __launch_bounds__(kBlock) void bfs_z_kernel(GridParams G, double cell, double bz, const T* __restrict__ Y0, const T* __restrict__ Y1,
                                                       const T* __restrict__ Y2, T* __restrict__ x) {
    const size_t plane = (size_t)G.n * G.n;
    const size_t ij = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (ij >= plane) return;
    const int kk0 = (G.k0 == 0) ? 1 : 0;  // first owned plane to fill (local owned index); its parent plane is kk0-1
    double phi = -(double)x[(size_t)kk0 * plane + ij];  // local plane index = owned index + 1 (ghost layout) -> parent at kk0
    for (int kk = kk0; kk < G.nzl; kk++) {
        const int k = G.k0 + kk;
        const double e2 = (k * cell + bz) - ((k - 1) * cell + bz);
        const size_t q = (size_t)(kk + 1) * plane + ij, pidx = q - plane;
        phi = phi + bfs_step<T>(Y0, Y1, Y2, pidx, q, 0., 0., e2);
        x[q] = (T)(-phi);
    }
}

// =================================================================================================
// Dual (Schur-complement) solver of the KKT system  [[L, A^T],[A, 0]] [x; mu] = [b; 0]  (signed_heat_grid_solver.cpp:101-107):
// with K = -L and K^+ its pseudo-inverse (the DCT solve):  x = K^+ (A^T mu - b) + c 1,  S mu + c 1 = A K^+ b,  1^T mu = 1^T b,
// S = A K^+ A^T (m x m, SPD).  CG on S restricted to {sum(nu) = 0}, preconditioned by G^-1 (A K A^T) G^-1 (G = A A^T):
// all CG vectors are m-dimensional; per iteration one K^+ (five DCT sweeps), one scatter A^T p, one gather A v, two dense
// G^-1 mat-vecs and one sparse (A K A^T) mat-vec.  The constant c drops out of phi after the shift (:110-111).
// The single-workgroup kernels below do the m-vector algebra with device-resident scalars.
// =================================================================================================
constexpr int kDualBlock = 1024;

__device__ __forceinline__ double block_sum_1024(double v, double* lds /* >= 17 doubles */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // protect lds reuse between consecutive calls
    if (lane == 0) lds[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.;
        for (int a = 0; a < kDualBlock / kWave; a++) s += lds[a];
        lds[16] = s;
    }
    __syncthreads();
    return lds[16];
}

// w[node] (+)= sum_e coef * v[row]   (node-major lists; accumulate = 0: assignment, the touched set is fixed so the rest of w stays 0)
template <typename T>
__global__ __launch_bounds__(kBlock) void scatter_rows_to_nodes_kernel(int nnodes, const uint32_t* __restrict__ node_id, const int* __restrict__ node_ptr,
                                                                       const int* __restrict__ ent_row, const double* __restrict__ ent_coef,
                                                                       const double* __restrict__ v, int accumulate, T* __restrict__ w, double scale = 1.0,
                                                                       T* __restrict__ saved = nullptr /* [nnodes]: the touched nodes' values before the update */) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= nnodes) return;
    double s = 0.;
    for (int e = node_ptr[t]; e < node_ptr[t + 1]; e++) s += ent_coef[e] * v[ent_row[e]];
    const T old = w[node_id[t]];
    if (saved) saved[t] = old;
    w[node_id[t]] = (T)((accumulate ? (double)old : 0.) + scale * s);
}
// w[node] = saved[...]: undoes an in-place scatter_rows_to_nodes_kernel exactly
template <typename T>
__global__ __launch_bounds__(kBlock) void restore_nodes_kernel(int nnodes, const uint32_t* __restrict__ node_id, const T* __restrict__ saved, T* __restrict__ w) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t < nnodes) w[node_id[t]] = saved[t];
}

// y = B x for the sparse m x m matrix B = A K A^T (CSR); one thread per row
static __global__ __launch_bounds__(kBlock) void csr_matvec_kernel(int m, const int* __restrict__ ptr, const int* __restrict__ col, const double* __restrict__ val,
                                                            const double* __restrict__ x, double* __restrict__ y) {
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= m) return;
    double s = 0.;
    for (int e = ptr[r]; e < ptr[r + 1]; e++) s += val[e] * x[col[e]];
    y[r] = s;
}

// out = -in over the whole ghost-layout array (right-hand side A^T mu - b is completed by a scatter-accumulate)
template <typename T>
__global__ __launch_bounds__(kBlock) void negate_kernel(size_t count, const T* __restrict__ in, T* __restrict__ out) {
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < count; v += (size_t)gridDim.x * kBlock) out[v] = -in[v];
}

// partial sum of the owned entries of v
template <typename T>
__global__ __launch_bounds__(kBlock) void sum_kernel(size_t nown, size_t off, const T* __restrict__ v, double* __restrict__ partials) {
    __shared__ double red[8];
    double acc = 0.;
    for (size_t a = (size_t)blockIdx.x * kBlock + threadIdx.x; a < nown; a += (size_t)gridDim.x * kBlock) acc += (double)v[off + a];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// init: mu = (sum b / m) 1;  (called once; sumb = all-reduced sum of b in red0)
static __global__ __launch_bounds__(kDualBlock) void dual_init_mu_kernel(int m, const double* __restrict__ sumb, double* __restrict__ mu, double* __restrict__ sc) {
    const double v = *sumb / (double)m;
    for (int a = threadIdx.x; a < m; a += kDualBlock) mu[a] = v;
    if (threadIdx.x == 0) sc[SC_SUMB] = *sumb;
}

// r = Pm(g - S mu)  with Smu given;  rr0 = r.r
static __global__ __launch_bounds__(kDualBlock) void dual_init_residual_kernel(int m, const double* __restrict__ g, const double* __restrict__ Smu,
                                                                        double* __restrict__ r, double* __restrict__ sc, int set_rr0 = 1) {
    __shared__ double lds[17];
    double s = 0.;
    for (int a = threadIdx.x; a < m; a += kDualBlock) s += g[a] - Smu[a];
    const double mean = block_sum_1024(s, lds) / (double)m;
    double rr = 0.;
    for (int a = threadIdx.x; a < m; a += kDualBlock) {
        const double v = g[a] - Smu[a] - mean;
        r[a] = v;
        rr += v * v;
    }
    rr = block_sum_1024(rr, lds);
    if (threadIdx.x == 0) {
        sc[SC_RR] = rr;
        if (set_rr0) sc[SC_RR0] = rr;
    }
}

// z = Pm(z);  rz_new = r.z;  init: p = z, else p = z + (rz_new/rz) p;  rz = rz_new
static __global__ __launch_bounds__(kDualBlock) void dual_direction_kernel(int m, int init, const double* __restrict__ r, double* __restrict__ z,
                                                                    double* __restrict__ p, double* __restrict__ sc) {
    __shared__ double lds[17];
    double s = 0.;
    for (int a = threadIdx.x; a < m; a += kDualBlock) s += z[a];
    const double mean = block_sum_1024(s, lds) / (double)m;
    double rz = 0.;
    for (int a = threadIdx.x; a < m; a += kDualBlock) {
        const double v = z[a] - mean;
        z[a] = v;
        rz += r[a] * v;
    }
    rz = block_sum_1024(rz, lds);
    const double rz_old = sc[SC_RZ];
    const double beta = (init || rz_old == 0.) ? 0. : rz / rz_old;
    for (int a = threadIdx.x; a < m; a += kDualBlock) p[a] = init ? z[a] : z[a] + beta * p[a];
    __syncthreads();
    if (threadIdx.x == 0) sc[SC_RZ] = rz;
}

// Sp = Pm(Sp);  alpha = rz / (p.Sp);  mu += alpha p;  r -= alpha Sp;  rr = r.r
static __global__ __launch_bounds__(kDualBlock) void dual_update_kernel(int m, const double* __restrict__ Sp, const double* __restrict__ p, double* __restrict__ mu,
                                                                 double* __restrict__ r, double* __restrict__ sc) {
    __shared__ double lds[17];
    double s = 0.;
    for (int a = threadIdx.x; a < m; a += kDualBlock) s += Sp[a];
    const double mean = block_sum_1024(s, lds) / (double)m;
    double pSp = 0.;
    for (int a = threadIdx.x; a < m; a += kDualBlock) pSp += p[a] * (Sp[a] - mean);
    pSp = block_sum_1024(pSp, lds);
    const double rz_cur = sc[SC_RZ];
    const double alpha = rz_cur == 0. ? 0. : rz_cur / pSp;  // r.z == 0: the system is already solved (m == 1, or an exact start) -- no 0/0
    double rr = 0.;
    for (int a = threadIdx.x; a < m; a += kDualBlock) {
        mu[a] += alpha * p[a];
        const double v = r[a] - alpha * (Sp[a] - mean);
        r[a] = v;
        rr += v * v;
    }
    rr = block_sum_1024(rr, lds);
    if (threadIdx.x == 0) sc[SC_RR] = rr;
}

// sum the per-slab reduction vectors of the slabs this process owns and write the result back to all of them
static __global__ __launch_bounds__(kBlock) void sum_slabs_kernel(int nslabs, double* const* __restrict__ bufs, int count) {
    const int a = blockIdx.x * kBlock + threadIdx.x;
    if (a >= count) return;
    double s = 0.;
    for (int t = 0; t < nslabs; t++) s += bufs[t][a];
    for (int t = 0; t < nslabs; t++) bufs[t][a] = s;
}

// dense[idx[a]] = val[a]  (non-zeros of G = A A^T, each index unique)
static __global__ __launch_bounds__(kBlock) void scatter_triplets_kernel(size_t cnt, const uint64_t* __restrict__ idx, const double* __restrict__ val,
                                                                  double* __restrict__ dense) {
    for (size_t a = (size_t)blockIdx.x * kBlock + threadIdx.x; a < cnt; a += (size_t)gridDim.x * kBlock) dense[idx[a]] = val[a];
}

template <typename TS, typename TD>
__global__ __launch_bounds__(kBlock) void convert_kernel(size_t count, const TS* __restrict__ src, TD* __restrict__ dst) {
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < count; v += (size_t)gridDim.x * kBlock) dst[v] = (TD)src[v];
}

// =================================================================================================
// Dense in-place inversion of the SPD matrix G = A A^T (m x m, padded to a multiple of 64 with an
// identity tail) by blocked Gauss-Jordan without pivoting (Schur complements of an SPD matrix stay
// SPD).  Setup-time only; replaces the reference's sparse LU of the KKT matrix (:101-107) together
// with the projected CG.  Per 64-wide pivot block kb:
//   1. P = inv(G[kb,kb])                                   (one workgroup, registers + LDS)
//   2. R = P * G[kb,:]  (row panel),  C = G[:,kb] (saved column panel)
//   3. G[i,j] -= C[i] * R[j]   for i,j outside kb           (rank-64 update on the matrix cores)
//   4. G[kb,:] = R ; G[:,kb] = -C * P ; G[kb,kb] = P
// Symmetry halves the work: between a processed block p and an unprocessed block u the running matrix satisfies
// M[p,u] = -M[u,p]^T, and M is symmetric inside the processed and inside the unprocessed set, so only the block-lower
// triangle (i >= j) is kept up to date; the panels are rebuilt from it with the right transposes and signs, the update
// visits nb(nb+1)/2 tiles, and one mirror pass at the end fills the upper triangle of the inverse.
// Large matrices (>= 64 blocks) run two-level: outer blocks of four pivot blocks whose rank-64 updates touch only the cross of tiles
// the next inner steps read, followed by ONE rank-256 update of everything else (see GjTiles).
// =================================================================================================
constexpr int kGJ = 64;

// Batched form of the blocked Gauss-Jordan kernels below (round 4: the boxes of the two-level inverse, shm_twolevel.hip.h -- one workgroup per box walking its
// matrix in global memory took 45 ms for a 280-row box; the blocked kernels do all boxes at once in 3 launches per 64 rows of the LARGEST box).  blockIdx.y = box a:
// its matrix is D + offD[a], ld = rows padded to 64 (identity on the padded diagonal), its 64 x 64 pivot scratch P + a 4096, its R / C panels at offW[a]
// (64 x ld and ld x 64).  D == nullptr: the unbatched call (one matrix, the kernel's own arguments).
struct GjBatch {
    const int* ptrI;
    const size_t* offD;
    const size_t* offW;
    double* D;
    double* P;
    double* R;
    double* C;
};
__device__ __forceinline__ int gj_batch_ld(const GjBatch& B, int a) { return (B.ptrI[a + 1] - B.ptrI[a] + kGJ - 1) / kGJ * kGJ; }

// step 1: invert the 64x64 pivot block (Gauss-Jordan, SPD -> no pivoting).  256 threads, each owning a 4x4 sub-block
// in registers; per elimination step only the pivot row and column travel through LDS (double-buffered: one barrier
// per step).
// E = elements per thread and dimension: E = 4 -> 256 threads (one wave per SIMD), E = 2 -> 1024 threads (four waves per SIMD).  The 64 elimination steps are a
// dependent chain; with one wave per SIMD every one of a step's ~95 fp64 instructions waits out its predecessor's latency, with four a quarter of the
// instructions per thread and three other waves to issue from meanwhile (round 3: 39 -> see DESIGN.md section 4 per 64 x 64 block).
// the elimination itself: r = this thread's E x E sub-block (row block ty, column block tx) of the 64 x 64 tile on entry, of its inverse on return;
// lds: 4 x 64 doubles (pivot row / column, double-buffered).  Callers that reuse `lds` afterwards put a barrier in between.
// One elimination step k = kq E + KR with KR -- the pivot's position inside a thread's E x E block -- a compile-time constant (round 4): only the block's row KR
// and column KR can be the pivot row / column, so 2 E - 1 of its E^2 elements need selects and the others are one FMA each.  With k a run-time value every element
// carried three 64-bit selects (the step was bound by its ~120 vector instructions per thread: 0.6 us); the values are the same.
template <int E, int KR>
__device__ __forceinline__ void gj_scalar_step(double (&r)[E][E], double (*rowk)[kGJ], double (*colk)[kGJ], int kq, int ty, int tx, int* __restrict__ flag) {
    const int k = kq * E + KR;
    constexpr int pb = KR & 1;   // (E is even)
    if (kq == ty) {
#pragma unroll
        for (int b = 0; b < E; b++) rowk[pb][tx * E + b] = r[KR][b];
    }
    if (kq == tx) {
#pragma unroll
        for (int a = 0; a < E; a++) colk[pb][ty * E + a] = r[a][KR];
    }
    __syncthreads();
    const double piv = rowk[pb][k];
    if (threadIdx.x == 0 && !(piv > 0.)) *flag = 1;
    // 1 / piv sits on the critical path of all 64 elimination steps: hardware reciprocal + two Newton steps (full precision for a positive,
    // normal pivot) instead of the IEEE division sequence
    double ip = __builtin_amdgcn_rcp(piv);
    ip = fma(fma(-piv, ip, 1.0), ip, ip);
    ip = fma(fma(-piv, ip, 1.0), ip, ip);
    double rv[E], cv[E];
#pragma unroll
    for (int b = 0; b < E; b++) rv[b] = rowk[pb][tx * E + b];
#pragma unroll
    for (int a = 0; a < E; a++) cv[a] = colk[pb][ty * E + a];
    const bool rowsel = ty == kq, colsel = tx == kq;
#pragma unroll
    for (int a = 0; a < E; a++)
#pragma unroll
        for (int b = 0; b < E; b++) {
            const double rowv = rv[b] * ip, colv = -cv[a] * ip;
            const double base = fma(-cv[a], rowv, r[a][b]);
            if (a == KR && b == KR) r[a][b] = rowsel ? (colsel ? ip : rowv) : (colsel ? colv : base);
            else if (a == KR) r[a][b] = rowsel ? rowv : base;
            else if (b == KR) r[a][b] = colsel ? colv : base;
            else r[a][b] = base;
        }
}
// Two elimination steps (k = kq E + KR and k + 1, KR even: both pivots lie in the same E x E blocks) behind ONE barrier (round 4).  The owners publish rows and
// columns k and k + 1 as they are BEFORE step k; every thread then forms row / column k + 1 as step k leaves them itself -- eight extra FMAs, the very
// operations their owners perform, so the values (and the inverse) are bit-identical to two single steps -- and applies both steps to its block.  Saves one
// LDS write -> barrier -> read round trip of the ~1000 cycles a single step takes.  lds: 2 buffers x 4 x 64 doubles.
template <int E, int KR>
__device__ __forceinline__ void gj_scalar_step2(double (&r)[E][E], double* __restrict__ lds, int kq, int ty, int tx, int* __restrict__ flag) {
    static_assert(KR % 2 == 0 && KR + 1 < E, "a pair of pivots inside one block");
    const int k = kq * E + KR;
    double* buf = lds + ((KR >> 1) & 1) * 4 * kGJ;
    double *rowA = buf, *rowB = buf + kGJ, *colA = buf + 2 * kGJ, *colB = buf + 3 * kGJ;
    if (kq == ty) {
#pragma unroll
        for (int b = 0; b < E; b++) {
            rowA[tx * E + b] = r[KR][b];
            rowB[tx * E + b] = r[KR + 1][b];
        }
    }
    if (kq == tx) {
#pragma unroll
        for (int a = 0; a < E; a++) {
            colA[ty * E + a] = r[a][KR];
            colB[ty * E + a] = r[a][KR + 1];
        }
    }
    __syncthreads();
    const double piv0 = rowA[k], m01 = rowA[k + 1], m10 = colA[k + 1], m11 = rowB[k + 1];
    if (threadIdx.x == 0 && !(piv0 > 0.)) *flag = 1;
    double ip0 = __builtin_amdgcn_rcp(piv0);
    ip0 = fma(fma(-piv0, ip0, 1.0), ip0, ip0);
    ip0 = fma(fma(-piv0, ip0, 1.0), ip0, ip0);
    double rv0[E], cv0[E], rv1[E], cv1[E];
#pragma unroll
    for (int b = 0; b < E; b++) {
        rv0[b] = rowA[tx * E + b];
        rv1[b] = rowB[tx * E + b];
    }
#pragma unroll
    for (int a = 0; a < E; a++) {
        cv0[a] = colA[ty * E + a];
        cv1[a] = colB[ty * E + a];
    }
    const bool rowsel = ty == kq, colsel = tx == kq;
    const double rowv0_k1 = m01 * ip0;   // M'[k][k+1]
    // row / column k + 1 after step k (what their owners hold then)
#pragma unroll
    for (int b = 0; b < E; b++) {
        const double base = fma(-m10, rv0[b] * ip0, rv1[b]);
        rv1[b] = (b == KR && colsel) ? -m10 * ip0 : base;
    }
#pragma unroll
    for (int a = 0; a < E; a++) {
        const double base = fma(-cv0[a], rowv0_k1, cv1[a]);
        cv1[a] = (a == KR && rowsel) ? rowv0_k1 : base;
    }
    const double piv1 = fma(-m10, rowv0_k1, m11);
    if (threadIdx.x == 0 && !(piv1 > 0.)) *flag = 1;
    double ip1 = __builtin_amdgcn_rcp(piv1);
    ip1 = fma(fma(-piv1, ip1, 1.0), ip1, ip1);
    ip1 = fma(fma(-piv1, ip1, 1.0), ip1, ip1);
    // step k, then step k + 1, on this thread's block (as gj_scalar_step)
#pragma unroll
    for (int a = 0; a < E; a++)
#pragma unroll
        for (int b = 0; b < E; b++) {
            const double rowv = rv0[b] * ip0, colv = -cv0[a] * ip0;
            const double base = fma(-cv0[a], rowv, r[a][b]);
            if (a == KR && b == KR) r[a][b] = rowsel ? (colsel ? ip0 : rowv) : (colsel ? colv : base);
            else if (a == KR) r[a][b] = rowsel ? rowv : base;
            else if (b == KR) r[a][b] = colsel ? colv : base;
            else r[a][b] = base;
        }
#pragma unroll
    for (int a = 0; a < E; a++)
#pragma unroll
        for (int b = 0; b < E; b++) {
            const double rowv = rv1[b] * ip1, colv = -cv1[a] * ip1;
            const double base = fma(-cv1[a], rowv, r[a][b]);
            if (a == KR + 1 && b == KR + 1) r[a][b] = rowsel ? (colsel ? ip1 : rowv) : (colsel ? colv : base);
            else if (a == KR + 1) r[a][b] = rowsel ? rowv : base;
            else if (b == KR + 1) r[a][b] = colsel ? colv : base;
            else r[a][b] = base;
        }
}
template <int E, bool PAIRS = false>
__device__ __forceinline__ void gj_invert64_scalar(double (&r)[E][E], double* __restrict__ lds, int* __restrict__ flag) {
    static_assert(E == 2 || E == 4, "block size");
    constexpr int kT = kGJ / E;   // threads per dimension
    double(*rowk)[kGJ] = reinterpret_cast<double(*)[kGJ]>(lds);
    double(*colk)[kGJ] = reinterpret_cast<double(*)[kGJ]>(lds + 2 * kGJ);
    const int ty = threadIdx.x / kT, tx = threadIdx.x % kT;
    if (PAIRS) {
        for (int kq = 0; kq < kT; kq++) {
            gj_scalar_step2<E, 0>(r, lds, kq, ty, tx, flag);
            if (E == 4) gj_scalar_step2<E, 2 % E>(r, lds, kq, ty, tx, flag);
        }
        return;
    }
    for (int kq = 0; kq < kT; kq++) {
        gj_scalar_step<E, 0>(r, rowk, colk, kq, ty, tx, flag);
        gj_scalar_step<E, 1>(r, rowk, colk, kq, ty, tx, flag);
        if (E == 4) {
            gj_scalar_step<E, 2 % E>(r, rowk, colk, kq, ty, tx, flag);
            gj_scalar_step<E, 3 % E>(r, rowk, colk, kq, ty, tx, flag);
        }
    }
}
template <int E>
__global__ __launch_bounds__((kGJ / E) * (kGJ / E)) void gj_pivot_kernel(double* __restrict__ G_, int ld, int kb, double* __restrict__ Pout_, int* __restrict__ flag, int prio,
                                                                         GjBatch Bt = GjBatch{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}) {
    constexpr int kT = kGJ / E;   // threads per dimension
    __shared__ double lds[4 * kGJ];
    if (prio) __builtin_amdgcn_s_setprio(3);   // (see gj_panels_kernel)
    double* G = G_;
    double* Pout = Pout_;
    if (Bt.D) {
        const int a = blockIdx.y;
        ld = gj_batch_ld(Bt, a);
        if (kb >= ld / kGJ) return;
        G = Bt.D + Bt.offD[a];
        Pout = Bt.P + (size_t)a * kGJ * kGJ;
    }
    const int ty = threadIdx.x / kT, tx = threadIdx.x % kT;
    const size_t o = (size_t)kb * kGJ;
    double r[E][E];
#pragma unroll
    for (int a = 0; a < E; a++)
#pragma unroll
        for (int b = 0; b < E; b++) r[a][b] = G[(o + ty * E + a) * ld + o + tx * E + b];
    gj_invert64_scalar<E>(r, lds, flag);
#pragma unroll
    for (int a = 0; a < E; a++)
#pragma unroll
        for (int b = 0; b < E; b++) Pout[(ty * E + a) * kGJ + tx * E + b] = r[a][b];
}

// The same inverse by BLOCK Gauss-Jordan with 4 x 4 pivot blocks (round 3): 16 elimination steps instead of 64.  Thread (ty, tx) of a 16 x 16 grid owns the 4 x 4
// block (ty, tx); in step q the owner of block (q, q) inverts it in registers (four unrolled scalar steps) while the threads of block row / column q publish
// their blocks through LDS; after ONE barrier every thread forms T = Binv * Rb and r <- base - Cb * T (two 4 x 4 x 4 products: 128 FMAs), with the
// operands selected branch-free so that the four cases (pivot block, pivot row, pivot column, elsewhere) run the same instructions:
//   elsewhere:  Rb = M[q, tx], Cb = M[ty, q], base = r      -> r - M[ty,q] Binv M[q,tx]
//   row q:      Rb = r,                                     -> Binv M[q,tx]                (result = T)
//   column q:   Rb = I,        Cb = r,       base = 0       -> - M[ty,q] Binv
//   (q, q):     Rb = I                                      -> Binv
// The scalar version's 64 steps each pay a barrier, an LDS round trip and a reciprocal chain (0.6 us per step: 39 us per pivot block, the longest
// link of the three-launch chain per pivot block that the constraint set-up waits for at <= 128^3).
// the elimination itself (r: this thread's 4 x 4 block (ty, tx) of the tile, replaced by the inverse's; lds: kGjBlock4Lds doubles; a caller that reuses
// `lds` afterwards puts a barrier in between)
constexpr int kGjBlock4Lds = 2 * (256 + 256 + 16);
__device__ __forceinline__ void gj_invert64_block4(double (&r)[4][4], double* __restrict__ lds, int* __restrict__ flag) {
    double(*rowb)[16][16] = reinterpret_cast<double(*)[16][16]>(lds);
    double(*colb)[16][16] = reinterpret_cast<double(*)[16][16]>(lds + 512);
    double(*binv)[16] = reinterpret_cast<double(*)[16]>(lds + 1024);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    for (int q = 0; q < 16; q++) {
        const int pb = q & 1;
        const bool in_row = ty == q, in_col = tx == q;
        if (in_row && in_col) {
            // 4 x 4 Gauss-Jordan in registers (SPD: no pivoting); a non-positive pivot raises the flag like the scalar kernel
            double B0[4][4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) B0[a][b] = r[a][b];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double piv = r[k][k];
                if (!(piv > 0.)) *flag = 1;
                double ip = __builtin_amdgcn_rcp(piv);
                ip = fma(fma(-piv, ip, 1.0), ip, ip);
                ip = fma(fma(-piv, ip, 1.0), ip, ip);
                double rowv[4], colv[4];
#pragma unroll
                for (int b = 0; b < 4; b++) rowv[b] = r[k][b] * ip;
#pragma unroll
                for (int a = 0; a < 4; a++) colv[a] = r[a][k];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (a == k) r[a][b] = (b == k) ? ip : rowv[b];
                        else r[a][b] = (b == k) ? -colv[a] * ip : fma(-colv[a], rowv[b], r[a][b]);
                    }
            }
            {   // one Newton step X <- X + X (I - B X): neighbouring source cells make near-dependent rows, and what the 4 x 4 inverse loses to them every
                // later block step would carry along (the projector test holds A P v to 1e-11)
                double E[4][4];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        double e = a == b ? 1.0 : 0.0;
#pragma unroll
                        for (int k = 0; k < 4; k++) e = fma(-B0[a][k], r[k][b], e);
                        E[a][b] = e;
                    }
                double X[4][4];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        double x = r[a][b];
#pragma unroll
                        for (int k = 0; k < 4; k++) x = fma(r[a][k], E[k][b], x);
                        X[a][b] = x;
                    }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) r[a][b] = X[a][b];
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) binv[pb][a * 4 + b] = r[a][b];
        } else if (in_row) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) rowb[pb][tx][a * 4 + b] = r[a][b];
        } else if (in_col) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) colb[pb][ty][a * 4 + b] = r[a][b];
        }
        __syncthreads();
        // T = Binv Rb, column by column (r is still intact: the pivot row's Rb is its own block), then r row by row.  The pivot column's Cb is its own
        // block -- which is exactly what it wrote to colb -- so Cb always comes from LDS and never needs a register copy; the live set is r, T and a few
        // operand rows: the kernel has to fit in the registers Step 1's waves leave free.
        double T[4][4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            double Rc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double rrow = rowb[pb][tx][k * 4 + b];   // (stale where unused: selected away)
                Rc[k] = in_col ? (k == b ? 1.0 : 0.0) : (in_row ? r[k][b] : rrow);
            }
#pragma unroll
            for (int a = 0; a < 4; a++) {
                double t = binv[pb][a * 4] * Rc[0];
#pragma unroll
                for (int k = 1; k < 4; k++) t = fma(binv[pb][a * 4 + k], Rc[k], t);
                T[a][b] = t;
            }
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
            double Cr[4];
#pragma unroll
            for (int k = 0; k < 4; k++) Cr[k] = colb[pb][ty][a * 4 + k];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                double v = in_col ? 0.0 : r[a][b];
#pragma unroll
                for (int k = 0; k < 4; k++) v = fma(-Cr[k], T[k][b], v);
                r[a][b] = in_row ? T[a][b] : v;
            }
        }
    }
}
static __global__ __launch_bounds__(256) void gj_pivot_block4_kernel(double* __restrict__ G, int ld, int kb, double* __restrict__ Pout, int* __restrict__ flag, int prio) {
    __shared__ double lds[kGjBlock4Lds];
    if (prio) __builtin_amdgcn_s_setprio(3);   // (see gj_panels_kernel)
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const size_t o = (size_t)kb * kGJ;
    double r[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) r[a][b] = G[(o + ty * 4 + a) * ld + o + tx * 4 + b];
    gj_invert64_block4(r, lds, flag);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) Pout[(ty * 4 + a) * kGJ + tx * 4 + b] = r[a][b];
}

// step 2: R[:, b] = P * G[kb, b] and C[b, :] = G[b, kb] for every block index b, from the block-lower triangle:
//   X = stored block (b >= kb ? G[b,kb] : G[kb,b]);   b > kb: G[kb,b] = X^T, G[b,kb] = X;   b < kb: G[kb,b] = X, G[b,kb] = -X^T.
typedef double gj_f64x4 __attribute__((ext_vector_type(4)));
static __global__ __launch_bounds__(kBlock) void gj_panels_kernel(const double* __restrict__ G_, int ld, int kb, const double* __restrict__ P_,
                                                           double* __restrict__ R_ /* [..][ld]: rows r_row0 .. r_row0+63 */, int r_row0,
                                                           double* __restrict__ C_ /* [ld][c_ld]: columns c_col0 .. c_col0+63 */, int c_ld, int c_col0, int prio,
                                                           GjBatch Bt = GjBatch{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}) {
    const double* G = G_;
    const double* P = P_;
    double* R = R_;
    double* C = C_;
    if (Bt.D) {
        const int a = blockIdx.y;
        ld = gj_batch_ld(Bt, a);
        if (kb >= ld / kGJ || (int)blockIdx.x >= ld / kGJ) return;
        G = Bt.D + Bt.offD[a];
        P = Bt.P + (size_t)a * kGJ * kGJ;
        R = Bt.R + Bt.offW[a];
        C = Bt.C + Bt.offW[a];
        c_ld = kGJ;
    }
    // X alone in LDS (33 KB; with P staged as well the kernel needed 65 KB and could not be placed on a CU beside two workgroups of the tiered Step-1
    // kernel).  The 64 x 64 x 64 product P X (or P X^T) runs on the matrix cores (round 3; the scalar LDS version took 23-32 us per launch, three quarters of
    // it waiting on LDS): one 32 x 32 quadrant per wave as 2 x 2 v_mfma_f64_16x16x4_f64 tiles; the A operand (P: the same 32 KB for every workgroup of
    // the launch, L2-resident) goes from global memory straight into the lanes that feed it.
    // Operand layout (one f64 per lane): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D: col = lane & 15, row = (lane >> 4) + 4 reg.
    __shared__ double x[kGJ][kGJ + 1];
    if (prio) __builtin_amdgcn_s_setprio(3);   // (prio: set by the host where Step 1 leaves the set-up little slack -- Solver::setup_prio) set-up kernels share their SIMDs with Step-1 waves: they are short and on the critical path
    const int b = blockIdx.x;
    const size_t o = (size_t)kb * kGJ, ob = (size_t)b * kGJ;
    const bool below = b >= kb;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1, l15 = lane & 15, l4 = lane >> 4;
    double pa[2][kGJ / 4];   // this lane's A operands: rows wr*32 + a*16 + l15, k = 4 s + l4
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int s4 = 0; s4 < kGJ / 4; s4++) pa[a][s4] = P[(wr * 32 + a * 16 + l15) * kGJ + 4 * s4 + l4];
    for (int t = threadIdx.x; t < kGJ * kGJ; t += kBlock) {
        const int i = t / kGJ, j = t % kGJ;
        x[i][j] = below ? G[(ob + i) * ld + o + j] : G[(o + i) * ld + ob + j];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kGJ * kGJ; t += kBlock) {
        const int i = t / kGJ, j = t % kGJ;
        C[(ob + i) * c_ld + c_col0 + j] = below ? x[i][j] : -x[j][i];
    }
    gj_f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++) acc[a][c] = gj_f64x4{0., 0., 0., 0.};
    if (b != kb) {
#pragma unroll
        for (int s4 = 0; s4 < kGJ / 4; s4++) {
            const int k = 4 * s4 + l4;
            double bf[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int j = wc * 32 + c * 16 + l15;
                bf[c] = (b > kb) ? x[j][k] : x[k][j];   // R[:, b] = P X^T below the pivot block, P X above it
            }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 2; c++) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[a][s4], bf[c], acc[a][c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = wr * 32 + a * 16 + l4 + 4 * r, j = wc * 32 + c * 16 + l15;
                // the pivot block column of R carries P: G[:,kb] = -C P, G[kb,kb] = P
                R[(size_t)(r_row0 + i) * ld + ob + j] = (b == kb) ? P[i * kGJ + j] : acc[a][c][r];
            }
}

// step 3+4: rank-64 update  G[i,j] -= C[i,:] R[:,j]  with the pivot row / column rewritten on the fly:
//   rows of the pivot block:     G[kb, j] = R[:, j]        (R's own pivot block column holds P, so G[kb,kb] = P)
//   columns of the pivot block:  G[i, kb] = -C[i,:] P
// Dense fp64 GEMM -> matrix cores: v_mfma_f64_16x16x4_f64.  64x64 output tile per workgroup, one 32x32 quadrant per wave
// (2x2 accumulators of 16x16), K = 64 in two LDS chunks of 32.  Operand layout (one f64 per lane):
//   A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],  D: col = lane&15, row = (lane>>4) + 4*reg.
constexpr int kGJK = 32;
// Which tiles of the block-lower triangle a launch updates, and with what:
//   GJ_ALL   every tile, rank 64 (pivot block kb): the single-level algorithm (small m: latency-bound, fewest launches)
//   GJ_CROSS tiles with a block index inside the current outer block [k0, k0+nO), rank 64 (inner pivot block kb): these are the only
//            tiles the next inner steps read, so they are brought up to date at once
//   GJ_REST  all other tiles, once per outer block, with the nO inner steps' panels at once: rank 64 nO.  The update is bound by the
//            read-modify-write of G (16 bytes per 2K flop): rank 256 instead of 64 lifts its ceiling from ~30 to ~120 TFLOP/s
enum GjTiles : int { GJ_ALL = 0, GJ_CROSS = 1, GJ_REST = 2 };
__device__ __forceinline__ void gj_tri_decode(unsigned t, int& bi, int& bj) {   // t -> (bi >= bj): t = bi (bi + 1) / 2 + bj
    bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((unsigned)bi * (unsigned)(bi + 1) / 2 > t) bi--;
    while ((unsigned)(bi + 1) * (unsigned)(bi + 2) / 2 <= t) bi++;
    bj = (int)(t - (unsigned)bi * (unsigned)(bi + 1) / 2);
}
// one 64 x 64 tile (bi >= bj) of the update; kb < 0: no pivot row / column among the tiles of this launch (GJ_REST).  smem: kGjUpdateLds doubles.
constexpr int kGjUpdateLds = kGJ * (kGJK + 1) + kGJK * (kGJ + 1);
__device__ __forceinline__ void gj_update_tile(double* __restrict__ G, int ld, int bi, int bj, int kb, const double* __restrict__ R, int r_row0, const double* __restrict__ C,
                                               int c_ld, int c_col0, int K, double* __restrict__ smem) {
    double(*cs)[kGJK + 1] = reinterpret_cast<double(*)[kGJK + 1]>(smem);                     // C chunk  [i][k]
    double(*rs)[kGJ + 1] = reinterpret_cast<double(*)[kGJ + 1]>(smem + kGJ * (kGJK + 1));   // R chunk  [k][j]
    const size_t oi = (size_t)bi * kGJ, oj = (size_t)bj * kGJ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    gj_f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = gj_f64x4{0., 0., 0., 0.};
    for (int kc = 0; kc < K; kc += kGJK) {
        __syncthreads();
        for (int t = threadIdx.x; t < kGJ * kGJK; t += kBlock) {
            const int k = t & (kGJK - 1), i = t >> 5;   // 32 consecutive k per row segment of C
            cs[i][k] = C[(oi + i) * c_ld + c_col0 + kc + k];
            const int j = t & (kGJ - 1), k2 = t >> 6;   // 64 consecutive j per row segment of R
            rs[k2][j] = R[(size_t)(r_row0 + kc + k2) * ld + oj + j];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kGJK; kk += 4) {
            double af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; a++) af[a] = cs[wr * 32 + a * 16 + l15][kk + l4];
#pragma unroll
            for (int b = 0; b < 2; b++) bf[b] = rs[kk + l4][wc * 32 + b * 16 + l15];
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    const bool prow = kb >= 0 && bi == kb, pcol = kb >= 0 && bj == kb;
    const size_t o = (size_t)kb * kGJ;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const size_t row = oi + wr * 32 + a * 16 + l4 + 4 * r, col = oj + wc * 32 + b * 16 + l15;
                double* dst = &G[row * ld + col];
                const double v = acc[a][b][r];
                if (prow) *dst = R[(size_t)(r_row0 + (int)(row - o)) * ld + col];
                else if (pcol) *dst = -v;
                else *dst -= v;
            }
}
template <int TILES>
__global__ __launch_bounds__(kBlock) void gj_update_kernel(double* __restrict__ G_, int ld, int nb, int kb /* inner pivot block; unused for GJ_REST */,
                                                           int k0, int nO, const double* __restrict__ R_ /* [..][ld] */, int r_row0,
                                                           const double* __restrict__ C_ /* [ld][c_ld] */, int c_ld, int c_col0, int K, int prio,
                                                           GjBatch Bt = GjBatch{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}) {
    __shared__ double smem[kGjUpdateLds];
    if (prio) __builtin_amdgcn_s_setprio(3);   // (see gj_panels_kernel)
    double* G = G_;
    const double* R = R_;
    const double* C = C_;
    if (Bt.D) {   // (GJ_ALL only)
        const int a = blockIdx.y;
        ld = gj_batch_ld(Bt, a);
        nb = ld / kGJ;
        if (kb >= nb || blockIdx.x >= (unsigned)(nb * (nb + 1) / 2)) return;
        G = Bt.D + Bt.offD[a];
        R = Bt.R + Bt.offW[a];
        C = Bt.C + Bt.offW[a];
        c_ld = kGJ;
    }
    int bi, bj;
    if (TILES == GJ_ALL) {
        gj_tri_decode(blockIdx.x, bi, bj);
    } else if (TILES == GJ_CROSS) {
        // column strip: (bi, bj = k0 + c), bi >= bj, for c < nO: indices [0, nO nb); row strip: (bi = k0 + r, bj < k0): [nO nb, nO nb + nO k0)
        const int idx = (int)blockIdx.x;
        if (idx < nO * nb) {
            bj = k0 + idx % nO;
            bi = idx / nO;
            if (bi < bj) return;
        } else {
            const int r = idx - nO * nb;
            bi = k0 + r % nO;
            bj = r / nO;
        }
    } else {
        gj_tri_decode(blockIdx.x, bi, bj);   // over the (nb - nO) blocks outside the outer block
        if (bi >= k0) bi += nO;
        if (bj >= k0) bj += nO;
    }
    gj_update_tile(G, ld, bi, bj, TILES != GJ_REST ? kb : -1, R, r_row0, C, c_ld, c_col0, K, smem);
}

// Round 4: the whole chain link of pivot block k in ONE launch.  The three launches above are a dependent chain per pivot block (pivot -> panels -> update: 80 us per
// block at m = 2842 on an idle device, and what the constraint set-up -- hence a 64^3 / 128^3 solve, or a thin slab of a multi-GPU run -- waits for).  But step k's pivot
// and panels read only the CROSS of tiles with a block index k, and the update of step k-1 rewrites that cross without reading it again at step k.  So launch k
//   * workgroups [0, nb) ("panel role", block b): form the pivot tile and the stored tile X_b of the cross with step k-1's rank-64 update applied ON THE FLY (two
//     64 x 64 x 64 products on the matrix cores, operands = the previous launch's panels, read straight from L2), invert the pivot tile -- every workgroup for itself:
//     nb redundant inversions in parallel cost nothing on the critical path and there is no flag to wait for --, and write R_k[:, b] and C_k[b] as gj_panels_kernel does;
//   * the remaining workgroups ("update role"): step k-1's update of every tile OUTSIDE the cross of k (the cross of k is rewritten by step k's own update from R_k, C_k
//     alone: rows of the pivot block = R_k, columns = -C_k P_k), exactly as gj_update_kernel<GJ_ALL>.
// The update of step k-1 thus runs beside the pivot / panels of step k: per pivot block one launch whose length is max(panel role, update role) instead of the sum of three.
// R / C alternate between two buffers (launch k reads k-1's, writes k's); launch nb is the last update alone.  Same arithmetic, operand order and results as the chain
// above.
// The pivot tile is inverted by the scalar 64-step elimination: inside this kernel's register budget it is the faster one (the 4 x 4 block steps want 140 registers).
// Resources: <= 128 registers (waves_per_eu; the compiler honours it only while the LDS lets four workgroups onto a CU) -- what two waves of the tiered Step-1
// kernel leave free on a SIMD is 512 - 2 x 184, and a 144-register build of this kernel measurably does NOT run beside them -- and 35.3 KB of LDS (one workgroup
// per CU beside two of Step 1's, like the 33.5 KB of the separate kernels): x[64][65] holds the pivot tile on its way from the accumulator layout to the 4 x 4
// blocks, then X_b (through the inversion), finally P; `scratch` serves the inversion.
#ifndef SHM_GJ_STEP_PAIRS
#define SHM_GJ_STEP_PAIRS 1   // gj_step_kernel's pivot inversion: two elimination steps per barrier (0: one, as in the separate pivot kernel -- A/B builds)
#endif
#ifndef SHM_GJ_STEP_WAVES
#define SHM_GJ_STEP_WAVES 3     // register cap of gj_step_kernel through its occupancy: 4 -> 128 registers, 21 of them spilled (rounds 4: Step 1 held 2 x 184 of a SIMD's 512
                                // and a 144-register build did not run beside it); 3 (round 5: Step 1 holds 2 x 176): 144 registers, no spill, runs beside it (wait 0.00 ms at
                                // 128^3 / 256^3, profiles/r05_gj_step.txt) -- set-up alone at m = 2842 4.05 -> 3.93 ms
#endif
constexpr int kGjStepScratch = 8 * (kGJ + 1);
constexpr int kGjStepLds = kGJ * (kGJ + 1) + kGjStepScratch;
static __global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(SHM_GJ_STEP_WAVES, SHM_GJ_STEP_WAVES))) void gj_step_kernel(
    double* __restrict__ G, int ld, int nb, int k, const double* __restrict__ Rp /* [64][ld] of step k-1 */, const double* __restrict__ Cp /* [ld][64] */,
    double* __restrict__ Rn, double* __restrict__ Cn, int* __restrict__ flag, int prio) {
    __shared__ double smem[kGjStepLds];
    static_assert(kGjStepLds >= kGjUpdateLds && kGjStepScratch >= 8 * kGJ, "LDS carve");
    if (prio) __builtin_amdgcn_s_setprio(3);   // (see gj_panels_kernel)
    const int nN = k < nb ? nb : 0;
    if ((int)blockIdx.x >= nN) {
        // ---- update role: step k-1 on every tile (bi >= bj) without a block index k
        int bi, bj;
        gj_tri_decode(blockIdx.x - (unsigned)nN, bi, bj);   // over nb - 1 blocks (k < nb) or all nb (last launch)
        if (k < nb) {
            if (bi >= k) bi += 1;
            if (bj >= k) bj += 1;
        }
        gj_update_tile(G, ld, bi, bj, k - 1, Rp, 0, Cp, kGJ, 0, kGJ, smem);
        return;
    }
    // ---- panel role
    double(*x)[kGJ + 1] = reinterpret_cast<double(*)[kGJ + 1]>(smem);
    double* scratch = smem + kGJ * (kGJ + 1);
    double(*prow)[kGJ + 1] = reinterpret_cast<double(*)[kGJ + 1]>(scratch);
    const int b = blockIdx.x;
    const bool below = b >= k;
    const int bi = below ? b : k, bj = below ? k : b;   // the stored tile of the cross
    const size_t o = (size_t)k * kGJ, ob = (size_t)b * kGJ, oi = (size_t)bi * kGJ, oj = (size_t)bj * kGJ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1, l15 = lane & 15, l4 = lane >> 4;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    double r[4][4];      // the pivot tile, thread (ty, tx) owning the 4 x 4 block (ty, tx)
    double* Cb = Cn + ob * kGJ;   // C_k[b], 64 x 64
    if (k > 0) {
        // accumulator layout: row = wr 32 + a 16 + l4 + 4 q, col = wc 32 + c 16 + l15.  Two products C[rows] R[:, cols], one after the other (together their
        // accumulators and operands do not fit the register cap), each in two groups of eight k-steps whose 32 operand loads are issued together, ahead of the
        // group's matrix instructions: left to itself the compiler waits for every k-step's loads before it issues the next ones -- 16 round trips to L2, a
        // quarter of this role's time.
        auto product = [&](gj_f64x4(&acc)[2][2], size_t orow, size_t ocol) {
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 2; c++) acc[a][c] = gj_f64x4{0., 0., 0., 0.};
#pragma unroll 1
            for (int g = 0; g < 2; g++) {
                double af[8][2], bf[8][2];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int kk = 32 * g + 4 * u + l4;
#pragma unroll
                    for (int a = 0; a < 2; a++) af[u][a] = Cp[(orow + wr * 32 + a * 16 + l15) * kGJ + kk];
#pragma unroll
                    for (int c = 0; c < 2; c++) bf[u][c] = Rp[(size_t)kk * ld + ocol + wc * 32 + c * 16 + l15];
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int c = 0; c < 2; c++) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][a], bf[u][c], acc[a][c], 0, 0, 0);
            }
        };
        {
            gj_f64x4 accp[2][2];
            product(accp, o, o);   // the pivot tile: G[k, k] - C[k] R[:, k]
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const size_t row = wr * 32 + a * 16 + l4 + 4 * q, col = wc * 32 + c * 16 + l15;
                        x[row][col] = G[(o + row) * ld + o + col] - accp[a][c][q];
                    }
        }
        gj_f64x4 accx[2][2];
        product(accx, oi, oj);     // X_b: G[bi, bj] - C[bi] R[:, bj]
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 4; c++) r[a][c] = x[ty * 4 + a][tx * 4 + c];
        __syncthreads();
        const double keep = b == k - 1 ? 0. : 1.;   // the tile (k, k-1) lies in the pivot column of step k-1: -C P, no base
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const size_t row = wr * 32 + a * 16 + l4 + 4 * q, col = wc * 32 + c * 16 + l15;
                    // (keep = 0 instead of a select: a select lets the compiler branch around each load, and it then waited for the 16 loads one by one)
                    x[row][col] = G[(oi + row) * ld + oj + col] * keep - accx[a][c][q];
                }
    } else {
        for (int t = threadIdx.x; t < kGJ * kGJ; t += kBlock) {
            const int i = t / kGJ, j = t % kGJ;
            x[i][j] = G[(oi + i) * ld + oj + j];
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 4; c++) r[a][c] = G[(o + ty * 4 + a) * ld + o + tx * 4 + c];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kGJ * kGJ; t += kBlock) {
        const int i = t / kGJ, j = t % kGJ;
        Cb[i * kGJ + j] = below ? x[i][j] : -x[j][i];
    }
    gj_invert64_scalar<4, SHM_GJ_STEP_PAIRS != 0>(r, scratch, flag);
    if (b == k) {   // the pivot block's own R column carries P
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 4; c++) Rn[(size_t)(ty * 4 + a) * ld + ob + tx * 4 + c] = r[a][c];
        return;
    }
    // P: registers -> scratch, 8 rows at a time -> the lanes that feed it to the matrix cores (rows wr 32 + a 16 + l15, k = 4 s + l4)
    double pa[2][kGJ / 4];
#pragma unroll
    for (int pass = 0; pass < 8; pass++) {
        __syncthreads();
        if ((ty >> 1) == pass) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int c = 0; c < 4; c++) prow[(ty & 1) * 4 + a][tx * 4 + c] = r[a][c];
        }
        __syncthreads();
        if (wr == (pass >> 2) && (l15 >> 3) == (pass & 1)) {
#pragma unroll
            for (int s4 = 0; s4 < kGJ / 4; s4++) pa[(pass >> 1) & 1][s4] = prow[l15 & 7][4 * s4 + l4];
        }
    }
    gj_f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++) acc[a][c] = gj_f64x4{0., 0., 0., 0.};
#pragma unroll
    for (int s4 = 0; s4 < kGJ / 4; s4++) {
        const int kk = 4 * s4 + l4;
        double bf[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int j = wc * 32 + c * 16 + l15;
            bf[c] = (b > k) ? x[j][kk] : x[kk][j];   // R[:, b] = P X^T below the pivot block, P X above it
        }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int c = 0; c < 2; c++) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[a][s4], bf[c], acc[a][c], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int q = 0; q < 4; q++) Rn[(size_t)(wr * 32 + a * 16 + l4 + 4 * q) * ld + ob + wc * 32 + c * 16 + l15] = acc[a][c][q];
}

// after the last pivot block: G[bj, bi] = G[bi, bj]^T for bi > bj (the inverse is symmetric; only its block-lower triangle was kept)
static __global__ __launch_bounds__(kBlock) void gj_mirror_kernel(double* __restrict__ G_, int ld, GjBatch Bt = GjBatch{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}) {
    __shared__ double x[kGJ][kGJ + 1];
    double* G = G_;
    if (Bt.D) {
        const int a = blockIdx.y;
        ld = gj_batch_ld(Bt, a);
        const int nb = ld / kGJ;
        if (blockIdx.x >= (unsigned)(nb * (nb - 1) / 2)) return;
        G = Bt.D + Bt.offD[a];
    }
    const unsigned t = blockIdx.x;   // strictly lower tile: bi >= 1, bj < bi
    int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((unsigned)bi * (unsigned)(bi + 1) / 2 > t) bi--;
    while ((unsigned)(bi + 1) * (unsigned)(bi + 2) / 2 <= t) bi++;
    const int bj = (int)(t - (unsigned)bi * (unsigned)(bi + 1) / 2);
    bi += 1;
    const size_t oi = (size_t)bi * kGJ, oj = (size_t)bj * kGJ;
    for (int e = threadIdx.x; e < kGJ * kGJ; e += kBlock) {
        const int i = e / kGJ, j = e % kGJ;
        x[i][j] = G[(oi + i) * ld + oj + j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kGJ * kGJ; e += kBlock) {
        const int i = e / kGJ, j = e % kGJ;
        G[(oj + i) * ld + oi + j] = x[j][i];
    }
}

// =================================================================================================
// Isosurface of the grid phi (headless replacement of the demo's Polyscope marching cubes, src/main.cpp:116-128,167-191):
// marching tetrahedra on the Kuhn split of every cell (six tets around the 000-111 diagonal; translation invariant, so
// faces of neighbouring cells agree and the surface is watertight).  inside = phi < iso.  Pass 1 counts, pass 2 appends
// triangles (three positions + three global edge keys for welding + a sort key that makes the final order deterministic).
// =================================================================================================
struct IsoParams {
    int n, nzl, k0;          // cells (i,j,k): i,j < n-1, k in [k0, min(k0+nzl, n-1))
    double bbox_min[3];
    double cell;
    double iso;
};

__device__ __constant__ int kKuhnTets[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};

// number of triangles of one tet given the inside mask of its four vertices
__device__ __forceinline__ int tet_tris(int mask) {
    const int c = __popc(mask);
    return c == 0 || c == 4 ? 0 : (c == 2 ? 2 : 1);
}

template <typename T, bool EMIT>
__global__ __launch_bounds__(kBlock) void iso_kernel(IsoParams P, const T* __restrict__ phi /* ghost layout */, unsigned long long* __restrict__ counter,
                                                     double* __restrict__ tri_pos /* [cap][9] */, unsigned long long* __restrict__ tri_keys /* [cap][3] */,
                                                     unsigned long long* __restrict__ tri_sort /* [cap] */, unsigned long long cap) {
    __shared__ double red[8];
    const int n = P.n, nc = n - 1;
    const size_t plane = (size_t)n * n;
    const int kend = min(P.nzl, n - 1 - P.k0);                 // local cell layers
    const size_t ncells = (size_t)nc * nc * (size_t)max(kend, 0);
    double mycount = 0.;
    for (size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x; c < ncells; c += (size_t)gridDim.x * kBlock) {
        const int kk = (int)(c / ((size_t)nc * nc));
        const int rem = (int)(c - (size_t)kk * nc * nc);
        const int j = rem / nc, i = rem - j * nc;
        const size_t base = (size_t)(kk + 1) * plane + (size_t)j * n + i;
        double v[8];
        int inside = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            v[q] = (double)phi[base + (q & 1) + ((q >> 1) & 1) * (size_t)n + ((q >> 2) & 1) * plane];
            inside |= (v[q] < P.iso ? 1 : 0) << q;
        }
        if (inside == 0 || inside == 255) continue;
        const int k = P.k0 + kk;
        const unsigned long long gnode = (unsigned long long)i + (unsigned long long)j * n + (unsigned long long)k * plane;
        for (int t = 0; t < 6; t++) {
            int tv[4], mask = 0;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                tv[a] = kKuhnTets[t][a];
                mask |= ((inside >> tv[a]) & 1) << a;
            }
            const int nt = tet_tris(mask);
            if (nt == 0) continue;
            if (!EMIT) {
                mycount += nt;
                continue;
            }
            // order the tet vertices: inside ones first
            int in_v[4], out_v[4], ni = 0, no = 0;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if ((mask >> a) & 1) in_v[ni++] = tv[a];
                else out_v[no++] = tv[a];
            }
            auto corner_pos = [&](int q, double* p) {
                p[0] = (i + (q & 1)) * P.cell + P.bbox_min[0];
                p[1] = (j + ((q >> 1) & 1)) * P.cell + P.bbox_min[1];
                p[2] = (k + ((q >> 2) & 1)) * P.cell + P.bbox_min[2];
            };
            auto edge_point = [&](int qa, int qb, double* p, unsigned long long& key) {   // qa inside, qb outside
                double pa[3], pb[3];
                corner_pos(qa, pa);
                corner_pos(qb, pb);
                const double tt = (P.iso - v[qa]) / (v[qb] - v[qa]);
#pragma unroll
                for (int a = 0; a < 3; a++) p[a] = pa[a] + tt * (pb[a] - pa[a]);
                const unsigned long long ga = gnode + (qa & 1) + ((qa >> 1) & 1) * (unsigned long long)n + ((qa >> 2) & 1) * plane;
                const unsigned long long gb = gnode + (qb & 1) + ((qb >> 1) & 1) * (unsigned long long)n + ((qb >> 2) & 1) * plane;
                key = ga < gb ? (ga << 32 | gb) : (gb << 32 | ga);
            };
            double pts[4][3];
            unsigned long long keys[4];
            int np = 0;
            if (ni == 1) for (int a = 0; a < 3; a++) { edge_point(in_v[0], out_v[a], pts[np], keys[np]); np++; }
            else if (ni == 3) for (int a = 0; a < 3; a++) { edge_point(in_v[a], out_v[0], pts[np], keys[np]); np++; }
            else {  // quad: (i0,o0) (i0,o1) (i1,o1) (i1,o0)
                edge_point(in_v[0], out_v[0], pts[0], keys[0]);
                edge_point(in_v[0], out_v[1], pts[1], keys[1]);
                edge_point(in_v[1], out_v[1], pts[2], keys[2]);
                edge_point(in_v[1], out_v[0], pts[3], keys[3]);
                np = 4;
            }
            // orientation: normals point from the inside vertices towards the outside ones (increasing phi)
            double cin[3] = {0, 0, 0}, cout[3] = {0, 0, 0};
            for (int a = 0; a < ni; a++) { double p[3]; corner_pos(in_v[a], p); for (int b = 0; b < 3; b++) cin[b] += p[b] / ni; }
            for (int a = 0; a < no; a++) { double p[3]; corner_pos(out_v[a], p); for (int b = 0; b < 3; b++) cout[b] += p[b] / no; }
            const int tris[2][3] = {{0, 1, 2}, {0, 2, 3}};
            for (int tr = 0; tr < nt; tr++) {
                int a0 = tris[tr][0], a1 = tris[tr][1], a2 = tris[tr][2];
                const double e1[3] = {pts[a1][0] - pts[a0][0], pts[a1][1] - pts[a0][1], pts[a1][2] - pts[a0][2]};
                const double e2[3] = {pts[a2][0] - pts[a0][0], pts[a2][1] - pts[a0][1], pts[a2][2] - pts[a0][2]};
                const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
                if (nx * (cout[0] - cin[0]) + ny * (cout[1] - cin[1]) + nz * (cout[2] - cin[2]) < 0.) { const int sw = a1; a1 = a2; a2 = sw; }
                const unsigned long long slot = atomicAdd(counter, 1ULL);
                if (slot < cap) {
                    const int ord[3] = {a0, a1, a2};
                    for (int a = 0; a < 3; a++) {
                        for (int b = 0; b < 3; b++) tri_pos[slot * 9 + a * 3 + b] = pts[ord[a]][b];
                        tri_keys[slot * 3 + a] = keys[ord[a]];
                    }
                    tri_sort[slot] = gnode * 16ULL + (unsigned long long)(t * 2 + tr);
                }
            }
        }
    }
    if (!EMIT) {
        mycount = block_sum(mycount, red);
        if (threadIdx.x == 0 && mycount > 0.) atomicAdd(counter, (unsigned long long)mycount);
    }
}

// Marching cubes on the same cells (what the demo's contour runs, src/main.cpp:121-124, through Polyscope): the case table of shm_mc_table.h (generated by
// tools/gen_mc_table.py: ambiguous faces resolved by the face's own flags, so the surface is watertight), one surface vertex per cut grid edge by linear
// interpolation from the edge's lower node, triangles oriented towards increasing phi.  Same two passes, buffers, welding keys and sort keys as iso_kernel.
template <typename T, bool EMIT>
__global__ __launch_bounds__(kBlock) void iso_mc_kernel(IsoParams P, const T* __restrict__ phi /* ghost layout */, unsigned long long* __restrict__ counter,
                                                        double* __restrict__ tri_pos /* [cap][9] */, unsigned long long* __restrict__ tri_keys /* [cap][3] */,
                                                        unsigned long long* __restrict__ tri_sort /* [cap] */, unsigned long long cap) {
    __shared__ double red[8];
    const int n = P.n, nc = n - 1;
    const size_t plane = (size_t)n * n;
    const int kend = min(P.nzl, n - 1 - P.k0);
    const size_t ncells = (size_t)nc * nc * (size_t)max(kend, 0);
    double mycount = 0.;
    for (size_t c = (size_t)blockIdx.x * kBlock + threadIdx.x; c < ncells; c += (size_t)gridDim.x * kBlock) {
        const int kk = (int)(c / ((size_t)nc * nc));
        const int rem = (int)(c - (size_t)kk * nc * nc);
        const int j = rem / nc, i = rem - j * nc;
        const size_t base = (size_t)(kk + 1) * plane + (size_t)j * n + i;
        double v[8];
        int inside = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            v[q] = (double)phi[base + (q & 1) + ((q >> 1) & 1) * (size_t)n + ((q >> 2) & 1) * plane];
            inside |= (v[q] < P.iso ? 1 : 0) << q;
        }
        const int nt = kMcCount[inside];
        if (nt == 0) continue;
        if (!EMIT) {
            mycount += nt;
            continue;
        }
        const int k = P.k0 + kk;
        const unsigned long long gnode = (unsigned long long)i + (unsigned long long)j * n + (unsigned long long)k * plane;
        for (int tr = 0; tr < nt; tr++) {
            const unsigned long long slot = atomicAdd(counter, 1ULL);
            if (slot >= cap) continue;
            for (int a = 0; a < 3; a++) {
                const int e = kMcTris[inside][3 * tr + a];
                const int qa = kMcEdge[e][0], qb = kMcEdge[e][1];   // qa < qb: the lower node of the grid edge first, whichever side is inside
                double va = v[0], vb = v[0];
#pragma unroll
                for (int q = 1; q < 8; q++) {   // (selects instead of a dynamically indexed register array)
                    va = q == qa ? v[q] : va;
                    vb = q == qb ? v[q] : vb;
                }
                const double tt = (P.iso - va) / (vb - va);
                const int ax = qa ^ qb;   // 1, 2 or 4: the edge's axis
                const double pa[3] = {(i + (qa & 1)) * P.cell + P.bbox_min[0], (j + ((qa >> 1) & 1)) * P.cell + P.bbox_min[1], (k + ((qa >> 2) & 1)) * P.cell + P.bbox_min[2]};
                tri_pos[slot * 9 + a * 3 + 0] = pa[0] + (ax == 1 ? tt * P.cell : 0.);
                tri_pos[slot * 9 + a * 3 + 1] = pa[1] + (ax == 2 ? tt * P.cell : 0.);
                tri_pos[slot * 9 + a * 3 + 2] = pa[2] + (ax == 4 ? tt * P.cell : 0.);
                const unsigned long long ga = gnode + (qa & 1) + ((qa >> 1) & 1) * (unsigned long long)n + ((qa >> 2) & 1) * plane;
                const unsigned long long gb = gnode + (qb & 1) + ((qb >> 1) & 1) * (unsigned long long)n + ((qb >> 2) & 1) * plane;
                tri_keys[slot * 3 + a] = ga << 32 | gb;
            }
            tri_sort[slot] = gnode * 16ULL + (unsigned long long)tr;
        }
    }
    if (!EMIT) {
        mycount = block_sum(mycount, red);
        if (threadIdx.x == 0 && mycount > 0.) atomicAdd(counter, (unsigned long long)mycount);
    }
}

}  // namespace shm
