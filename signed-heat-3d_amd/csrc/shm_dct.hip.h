// Fast Poisson preconditioner for the projected CG: M^-1 = C3^T D C3, the exact pseudo-inverse of the
// reference's 7-point Neumann Laplacian K = -L (laplacian(): signed_heat_grid_solver.cpp:278-334), which the
// 3-D DCT-II diagonalises (eigenvalues (2-2cos(pi k/n))/h^2 per axis).  Used sandwiched between two
// constraint projections, P M^-1 P (SURVEY 7.3 "preconditioning"): the answer of the KKT system is unchanged,
// only the iteration count drops (~25-30x at 128^3..512^3).
//
// One kernel template does a batch of 1-D transforms along any axis, entirely in LDS:
//   tile = 16 real lines (two real lines share one complex FFT), 256 threads, transform length n = 2^LOG2N fixed at
//   compile time; Stockham radix-16/8/4 passes with the butterflies in registers (shm_fft_core.h): 2 LDS round trips
//   at n <= 256, 3 at n = 512/1024, instead of log2(n).
//   DCT-II  = Makhoul permutation on load -> FFT -> twiddle e^{-i pi k/2n}             (mode FWD)
//   DCT-III = pair pre-twiddle -> inverse FFT -> inverse permutation on store          (mode INV)
//   FUSED   = FWD, spectral scaling D(kx,ky,kz), INV without leaving LDS (the z axis)
// Each sweep reads and writes every element once: x-fwd, y-fwd, z-fused, y-inv, x-inv (+ r.z partial) = 5 sweeps,
// algorithmic traffic 3T + 8TP bytes per node.  Global accesses: XPASS lines are contiguous (2 KB per wave); y/z
// lines are gathered as 16 consecutive x per element row = one 128-byte line per 16 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shm_fft_core.h"
#include "shm_kernels.hip.h"

// build-time variants (defaults = the measured best; tools/dct_variants.sh rebuilds the others for A/B runs)
#ifndef SHM_DCT_HOIST
#define SHM_DCT_HOIST 0        // 1: spectral-step constants of a thread fetched once into registers
#endif
#ifndef SHM_DCT_WAVES_HINT
#define SHM_DCT_WAVES_HINT 9   // smallest log2(n) at which the register allocator is held to the occupancy the LDS footprint allows
#endif                         // (n = 512: 268 registers without it = one tile per CU instead of two, 2.18 vs 1.62 ms per dual iteration)
#ifndef SHM_DCT_WAVES_F32
#define SHM_DCT_WAVES_F32 2     // the same for fp32 tiles (half the LDS, about two thirds of the registers)
#endif
#ifndef SHM_DCT_SPECTRAL_ORIG
#define SHM_DCT_SPECTRAL_ORIG 0
#endif
#ifndef SHM_DCT_RUNTIME_GATE
#define SHM_DCT_RUNTIME_GATE 1   // the phases of a tile (load, FFT, spectral step, store) sit behind run-time conditions on DctParams::debug_skip
#endif                         // (always 0 in product builds).  Opaque to the compiler, they stop it from hoisting one phase's loads across the
                               // previous phase, which costs ~60 registers and with them a tile of occupancy at n = 512 (fp32: 1.82 vs 2.18 ms
                               // per dual iteration at 512^3; fp64 the same without the occupancy hint below)
#ifndef SHM_DCT_WAVES_256
#define SHM_DCT_WAVES_256 0     // > 0: occupancy asked of the register allocator for n <= 256 as well
#endif
#ifndef SHM_DCT_WAVE_FFT
#define SHM_DCT_WAVE_FFT 0x100  // bit log2(n): FFT passes of that length are wave-local (a wave owns whole lines: no workgroup barriers).
#endif                         // Measured: n = 256 4 % faster per dual iteration (dense sweeps +8 %), n = 512 14 % slower -> 256 only
#ifndef SHM_DCT_LC9_F64
#define SHM_DCT_LC9_F64 4      // complex lines per tile at n = 512, fp64 (4: 1.51 ms per dual iteration at 512^3, 8: 1.62)
#endif
#ifndef SHM_DCT_LC10_F32
#define SHM_DCT_LC10_F32 4     // complex lines per tile at n = 1024, fp32 (fp64: 4, eight would not fit the LDS)
#endif
#ifndef SHM_DCT_LC9_F32
#define SHM_DCT_LC9_F32 8      // ... fp32 (8: 1.83, 4: 1.89)
#endif
#ifndef SHM_DCT_LC8
#define SHM_DCT_LC8 4          // complex lines per tile at n = 256 (4: 0.224 ms per dual iteration, 8: 0.234)
#endif
#ifndef SHM_DCT_TW_LDS_MAX
#define SHM_DCT_TW_LDS_MAX 8   // largest log2(n) whose twiddle table is copied to LDS
#endif
#ifndef SHM_DCT_POW_ALWAYS
#define SHM_DCT_POW_ALWAYS 1   // 1: inter-pass twiddles from one table read + multiplication tree at every n (else only when the table is global)
#endif

namespace shm {

#if defined(SHM_EXPERIMENT_KNOBS) || SHM_DCT_RUNTIME_GATE
#define SHM_DCT_DBG(P, bit) ((P).debug_skip & (bit))
#else
#define SHM_DCT_DBG(P, bit) 0
#endif

enum DctMode : int { DCT_FWD = 0, DCT_INV = 1, DCT_FUSED = 2 };
// complex lines per tile: 8 (16 real lines = one 128-byte row per element row); 4 at n = 1024, where an fp64 tile of 8 would take
// 147 KB of LDS (one workgroup per CU).  (Measured at n = 512: 4 instead of 8 lines doubles the residency but halves the access
// granularity to 64 bytes -- no net change, so 512 keeps the full 128-byte rows.)
template <int LOG2N, int TB /* sizeof(real) */> constexpr int dct_lc() {
    return LOG2N >= 10 ? (TB == 8 ? 4 : SHM_DCT_LC10_F32) : (LOG2N == 8 ? SHM_DCT_LC8 : (LOG2N == 9 ? (TB == 8 ? SHM_DCT_LC9_F64 : SHM_DCT_LC9_F32) : 8));
}
constexpr int dct_lines_for(int log2n, int tb) {
    return log2n >= 10 ? (tb == 8 ? 8 : 2 * SHM_DCT_LC10_F32) : (log2n == 8 ? 2 * SHM_DCT_LC8 : (log2n == 9 ? 2 * (tb == 8 ? SHM_DCT_LC9_F64 : SHM_DCT_LC9_F32) : 16));
}

// Address of element k of line l of tile t:
//   off + (t % tiles_a) a_stride + (t / tiles_a) b_stride + l line_stride + (k >> seg_shift) seg_stride + (k & seg_mask) elem_stride
// The segment term expresses the packed all-to-all layout of the multi-slab y sweeps ([dest slab][z][y_local][x]);
// plain layouts use seg_shift = 30 (k < 2^30 -> no segment).
struct DctAddr {
    long long off;          // ghost-plane offset of the CG vectors
    long long a_stride, b_stride;
    long long line_stride;  // distance between consecutive lines of a tile
    long long elem_stride;  // distance between consecutive elements of a line (inside a segment)
    long long seg_stride;
    int seg_shift, seg_mask;
};
struct DctParams {
    int ntiles;             // tiles of this sweep; the workgroups stride over them (persistent: a workgroup costs more to start than a tile to set up)
    int tiles_a;
    int debug_skip;         // always 0 in product builds (see SHM_DCT_RUNTIME_GATE); -DSHM_EXPERIMENT_KNOBS builds take it from env SHM_DCT_SKIP for
                            // timing experiments: 1 = no FFT passes, 2 = no spectral step, 4 = no global loads, 8 = no global stores
    DctAddr in, out;
    // FUSED only: spectral coordinates of line l of tile t: kx = (t % tiles_a)*16 + l ; ky = ky0 + t / tiles_a
    int ky0;
    double inv_n3_8;        // 8/n^3 (product of the three 2/n normalisations; the k=0 halvings are applied per axis)
};
__device__ __forceinline__ long long dct_addr(const DctAddr& A, long long base, int l, int k) {
    return A.off + base + (long long)l * A.line_stride + (long long)(k >> A.seg_shift) * A.seg_stride + (long long)(k & A.seg_mask) * A.elem_stride;
}

// Twiddles live in LDS up to n = 256; from n = 512 on they are read from global memory (L1/L2 resident, 8-16 KB) so that
// two fp64 tiles (73.8 KB each at n = 512) fit in one CU's 160 KB and the load/FFT/store phases of two blocks overlap.
template <int LOG2N> constexpr bool dct_tw_in_lds() { return LOG2N <= SHM_DCT_TW_LDS_MAX; }
template <int LOG2N, int TB> constexpr size_t dct_lds_bytes() {
    return ((size_t)(1 << LOG2N) * (dct_lc<LOG2N, TB>() + 1) + (dct_tw_in_lds<LOG2N>() ? (size_t)(1 << LOG2N) : 0)) * (size_t)(2 * TB) + 64;
}

// PF (round 6: LDS-DMA double buffer): the staging copy of the NEXT tile's input, in the input's natural order, behind the FFT buffer
template <int LOG2N, int TB> constexpr size_t dct_stage_bytes() { return (size_t)(1 << LOG2N) * (size_t)(2 * dct_lc<LOG2N, TB>()) * (size_t)TB; }

// One Stockham pass over the tile: every thread holds its work items in registers across the barrier.
template <typename TP, int LOG2N, int R, int NS, int SIGN>
__device__ __forceinline__ void dct_fft_pass(Cplx<TP>* buf, const Cplx<TP>* tw, int tid) {
    constexpr int LC = dct_lc<LOG2N, (int)sizeof(TP)>();
    constexpr int items = PassGeom<LOG2N, R, LC>::items;
    constexpr int IPT = (items + kBlock - 1) / kBlock;
    Cplx<TP> v[IPT][R];
#pragma unroll
    for (int a = 0; a < IPT; a++) {
        const int w = tid + a * kBlock;
        if (items % kBlock == 0 || w < items) pass_load<TP, LOG2N, R, NS, SIGN, LC, SHM_DCT_POW_ALWAYS || !dct_tw_in_lds<LOG2N>()>(buf, tw, w, v[a]);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < IPT; a++) {
        const int w = tid + a * kBlock;
        if (items % kBlock == 0 || w < items) pass_store<TP, LOG2N, R, NS, LC>(buf, w, v[a]);
    }
    __syncthreads();
}

template <typename TP, int LOG2N, int SIGN>
__device__ __forceinline__ void dct_fft(Cplx<TP>* buf, const Cplx<TP>* tw, int tid) {
    typedef FftPlan<LOG2N> P;
    dct_fft_pass<TP, LOG2N, P::R0, 1, SIGN>(buf, tw, tid);
    if constexpr (P::npass > 1) dct_fft_pass<TP, LOG2N, P::R1, P::R0, SIGN>(buf, tw, tid);
    if constexpr (P::npass > 2) dct_fft_pass<TP, LOG2N, P::R2, P::R0 * P::R1, SIGN>(buf, tw, tid);
    if constexpr (P::npass > 3) dct_fft_pass<TP, LOG2N, P::R3, P::R0 * P::R1 * P::R2, SIGN>(buf, tw, tid);
}

// ---- wave-local variant (n >= 256).  Every wave owns LC/4 complete lines of the tile: the butterflies of a line read and write only
// that line, the LDS executes one wave's instructions in order, and all reads of a pass are issued before its writes -- so a pass
// needs no workgroup barrier at all (only a compiler fence), the four waves drift apart and hide each other's LDS latency, and a
// tile is left with the two barriers around its global load / store phases.  Radix-4 passes keep all 64 lanes busy (n/4 >= 64).
template <int LOG2N> struct FftPlanWave;
template <> struct FftPlanWave<8> { static constexpr int npass = 4; static constexpr int R0 = 4, R1 = 4, R2 = 4, R3 = 4; };
#ifndef SHM_DCT_WAVE_PLAN9
#define SHM_DCT_WAVE_PLAN9 0
#endif
#if SHM_DCT_WAVE_PLAN9
template <> struct FftPlanWave<9> { static constexpr int npass = 3; static constexpr int R0 = 8, R1 = 8, R2 = 8, R3 = 1; };
#else
template <> struct FftPlanWave<9> { static constexpr int npass = 4; static constexpr int R0 = 8, R1 = 4, R2 = 4, R3 = 4; };
#endif
template <> struct FftPlanWave<10> { static constexpr int npass = 4; static constexpr int R0 = 16, R1 = 4, R2 = 4, R3 = 4; };
template <int LOG2N> constexpr bool dct_wave_fft() { return ((SHM_DCT_WAVE_FFT >> LOG2N) & 1) && LOG2N >= 8 && LOG2N <= 10; }

template <typename TP, int LOG2N, int R, int NS, int SIGN>
__device__ __forceinline__ void dct_fft_pass_wave(Cplx<TP>* buf, const Cplx<TP>* tw, int tid) {
    constexpr int LC = dct_lc<LOG2N, (int)sizeof(TP)>(), n = 1 << LOG2N;
    constexpr int B = n / R;                      // butterflies per line
    constexpr int LW = LC / (kBlock / kWave);     // lines per wave
    constexpr int IPL = (B + kWave - 1) / kWave;  // butterflies per lane and line
    static_assert(LW >= 1 && B % kWave == 0, "wave-local FFT: whole lines per wave, full waves of butterflies");
    const int wave = tid >> 6, lane = tid & 63;
    Cplx<TP> v[LW * IPL][R];
#pragma unroll
    for (int a = 0; a < LW * IPL; a++) {
        const int c = wave + (kBlock / kWave) * (a / IPL), jj = lane + kWave * (a % IPL);
        pass_load<TP, LOG2N, R, NS, SIGN, LC, SHM_DCT_POW_ALWAYS || !dct_tw_in_lds<LOG2N>()>(buf, tw, jj * LC + c, v[a]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int a = 0; a < LW * IPL; a++) {
        const int c = wave + (kBlock / kWave) * (a / IPL), jj = lane + kWave * (a % IPL);
        pass_store<TP, LOG2N, R, NS, LC>(buf, jj * LC + c, v[a]);
    }
    __builtin_amdgcn_wave_barrier();
}

template <typename TP, int LOG2N, int SIGN>
__device__ __forceinline__ void dct_fft_wave(Cplx<TP>* buf, const Cplx<TP>* tw, int tid) {
    typedef FftPlanWave<LOG2N> P;
    dct_fft_pass_wave<TP, LOG2N, P::R0, 1, SIGN>(buf, tw, tid);
    dct_fft_pass_wave<TP, LOG2N, P::R1, P::R0, SIGN>(buf, tw, tid);
    dct_fft_pass_wave<TP, LOG2N, P::R2, P::R0 * P::R1, SIGN>(buf, tw, tid);
    if constexpr (P::npass > 3) dct_fft_pass_wave<TP, LOG2N, P::R3, P::R0 * P::R1 * P::R2, SIGN>(buf, tw, tid);
}

// tw_g[t] = e^{-2 pi i t/n}, t<n ; om_g[k] = e^{-i pi k/(2n)}, k<n ; lam_g[k] = (2-2cos(pi k/n))/h^2
// Occupancy the register allocator is held to for the long transforms: two waves per SIMD (= two tiles per CU) when the LDS footprint
// allows them.  Without it the n = 512 fp64 kernels land just above 256 registers, i.e. one tile per CU; asking for more than two
// (fp32 tiles are half the size) makes the allocator spill.
#ifndef SHM_DCT_SEG9_ONE_WAVE
#define SHM_DCT_SEG9_ONE_WAVE 1
#endif
template <int LOG2N, int CPLX_BYTES> constexpr int dct_waves_per_simd() {
    constexpr int by_lds = (int)((size_t)(160 * 1024) / dct_lds_bytes<LOG2N, CPLX_BYTES / 2>());
    constexpr int want = LOG2N <= 8 ? (SHM_DCT_WAVES_256 > 0 ? SHM_DCT_WAVES_256 : 1) : (CPLX_BYTES == 8 ? SHM_DCT_WAVES_F32 : 2);
    return by_lds >= want ? want : (by_lds >= 1 ? by_lds : 1);
}
// PF (dense sweeps of the long transforms).  A workgroup alternates load -> FFT -> store; the two or three workgroups a CU holds are out of phase, but each has its 16-32 KB
// in flight only during its own load phase: ~40 KB per CU on average where 8 TB/s at ~2 us of loaded latency needs ~60 -- the sweeps sat at 0.45 of the HBM peak at 512^3
// (fused z: 0.28) from round 2 to round 5.  Round 4 prefetched the next tile into REGISTERS and lost a workgroup of occupancy to them (slower).  Round 6: the next tile's input
// travels global -> LDS directly (global_load_lds_dwordx4, no registers) into a staging region in the input's NATURAL order -- issued right after the current tile has been
// copied out of it, in flight under this tile's FFT passes and stores --, and a permute-copy LDS -> LDS puts it into the padded, Makhoul-ordered FFT rows (the layout the DMA
// cannot write).  One more LDS pass per tile (1/7 of the fused sweep's), 2 x the tile in LDS (73.5 KB at n = 512 fp64: two workgroups per CU, each with a tile in flight
// all the time).  Dense sweeps only (no tile list, no element mask, plain layout, unit stride between the lines of a y / z tile).
// MEASURED SLOWER (profiles/r06_dct_dma_rejected.txt: 512^3 fp64 0.46 -> 0.43 of the HBM peak, fp32 0.44 -> 0.36): the sweeps are bound by their LDS passes and barriers at the
// occupancy the LDS allows, not by bytes in flight.  Kept as a verified A/B variant (-DSHM_DCT_PF), not shipped.
template <typename TP, typename TIn, typename TOut, int MODE, bool DOT, int LOG2N, bool XPASS, bool SEG, bool PF = false>
__global__ __launch_bounds__(kBlock, ((LOG2N >= SHM_DCT_WAVES_HINT || SHM_DCT_WAVES_256 > 0) && !(SHM_DCT_SEG9_ONE_WAVE && SEG && LOG2N == 9 && sizeof(TP) == 8) ? dct_waves_per_simd<LOG2N, (int)sizeof(Cplx<TP>)>() : 1)) void dct_lines_kernel(DctParams P, const TIn* __restrict__ in, TOut* __restrict__ out,
                                                           const Cplx<TP>* __restrict__ tw_g, const Cplx<TP>* __restrict__ om_g,
                                                           const TP* __restrict__ lam_g, const TOut* __restrict__ dot_with,
                                                           double* __restrict__ partials, const int* __restrict__ tile_list /* nullptr: all tiles */,
                                                           const unsigned* __restrict__ elem_mask /* nullptr: all elements; else bit k = element k is non-zero on input / needed on output */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LC = dct_lc<LOG2N, (int)sizeof(TP)>(), kFftRow = LC + 1, kFftLC = LC;
    constexpr int n = 1 << LOG2N, L = 2 * LC, LOG2L = ilog2(L), total = n * L;
    constexpr int EPT = (total + kBlock - 1) / kBlock;  // elements per thread (n/16; 1 at n = 16)
    constexpr int CH = EPT < 16 ? EPT : 16;              // register chunk of the global <-> LDS copies
    Cplx<TP>* buf = reinterpret_cast<Cplx<TP>*>(smem);   // [n][kFftRow]
    Cplx<TP>* tw_l = buf + (size_t)n * kFftRow;           // [n] (only when dct_tw_in_lds)
    double* red = reinterpret_cast<double*>(tw_l + (dct_tw_in_lds<LOG2N>() ? n : 0));  // [8] block-reduction scratch
    const Cplx<TP>* tw = dct_tw_in_lds<LOG2N>() ? tw_l : tw_g;
    const int tid = threadIdx.x;
    if (dct_tw_in_lds<LOG2N>())
        for (int a = tid; a < n; a += kBlock) tw_l[a] = tw_g[a];
    // spectral-step constants of this thread (tile independent), fetched up front so that their latency hides behind the tile's loads
    constexpr int kSpStep = kBlock / kFftLC;                          // k distance between a thread's consecutive pairs
    constexpr int kSpIters = (n / 2 + 1 + kSpStep - 1) / kSpStep;
    Cplx<TP> sp_om[(MODE == DCT_FWD || !SHM_DCT_HOIST) ? 1 : kSpIters];
    TP sp_lam[(MODE == DCT_FUSED && SHM_DCT_HOIST) ? kSpIters : 1];
    TP sp_lam_n = (TP)0;
    if (MODE != DCT_FWD && SHM_DCT_HOIST) {
#pragma unroll
        for (int i = 0; i < kSpIters; i++) {
            const int k = min((tid >> ilog2(kFftLC)) + i * kSpStep, n / 2);
            sp_om[i] = om_g[k];
            if (MODE == DCT_FUSED) sp_lam[i] = lam_g[k];
        }
    }
    if (MODE == DCT_FUSED) sp_lam_n = (TP)2 * lam_g[n / 2];   // lam_n = 4/h^2 = 2 lam_{n/2}
    static_assert(!PF || !SEG, "the prefetching variant serves plain layouts");
    static_assert(!PF || (sizeof(TIn) == sizeof(TP) && (total * sizeof(TIn)) % (16 * kBlock) == 0), "the prefetching variant moves whole 16-byte chunks of same-width input");
    // PF: the staging region (input order: y / z sweeps [element][line], x sweep [line][element] -- in both, element idx of the copy loops below sits at stage[idx])
    TIn* stage = reinterpret_cast<TIn*>(smem + dct_lds_bytes<LOG2N, (int)sizeof(TP)>());
    auto pf_issue = [&](int lbx) {
        if constexpr (PF) {
            constexpr int kChunkElems = 16 / (int)sizeof(TIn);                       // elements per 16-byte chunk
            constexpr int kChunks = total / kChunkElems, kPer = kChunks / kBlock;      // chunks per tile / per thread
            const int tx = lbx;
            const long long bin = P.in.off + (long long)(tx % P.tiles_a) * P.in.a_stride + (long long)(tx / P.tiles_a) * P.in.b_stride;
            const int wave0 = (tid >> 6) << 6;
#pragma unroll
            for (int a = 0; a < kPer; a++) {
                const int q = a * kBlock + tid;                                          // this lane's chunk; the wave's 64 chunks land at consecutive 16-byte slots
                long long g;
                if (XPASS) {
                    constexpr int cpl = n / kChunkElems;                                 // chunks per line
                    g = (long long)(q / cpl) * P.in.line_stride + (long long)((q % cpl) * kChunkElems);
                } else {
                    constexpr int cpr = L / kChunkElems;                                 // chunks per element row (L consecutive lines = L consecutive x)
                    g = (long long)(q / cpr) * P.in.elem_stride + (long long)((q % cpr) * kChunkElems);
                }
                // (the source as const char*: with the typed pointer in this dependent context the compiler silently emitted no host stub for the instantiation -- hipcc 7.2)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(in + (bin + g)), reinterpret_cast<char*>(stage) + (size_t)(a * kBlock + wave0) * 16, 16, 0, 0);
            }
        }
    };
    if (PF) {
        const int lb0 = (int)xcd_remap(blockIdx.x, gridDim.x);
        if (lb0 < P.ntiles) pf_issue(lb0);
    }
    // concurrently running workgroups take neighbouring tiles (adjacent x chunks of the same rows share an XCD's L2)
    for (int lb = (int)xcd_remap(blockIdx.x, gridDim.x); lb < P.ntiles; lb += (int)gridDim.x) {
    __syncthreads();                                       // the previous tile's LDS reads are done (and the twiddle table is in place)
    const int t = tile_list ? tile_list[lb] : lb;          // sparse sweeps visit only the tiles that can hold non-zeros / needed outputs
    const long long base_in = (long long)(t % P.tiles_a) * P.in.a_stride + (long long)(t / P.tiles_a) * P.in.b_stride;
    const long long base_out = (long long)(t % P.tiles_a) * P.out.a_stride + (long long)(t / P.tiles_a) * P.out.b_stride;

    // element index of the a-th element this thread moves: idx = tid + a*256
    //   y/z sweeps: l = idx & (L-1) (consecutive lanes = consecutive lines = consecutive x), j = idx >> log2(L)
    //   x sweep   : j = idx & (n-1) (consecutive lanes = consecutive x),                  l = idx >> LOG2N
    auto line_of = [&](int idx) { return XPASS ? (idx >> LOG2N) : (idx & (L - 1)); };
    auto elem_of = [&](int idx) { return XPASS ? (idx & (n - 1)) : (idx >> LOG2L); };
    // Plain layouts (SEG = false): the address of element a of this thread splits into a per-thread 32-bit offset (fixed) and a
    // workgroup-uniform part that depends on a only -- scalar registers, no per-element vector address arithmetic:
    //   y/z sweeps: l = tid % L, j = tid / L + a (256 / L)                  -> uniform part a (256 / L) elem_stride
    //   x sweep, n >= 256: l = a / (n/256), j = tid + (a % (n/256)) 256     -> l line_stride + (a % (n/256)) 256 elem_stride
    //   x sweep, n <  256: l = tid / n + a (256/n), j = tid % n             -> a (256/n) line_stride
    auto thread_off = [&](const DctAddr& A) -> unsigned {
        if (XPASS) return n >= kBlock ? (unsigned)(tid * A.elem_stride) : (unsigned)((tid >> LOG2N) * A.line_stride + (tid & (n - 1)) * A.elem_stride);
        return (unsigned)((tid & (L - 1)) * A.line_stride + (tid >> LOG2L) * A.elem_stride);
    };
    auto uniform_off = [&](const DctAddr& A, int a) -> long long {
        if (XPASS) {
            constexpr int per = n >= kBlock ? n / kBlock : 1, lp = n >= kBlock ? 1 : kBlock / n;
            return n >= kBlock ? (long long)(a / per) * A.line_stride + (long long)((a % per) * kBlock) * A.elem_stride : (long long)(a * lp) * A.line_stride;
        }
        return (long long)(a * (kBlock / L)) * A.elem_stride;
    };
    const unsigned toff_in = SEG ? 0u : thread_off(P.in), toff_out = SEG ? 0u : thread_off(P.out);
    auto in_at = [&](const TIn* base, int a, int idx) -> const TIn* {
        return SEG ? base + dct_addr(P.in, base_in, line_of(idx), elem_of(idx)) : (base + (P.in.off + base_in + uniform_off(P.in, a))) + toff_in;
    };
    auto out_at = [&](auto* base, int a, int idx) {
        return SEG ? base + dct_addr(P.out, base_out, line_of(idx), elem_of(idx)) : (base + (P.out.off + base_out + uniform_off(P.out, a))) + toff_out;
    };

    // ---------------- load: global -> registers (CH loads in flight) -> LDS ----------------
    if constexpr (PF) {
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this thread's chunks of the tile have landed in the staging region (and its stores of the previous tile have left)
        __syncthreads();                       // ... and everyone else's
#pragma unroll
        for (int a0 = 0; a0 < EPT; a0 += CH) {
            TP v[CH];
#pragma unroll
            for (int a = 0; a < CH; a++) v[a] = (TP)stage[tid + (a0 + a) * kBlock];
#pragma unroll
            for (int a = 0; a < CH; a++) {
                const int idx = tid + (a0 + a) * kBlock;
                const int l = line_of(idx), j = elem_of(idx);
                const int slot = (MODE == DCT_INV) ? j : makhoul_slot(j, n);
                reinterpret_cast<TP*>(&buf[slot * kFftRow + (l >> 1)])[l & 1] = v[a];
            }
        }
        __syncthreads();                       // the staging region has been read out: the next tile may land in it (the barrier below the load phase would do for buf alone)
        if (lb + (int)gridDim.x < P.ntiles) pf_issue(lb + (int)gridDim.x);   // in flight under this tile's FFT passes and stores
    } else
#pragma unroll 1
    for (int a0 = 0; a0 < EPT; a0 += CH) {
        TP v[CH];
#pragma unroll
        for (int a = 0; a < CH; a++) {
            const int idx = tid + (a0 + a) * kBlock;
            if (total % kBlock == 0 || idx < total) {
                const int j = elem_of(idx);
                v[a] = (!SHM_DCT_DBG(P, 4) && (!elem_mask || ((elem_mask[j >> 5] >> (j & 31)) & 1u))) ? (TP)*in_at(in, a0 + a, idx) : (TP)0;
            }
        }
#pragma unroll
        for (int a = 0; a < CH; a++) {
            const int idx = tid + (a0 + a) * kBlock;
            if (total % kBlock == 0 || idx < total) {
                const int l = line_of(idx), j = elem_of(idx);
                // FWD/FUSED: Makhoul slot v[pj(j)] = x[j]; INV: raw spectrum X[k] in natural order; a/b lines side by side
                const int slot = (MODE == DCT_INV) ? j : makhoul_slot(j, n);
                reinterpret_cast<TP*>(&buf[slot * kFftRow + (l >> 1)])[l & 1] = v[a];
            }
        }
    }
    __syncthreads();

    constexpr bool kWaveFft = dct_wave_fft<LOG2N>();
    if ((MODE == DCT_FWD || MODE == DCT_FUSED) && !SHM_DCT_DBG(P, 1)) {
        if constexpr (kWaveFft) dct_fft_wave<TP, LOG2N, -1>(buf, tw, tid);
        else dct_fft<TP, LOG2N, -1>(buf, tw, tid);
    }
    if (kWaveFft && MODE == DCT_FWD) __syncthreads();   // the output phase below reads lines other waves transformed

    if (MODE == DCT_FWD) {
        // ---------------- X[k] = Re(om_k V[k]) straight to global ----------------
#pragma unroll 1
        for (int a0 = 0; a0 < EPT; a0 += CH) {
#pragma unroll
            for (int a = 0; a < CH; a++) {
                const int idx = tid + (a0 + a) * kBlock;
                if (total % kBlock == 0 || idx < total) {
                    const int l = line_of(idx), k = elem_of(idx), c = l >> 1;
                    TP xa, xb;
                    dct_fwd_post<TP>(buf[k * kFftRow + c], buf[((n - k) & (n - 1)) * kFftRow + c], om_g[k], xa, xb);
                    *out_at(out, a0 + a, idx) = (TOut)((l & 1) ? xb : xa);
                }
            }
        }
        continue;
    }

#if SHM_DCT_SPECTRAL_ORIG
    // ---------------- spectral step on (k, n-k) pairs, k = 0..n/2 ----------------
    if (!SHM_DCT_DBG(P, 2)) {
        const int kx0 = (t % P.tiles_a) * L, ky = P.ky0 + t / P.tiles_a;
        constexpr int pairs = (n / 2 + 1) * kFftLC;
        for (int b = tid; b < pairs; b += kBlock) {
            const int c = b & (kFftLC - 1), k = b >> ilog2(kFftLC);
            const int nk = (n - k) & (n - 1);
            const Cplx<TP> omk = om_g[k], omn = om_g[nk];
            TP xa_k, xb_k, xa_n, xb_n;
            if (MODE == DCT_FUSED) {
                const Cplx<TP> zk = buf[k * kFftRow + c], znk = buf[nk * kFftRow + c];
                dct_fwd_post<TP>(zk, znk, omk, xa_k, xb_k);
                dct_fwd_post<TP>(znk, zk, omn, xa_n, xb_n);
                // D(kx,ky,kz) = sx sy sz / (lam_x + lam_y + lam_z), s_0 = 1/n, s_k = 2/n ; zero mode -> 0
                const int kxa = kx0 + 2 * c, kxb = kxa + 1;
                const TP lxy_a = lam_g[kxa] + lam_g[ky], lxy_b = lam_g[kxb] + lam_g[ky];
                const TP sy = ky == 0 ? (TP)0.5 : (TP)1;
                const TP sa = (kxa == 0 ? (TP)0.5 : (TP)1) * sy * (TP)P.inv_n3_8, sb = sy * (TP)P.inv_n3_8;  // kxb >= 1
                const TP lk = lam_g[k], ln = lam_g[nk];
                const TP szk = k == 0 ? (TP)0.5 : (TP)1;  // nk == 0 only together with k == 0
                const TP da_k = lxy_a + lk, db_k = lxy_b + lk, da_n = lxy_a + ln, db_n = lxy_b + ln;
                // 1/d by the hardware seed and two Newton steps (1-2 ulp; the IEEE division sequence is ~10x the instructions)
                auto recip = [](TP d) {
                    TP r = t_rcp<TP>(d);
                    r = fma(fma(-d, r, (TP)1), r, r);
                    r = fma(fma(-d, r, (TP)1), r, r);
                    return r;
                };
                xa_k = da_k > (TP)0 ? xa_k * (sa * szk) * recip(da_k) : (TP)0;
                xb_k = xb_k * (sb * szk) * recip(db_k);
                xa_n = da_n > (TP)0 ? xa_n * (sa * szk) * recip(da_n) : (TP)0;
                xb_n = xb_n * (sb * szk) * recip(db_n);
            } else {
                const Cplx<TP> rk = buf[k * kFftRow + c], rn = buf[nk * kFftRow + c];
                xa_k = rk.x; xb_k = rk.y; xa_n = rn.x; xb_n = rn.y;
            }
            buf[k * kFftRow + c] = dct_inv_pre<TP>(k, xa_k, xa_n, xb_k, xb_n, omk);
            if (nk != k) buf[nk * kFftRow + c] = dct_inv_pre<TP>(nk, xa_n, xa_k, xb_n, xb_k, omn);
        }
        __syncthreads();
    }

#else
    // ---------------- spectral step on (k, n-k) pairs, k = 0..n/2 ----------------
    // A thread's pairs sit at k = tid / LC + i * (256 / LC): its om_k and lam_k were fetched before the tile's loads (sp_om, sp_lam);
    // the partners follow from om_{n-k} = -i conj(om_k) and lam_{n-k} = lam_n - lam_k (n - k >= n/2: no cancellation).
    if (!SHM_DCT_DBG(P, 2)) {
        const int kx0 = (t % P.tiles_a) * L, ky = P.ky0 + t / P.tiles_a;
        // wave-local FFTs: a wave post-processes its own lines (kSpLines of them, pairs k = lane + 64 i), no barrier on either side
        constexpr int kSpLines = kWaveFft ? kFftLC / (kBlock / kWave) : 1;
#pragma unroll
        for (int cl = 0; cl < kSpLines; cl++) {
        const int c = kWaveFft ? (tid >> 6) + (kBlock / kWave) * cl : (tid & (kFftLC - 1));
        const int kxa = kx0 + 2 * c, kxb = kxa + 1;
        TP lxy_a = (TP)0, lxy_b = (TP)0, sa = (TP)0, sb = (TP)0;
        if (MODE == DCT_FUSED) {
            // D(kx,ky,kz) = sx sy sz / (lam_x + lam_y + lam_z), s_0 = 1/n, s_k = 2/n ; zero mode -> 0
            const TP lky = lam_g[ky];
            lxy_a = lam_g[kxa] + lky;
            lxy_b = lam_g[kxb] + lky;
            const TP sy = ky == 0 ? (TP)0.5 : (TP)1;
            sa = (kxa == 0 ? (TP)0.5 : (TP)1) * sy * (TP)P.inv_n3_8;
            sb = sy * (TP)P.inv_n3_8;  // kxb >= 1
        }
#pragma unroll 1   // fully unrolled, the table reads cost > 100 registers at n = 512 and with them half the occupancy
        for (int i = 0; i < (kWaveFft ? (n / 2 + kWave) / kWave : kSpIters); i++) {
            const int k = kWaveFft ? (tid & 63) + i * kWave : (tid >> ilog2(kFftLC)) + i * kSpStep;
            if (k > n / 2) break;
            const int nk = (n - k) & (n - 1);
            const Cplx<TP> omk = SHM_DCT_HOIST ? sp_om[SHM_DCT_HOIST ? i : 0] : om_g[k];
            const Cplx<TP> omn = k == 0 ? omk : Cplx<TP>{-omk.y, -omk.x};
            TP xa_k, xb_k, xa_n, xb_n;
            if (MODE == DCT_FUSED) {
                const Cplx<TP> zk = buf[k * kFftRow + c], znk = buf[nk * kFftRow + c];
                dct_fwd_post<TP>(zk, znk, omk, xa_k, xb_k);
                dct_fwd_post<TP>(znk, zk, omn, xa_n, xb_n);
                const TP lk = SHM_DCT_HOIST ? sp_lam[SHM_DCT_HOIST ? i : 0] : lam_g[k], ln = k == 0 ? lk : sp_lam_n - lk;
                const TP szk = k == 0 ? (TP)0.5 : (TP)1;  // nk == 0 only together with k == 0
                const TP da_k = lxy_a + lk, db_k = lxy_b + lk, da_n = lxy_a + ln, db_n = lxy_b + ln;
                // 1/d by the hardware seed and two Newton steps (1-2 ulp; the IEEE division sequence is ~10x the instructions)
                auto recip = [](TP d) {
                    TP r = t_rcp<TP>(d);
                    r = fma(fma(-d, r, (TP)1), r, r);
                    r = fma(fma(-d, r, (TP)1), r, r);
                    return r;
                };
                xa_k = da_k > (TP)0 ? xa_k * (sa * szk) * recip(da_k) : (TP)0;
                xb_k = xb_k * (sb * szk) * recip(db_k);
                xa_n = da_n > (TP)0 ? xa_n * (sa * szk) * recip(da_n) : (TP)0;
                xb_n = xb_n * (sb * szk) * recip(db_n);
            } else {
                const Cplx<TP> rk = buf[k * kFftRow + c], rn = buf[nk * kFftRow + c];
                xa_k = rk.x; xb_k = rk.y; xa_n = rn.x; xb_n = rn.y;
            }
            buf[k * kFftRow + c] = dct_inv_pre<TP>(k, xa_k, xa_n, xb_k, xb_n, omk);
            if (nk != k) buf[nk * kFftRow + c] = dct_inv_pre<TP>(nk, xa_n, xa_k, xb_n, xb_k, omn);
        }
        }  // lines of this wave
        if (!kWaveFft) __syncthreads();
    }

#endif
    if (!SHM_DCT_DBG(P, 1)) {
        if constexpr (kWaveFft) dct_fft_wave<TP, LOG2N, +1>(buf, tw, tid);
        else dct_fft<TP, LOG2N, +1>(buf, tw, tid);
    }
    if (kWaveFft) __syncthreads();   // the store phase reads lines other waves transformed

    // ---------------- store: x[j] = v[makhoul_slot(j)] ----------------
    double acc = 0.;
#pragma unroll 1
    for (int a0 = 0; a0 < EPT; a0 += CH) {
        TOut dv[CH];
        if (DOT) {
#pragma unroll
            for (int a = 0; a < CH; a++) {
                const int idx = tid + (a0 + a) * kBlock;
                if (total % kBlock == 0 || idx < total)
                    dv[a] = *out_at(dot_with, a0 + a, idx);
            }
        }
#pragma unroll
        for (int a = 0; a < CH; a++) {
            const int idx = tid + (a0 + a) * kBlock;
            if (total % kBlock == 0 || idx < total) {
                const int l = line_of(idx), j = elem_of(idx);
                const TP v = reinterpret_cast<const TP*>(&buf[makhoul_slot(j, n) * kFftRow + (l >> 1)])[l & 1];
                if (!SHM_DCT_DBG(P, 8) && (!elem_mask || ((elem_mask[j >> 5] >> (j & 31)) & 1u))) *out_at(out, a0 + a, idx) = (TOut)v;
                if (DOT) acc += (double)v * (double)dv[a];
            }
        }
    }
    if (DOT) {
        acc = block_sum(acc, red);
        if (tid == 0) partials[t] = acc;
    }
    }  // tile loop
}

// =================================================================================================
// z step of the SPARSE fast Poisson solve (one iteration of the dual solver: input non-zero only on the "active" z-planes, output
// needed only there).  After the x and y transforms the operator is, for every (kx, ky) line,
//       u = s_x s_y (d I + K_z)^-1 f ,   d = lam_kx + lam_ky ,   K_z = 1-D Neumann Laplacian / h^2 ,
// which dct_lines_kernel<DCT_FUSED> evaluates as DCT -> divide -> inverse DCT on ALL n planes of ALL n^2 lines.  The constant-coefficient
// tridiagonal system has the closed-form Green's function C r^|z-j| (r < 1 the small root of a r^2 - (d + 2a) r + a = 0, a = 1/h^2,
// C = r / (a (1 - r^2))); the Neumann ends are half-sample mirrors, i.e. the images of f at -1-j and 2n-1-j repeated with period 2n.
// Summing the images gives two decaying recursions plus two boundary terms:
//       u_z = C [ F_z + G_z + Fm r^(z+1) + Gn r^(n-1-z) ] ,
//       F_z = sum_{j<=z} f_j r^(z-j) (ascending) ,   G_z = sum_{j>z} f_j r^(j-z) (descending) ,
//       Fm = q (A + r^n B) ,  Gn = r q (B + r^n A) ,  A = sum f_j r^j ,  B = sum f_j r^(n-1-j) ,  q = 1 / (1 - r^2n) ,
// every power with a non-negative exponent (no overflow; underflow is harmless).  Only the active planes are visited: one thread per
// line, two passes over its <= n_act values, consecutive threads = consecutive kx (coalesced).  The (0, 0) line (d = 0: singular)
// gets the pseudo-inverse of K_z by two prefix sums, which is what zeroing the (0,0,0) mode does in the transform version.
// Arithmetic in double whatever the storage type.  Replaces ~55 % of the per-iteration transform time at 256^3 (79 of 140 us).
// =================================================================================================
// The (0, 0) line of the sparse z step: d = 0, K_z singular -> pseudo-inverse (what zeroing the (0,0,0) mode does in the transform
// version).  f~ = f - mean f ; flux g_z = -sum_{j<=z} f~_j ; u_z = (1/a) sum_{j<z} g_j ; u -= mean u : two prefix sums over the n planes,
// done by ONE extra workgroup of the same launch (a single thread walking the planes with dependent loads cost ~100 us, a launch of its
// own 8 us): a thread owns kZsPer consecutive planes, the 256 thread totals are scanned in LDS (Hillis-Steele).
constexpr int kZsPer = 4;   // n <= 1024 = 256 threads x 4 planes
__device__ __forceinline__ double zs_block_exclusive(double v, double* buf /* [2][256] */, double& total) {
    const int t = threadIdx.x;
    int cur = 0;
    buf[t] = v;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        const double x = buf[cur * kBlock + t] + (t >= off ? buf[cur * kBlock + t - off] : 0.);
        buf[(cur ^ 1) * kBlock + t] = x;
        cur ^= 1;
        __syncthreads();
    }
    const double incl = buf[cur * kBlock + t];
    total = buf[cur * kBlock + kBlock - 1];
    __syncthreads();
    return incl - v;
}
template <typename T>
__device__ void zsolve_zero_line(int n, int n_act, const int* __restrict__ planes, double inv_h2, const T* __restrict__ in, T* __restrict__ out) {
    __shared__ double fz[kBlock * kZsPer], buf[2 * kBlock];
    const int t = threadIdx.x;
    const size_t plane = (size_t)n * n;
    for (int a = t; a < kBlock * kZsPer; a += kBlock) fz[a] = 0.;
    __syncthreads();
    for (int a = t; a < n_act; a += kBlock) fz[planes[a]] = (double)in[(size_t)planes[a] * plane];
    __syncthreads();
    double f[kZsPer], loc = 0., tot = 0.;
#pragma unroll
    for (int e = 0; e < kZsPer; e++) {
        const int z = t * kZsPer + e;
        f[e] = z < n ? fz[z] : 0.;
        loc += f[e];
    }
    zs_block_exclusive(loc, buf, tot);
    const double fmean = tot / n;
    // g_z = -sum_{j<=z} (f_j - fmean)
    loc = 0.;
#pragma unroll
    for (int e = 0; e < kZsPer; e++) {
        const int z = t * kZsPer + e;
        f[e] = z < n ? f[e] - fmean : 0.;
        loc += f[e];
    }
    double before = zs_block_exclusive(loc, buf, tot);
    double g[kZsPer];
#pragma unroll
    for (int e = 0; e < kZsPer; e++) {
        before += f[e];
        g[e] = -before;
    }
    // u_z = (1/a) sum_{j<z} g_j
    loc = 0.;
#pragma unroll
    for (int e = 0; e < kZsPer; e++) loc += (t * kZsPer + e < n) ? g[e] : 0.;
    before = zs_block_exclusive(loc, buf, tot);
    double u[kZsPer], usum = 0.;
#pragma unroll
    for (int e = 0; e < kZsPer; e++) {
        const int z = t * kZsPer + e;
        u[e] = before / inv_h2;
        before += z < n ? g[e] : 0.;
        usum += z < n ? u[e] : 0.;
    }
    zs_block_exclusive(usum, buf, tot);
    const double umean = tot / n;
    const double scale = 0.25 * 4.0 / ((double)n * (double)n);   // s_x s_y with kx = ky = 0
#pragma unroll
    for (int e = 0; e < kZsPer; e++) fz[t * kZsPer + e] = u[e] - umean;
    __syncthreads();
    for (int a = t; a < n_act; a += kBlock) out[(size_t)planes[a] * plane] = (T)(scale * fz[planes[a]]);
}

constexpr int kZsChunk = 16;
template <typename T>
__global__ __launch_bounds__(kBlock) void zsolve_sparse_kernel(int n, int n_act, const int* __restrict__ planes /* ascending */, const double* __restrict__ lam /* [n] */,
                                                               double inv_h2, const T* __restrict__ in, T* __restrict__ out) {
    const size_t plane = (size_t)n * n;
    if (blockIdx.x == gridDim.x - 1) {   // one extra workgroup: the singular (0, 0) line
        zsolve_zero_line<T>(n, n_act, planes, inv_h2, in, out);
        return;
    }
    const size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= plane) return;
    const int kx = (int)(gid % (size_t)n), ky = (int)(gid / (size_t)n);
    const double scale = (kx == 0 ? 0.5 : 1.0) * (ky == 0 ? 0.5 : 1.0) * 4.0 / ((double)n * (double)n);
    const double a = inv_h2;
    if (gid == 0) return;  // the singular (0, 0) line: the extra workgroup above
    const double d = lam[kx] + lam[ky];
    const double delta = d / (2. * a), tt = delta + sqrt(delta * (delta + 2.));
    const double r = 1. / (1. + tt), lnr = -log1p(tt);
    const double C = scale * r / (a * (tt / (1. + tt)) * (1. + r));
    const double rinv = 1. + tt;
    auto rpow = [&](int e) { return e == 1 ? r : exp((double)e * lnr); };
    auto rinvpow = [&](int e) { return e == 1 ? rinv : exp(-(double)e * lnr); };
    // Loads are issued kZsChunk planes at a time ahead of the (sequential) recursions: the kernel runs only n^2 / 64 waves, each step's
    // load would otherwise cost a full memory latency (measured at 256^3: 130 us unbatched against 79 us for the transform version).
    constexpr int CH = kZsChunk;
    // ---- ascending: F_z (stored in `out`), A = sum f_j r^j with the running power w = r^z (decaying in the direction of travel)
    double F = 0., A = 0.;
    int zprev = planes[0];
    double w = exp((double)zprev * lnr);
    for (int t0 = 0; t0 < n_act; t0 += CH) {
        double fv[CH];
#pragma unroll
        for (int c = 0; c < CH; c++)
            if (t0 + c < n_act) fv[c] = (double)in[(size_t)planes[t0 + c] * plane + gid];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (t0 + c < n_act) {
                const int z = planes[t0 + c];
                if (t0 + c > 0) {
                    const double g = rpow(z - zprev);
                    F *= g;
                    w *= g;
                }
                F += fv[c];
                A += fv[c] * w;
                out[(size_t)z * plane + gid] = (T)F;   // T = float: F is re-read below in storage precision, like every other sweep's intermediate
                zprev = z;
            }
        }
    }
    const int zlast = zprev;
    const double rn = exp((double)n * lnr), q = 1. / (1. - rn * rn);
    double v = exp((double)(n - 1 - zlast) * lnr);   // r^(n-1-z): decays as z descends
    const double B = F * v;
    const double Fm = q * (A + rn * B), Gn = r * q * (B + rn * A);
    // ---- descending: G_z, combine.  r^(z+1) GROWS as z descends: it is started by one exp at the first plane where it is representable
    // (above that its term is below the smallest double) and continued with powers of 1/r
    double G = 0., wz = 0., fprev = 0.;
    bool wz_on = false;
    for (int t0 = n_act - 1; t0 >= 0; t0 -= CH) {
        double fv[CH], Fv[CH];
#pragma unroll
        for (int c = 0; c < CH; c++)
            if (t0 - c >= 0) {
                fv[c] = (double)in[(size_t)planes[t0 - c] * plane + gid];
                Fv[c] = (double)out[(size_t)planes[t0 - c] * plane + gid];
            }
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int t = t0 - c;
            if (t >= 0) {
                const int z = planes[t];
                if (t < n_act - 1) {
                    const int gap = zprev - z;
                    const double g = rpow(gap);
                    G = (G + fprev) * g;
                    v *= g;
                    if (wz_on) wz *= rinvpow(gap);
                }
                if (!wz_on && -(double)(z + 1) * lnr < 700.) {
                    wz = exp((double)(z + 1) * lnr);
                    wz_on = true;
                }
                out[(size_t)z * plane + gid] = (T)(C * (Fv[c] + G + Fm * wz + Gn * v));
                fprev = fv[c];
                zprev = z;
            }
        }
    }
}

}  // namespace shm
