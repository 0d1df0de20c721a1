// Step 1 + 2 with error-budgeted precision tiers (gfx950, wave64): the fp64 solve's kernel since round 3, the fp32 solve's as well since round 5 (every kept
// pair in the packed-fp32 body, fp32 output: conv_tiered_kernel<NPT, float, false>).
//   reference: signed_heat_grid_solver.cpp:48-65 (mesh), :157-174 (points); yukawaPotential signed_heat_3d.cpp:45-49.
//
// X(x) = sum_s w_s e^{-lambda r}/r is dominated, at every node, by the handful of sources nearest to it: a source whose term is a factor
// e^-G below the node's largest term needs a relative accuracy of only budget * e^G.  So the pairs are evaluated in two tiers:
//   near  (fp64, division-free, 2^(k/2048) table + short polynomial: relative error ~1e-12 per term), and
//   far   (packed fp32 with the sub-tile's exponent offset, two nodes per v_pk_*_f32 instruction, ~1e-6 per term and more with the exponent),
// classified per (wave sub-tile, SOURCE) -- not per (workgroup tile, cluster of 64) as the e^-25 branch of conv_normalize_kernel was: a
// wave owns a compact 8 x 8 x NPT block of nodes and each of its 64 lanes tests one source of a 64-source cluster against that block's
// bounds, a ballot turns the 64 answers into two scalar masks, every lane stages its source at its rank in its mask, and the wave then walks the dense
// near list through the fp64 body and the dense far list through the packed fp32 body (no divergence, no per-cluster bounding sphere in the bound).
//   far(s)  <=>  lambda (dist(b_s, box_w) - r_hi_w) > G + ln(|w_s| / |w_near|)
// with box_w the block's bounding box, r_hi_w the distance from the block's farthest corner to the source s* nearest to its centre -- an upper
// bound of every node's distance to its nearest source -- and w_near the weight of s*: every term of s is then below e^-G of the dominant term
// of every node of the block.
// Sources whose terms vanish against the budget are dropped altogether -- round 6: while the RUNNING SUM of their bounds stays inside the budget, not S times the worst case;
// whole clusters first, 64 at a time (see "The drop rule by ACCUMULATED bound" in the kernel).
// Exponent range: the far tier carries ONE offset d0_w per block.  Over the block's nodes a source's exponent lambda' (r - d0_w) reaches up to
// lambda' (dist(b_s, box_w) + 2 rt_w - d0_w); a source that is neither dropped nor guaranteed to stay a normal fp32 number there -- that bound, plus
// log2(w_max / w_s) (the weights are staged relative to the largest one) and 13 bits for the factor 1 / r' and the accumulation -- is evaluated in fp64
// whatever its size (`in_range` below).  Bites on fine meshes over coarse grids and with small tCoef (lambda * cell >~ 2); a few sources per block otherwise.
//
// The budget is ENFORCED, not assumed (round 4).  Y = X / |X| amplifies any error of X by (sum of |terms|) / |X|, which is unbounded where the sheets of the
// source geometry cancel (medial axis): measured at 512^3 / 1024^3 the tiered result left the 1e-8 budget exactly there (|X| / L1 = 2e-3 ... 8e-3, max|dY|
// 1.1e-8 ... 3.3e-8) and nowhere else (tools/tier_worst_nodes.py, profiles/r04_tier_worst_nodes.txt).  So the far loop also accumulates, per node,
// L1_far = sum_far |w_s|_1 e^{-lambda r} / r, and when the block is done every node is tested:
//     eps_far(u) * L1_far  <=  budget * |X|          (P.far_redo_ratio = budget / eps_far(0))
// with eps_far the calibrated relative error of a packed-fp32 term as it shows up in X.  Rounds 4-6: a flat 3e-6.  Round 6, late (tools/r06_tier_calib.sh,
// profiles/r06_tier_calib.txt): two things that number had been covering are now dealt with where they arise --
//   * the ACCUMULATION: an fp32 accumulator that has taken N terms carries ~6e-8 sqrt(N) of its partial sums (thousands of far sources per block on SprayBottle.pc, 1700 on the
//     bunny): the packed-fp32 sums are flushed into the fp64 accumulators every kTierFlush = 256 far sources (12 conversions and 12 fp64 fmas per flush, 0.6 % of the far
//     loop) -- max|dY| against the all-fp64 arithmetic fell by 1.3 ... 2.7 x on every file (bunny 256^3 2.55e-9 -> 9.3e-10);
//   * the EXPONENT: a term's scaled distance u = lambda' r carries ~1e-7 u of rounding into 2^-u, so the price grows with u: eps_far(u) = 1e-6 max(1, u / 24) with u the
//     block's smallest far exponent (coff + G log2 e) -- 3e-6, the old flat price, at u = 72; less below (the blocks near the surface: fewer second passes, the bunny's
//     Step 1 2.8 % shorter), more beyond (the nodes far from every source, where the flat price had let 6e-9 ... 8e-9 through: SprayBottle.pc 1024^3 8.1e-9 -> 6.0e-9).
// Still a calibration, not a bound (the terms' rounding errors are independent, their sum grows slower than L1_far); what is measured is max|dY| <= 6.0e-9 over the ten
// files at up to 1024^3 and the adversarial inputs of the tests.  A block with a failing node walks the sources a second time and evaluates EVERY source in fp64 from
// scratch (the fp64 accumulators hold flushed packed-fp32 sums by then) -- no packed-fp32 term is left in it.  Round 6: the bound of what the block DROPPED enters the
// same test (R |term_s*(x)|), and the second pass evaluates the dropped sources as well.
// The GPU tests hold Y to the stated budget against the C oracle at the full sizes of BASELINE.json (planes through the measured worst nodes) and against
// the all-fp64 kernel (shm_opts.step1_arith = EXACT_F64); shm_opts.step1_budget moves the budget (G, the test and the drop threshold together).
// Round 5 (profiles/NOTES_r01_r05.md section 4.1b): exponent insertion by an integer add with a per-block exponent (yukawa_near), squared z offsets staged per source,
// weights broadcast by op_sel, and a register diet (181 -> 171 with four pairs of a near source in flight) -- 16.4 -> 15.4 VALU instructions per nominal pair.
// Round 6 (profiles/NOTES_r06.md): the accumulated drop rule, the cluster scan as a lane-parallel batch test, the far loop written stage by stage
// (the scheduler's 24 s_nop are gone), the in-launch verdict of the far-rule sample as one word written by compare-and-swap.  176 registers.
#pragma once
#include "shm_kernels.hip.h"

namespace shm {

// e^{-lambda r}/r from x = r^2 for the near tier: hardware v_rsq_f64 seed (2^-24) + one second-order step (relative error 3/8 e^2 ~ 1e-15 in r
// and 1/r), exponent remainder with a single rounding against the 2048-entry table of 2^(j/2048), degree-2 polynomial of 2^(f/2048)
// (truncation (ln2/4096)^3/6 = 8e-13).
// Round 5: the power of two is no longer applied by v_ashrrev + v_ldexp_f64 but ADDED INTO THE EXPONENT FIELD of the table entry by one v_lshl_add_u32:
// with ki = 2048 k + j the low word of the rounding trick holds ki, and (ki << 9) = (k << 20) + (j << 9) -- the table is stored with (j << 9) subtracted
// from the high word of entry j, so hi(entry) + (ki << 9) is the high word of 2^(j/2048) 2^k.  That is only valid while 2^k stays a normal number, so the
// block works relative to ITS OWN exponent: m1 = 1.5 2^52 - 2048 k0 with 2^k0 ~ e^{-lambda d0} (d0: no source is closer to any node of the block), which
// makes the low word ki - 2048 k0 -- every near term of the block comes out scaled by 2^-k0 exactly, the scale is put back once per node at the end of
// the block (where the reference's underflow to 0 -> NaN far from the sources is reproduced by that one v_ldexp_f64).  The host keeps blocks whose near
// terms could span more than 2^-990 off this kernel (Solver::tier_exponent_span_ok).
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
// The quadratic of the near tier's exponential, 2^(f / 2048) ~ 1 + A1 f + A2 f^2 on |f| <= 1/2: with c = ln 2 / 2048 the Taylor coefficient A1 = c leaves the cubic term
// (c f)^3 / 6 <= 8.1e-13; A1 = c (1 + c^2 / 32) -- the best linear stand-in for that cubic on the interval (Chebyshev) -- leaves a quarter of it, 2.0e-13, for nothing
// (round 6, late: at nodes where the sources' terms cancel to 1e-4 of their sum -- inside SprayBottle.pc at 1024^3 -- the 8e-13 was the whole 8.1e-9 of the measured
// error of Y: tools/r06_worst_node_diag.py, NOTES_r06.md section 7)
constexpr double kNearA1 = 3.384507729693224e-04;
__device__ __forceinline__ double yukawa_near(double x, double c, double m1, const uint2v* __restrict__ tab) {
    const double y0 = __builtin_amdgcn_rsq(x);
    const double t = x * y0;
    const double h = 0.5 * y0;
    const double e = fma(-t, h, 0.5);            // (1 - x y0^2) / 2
    const double r = fma(t, e, t);
    const double rinv = fma(y0, e, y0);
    const double tm = fma(r, c, m1);             // round(r c) - 2048 k0 lands in the low mantissa bits
    const double kf = tm - m1;
    const unsigned ki = (unsigned)__double_as_longlong(tm);
    const double f = fma(r, c, -kf);
    double p = 5.727446245172041e-08;                   // (ln2/2048)^2 / 2
    p = fma(p, f, kNearA1);
    p = fma(p, f, 1.0);
    uint2v tv = tab[ki & 2047u];
    tv.y += ki << 9;
    return __builtin_bit_cast(double, tv) * (p * rinv);   // r = 0 -> NaN (0 * inf), like exp(0)/0 -> inf -> NaN after normalise
}

// the same for B values at once, stage by stage (breadth-first): the source order the scheduler starts from interleaves the B dependent chains
template <int B>
__device__ __forceinline__ void yukawa_near_batch(const double* __restrict__ x, double c, double m1, const uint2v* __restrict__ tab, double* __restrict__ g) {
    double y0[B], t[B], e[B], r[B], rinv[B], tm[B], f[B], p[B];
    unsigned ki[B];
    uint2v tv[B];
#pragma unroll
    for (int b = 0; b < B; b++) y0[b] = __builtin_amdgcn_rsq(x[b]);
#pragma unroll
    for (int b = 0; b < B; b++) t[b] = x[b] * y0[b];
#pragma unroll
    for (int b = 0; b < B; b++) e[b] = fma(-t[b], 0.5 * y0[b], 0.5);
#pragma unroll
    for (int b = 0; b < B; b++) r[b] = fma(t[b], e[b], t[b]);
#pragma unroll
    for (int b = 0; b < B; b++) tm[b] = fma(r[b], c, m1);
#pragma unroll
    for (int b = 0; b < B; b++) {
        ki[b] = (unsigned)__double_as_longlong(tm[b]);
        tv[b] = tab[ki[b] & 2047u];
        tv[b].y += ki[b] << 9;
    }
#pragma unroll
    for (int b = 0; b < B; b++) f[b] = fma(r[b], c, -(tm[b] - m1));
#pragma unroll
    for (int b = 0; b < B; b++) rinv[b] = fma(y0[b], e[b], y0[b]);
#pragma unroll
    for (int b = 0; b < B; b++) p[b] = fma(fma(5.727446245172041e-08, f[b], kNearA1), f[b], 1.0);
#pragma unroll
    for (int b = 0; b < B; b++) {
        g[b] = __builtin_bit_cast(double, tv[b]) * (p[b] * rinv[b]);
    }
}

// acc += w.lo * g  /  acc += w.hi * g  on two nodes at once: the weight is one half of a register pair, broadcast to both halves of the instruction by op_sel.  The
// compiler finds the low-half form by itself but copies a high half into a fresh pair first (v_mov_b32: 8 of the far loop's 112 vector instructions); spelled out here.
#ifndef SHM_TIER_ASM_BCAST
#define SHM_TIER_ASM_BCAST 1
#endif
__device__ __forceinline__ void pk_fma_lo(float2v& acc, float2v w, float2v g) {
#if SHM_TIER_ASM_BCAST
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(g));
#else
    acc = __builtin_elementwise_fma(float2v{w.x, w.x}, g, acc);
#endif
}
__device__ __forceinline__ void pk_fma_hi(float2v& acc, float2v w, float2v g) {
#if SHM_TIER_ASM_BCAST
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(g));
#else
    acc = __builtin_elementwise_fma(float2v{w.y, w.y}, g, acc);
#endif
}

// set bits of a wave mask below this lane (v_mbcnt_lo / _hi: no (1 << lane) - 1 held in a register pair)
__device__ __forceinline__ int mask_rank(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// 2^(u / 2048) for 0 <= u < 4096 from the near tier's table (same remainder polynomial): the block's far-tier scale, once per block
__device__ __forceinline__ double exp2_tab(double u, const uint2v* __restrict__ tab) {
    const double tm = u + 6755399441055744.0;
    const unsigned ki = (unsigned)__double_as_longlong(tm);
    const double f = u - (tm - 6755399441055744.0);
    uint2v tv = tab[ki & 2047u];
    tv.y += ki << 9;
    return __builtin_bit_cast(double, tv) * fma(fma(5.727446245172041e-08, f, kNearA1), f, 1.0);
}

// a wave-uniform float as a scalar register (the builtin is integer-typed: pass the bits, not the value)
__device__ __forceinline__ float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// ---- The sample's verdict (far rule of the fp64 solve, Solver::far_rule_plan), handed from the sample blocks to every wave of the SAME launch (round 6: hardened) ----
// ctr[0] far pairs, ctr[1] pairs evaluated again, ctr[2] sample blocks that have reported, ctr[3] the verdict: 0 undecided, 1 differential rule, 2 box rule.
// All four words are only ever touched by agent-scope atomics (served by the L2 / fabric: no CU's L1 can hold a stale copy; MI355X_MICROARCH.md, inter-workgroup
// visibility).  A sample block adds its two counts (relaxed) and then arrives with a RELEASE add on ctr[2] -- its counts are visible before its arrival is --;
// the block whose arrival completes the sample reads the totals behind an ACQUIRE fence and publishes the verdict with ONE compare-and-swap on ctr[3].  A reader
// polls ctr[3] with relaxed loads and s_sleep; there is no payload behind the flag, so the value it reads IS the verdict.  The spin is bounded: a reader that
// gives up (~1 s: it means the sample's blocks are not being worked on -- a launch smaller than the resident set its queues assume) tries to publish "box rule"
// itself, and whoever's compare-and-swap lands first decides for the WHOLE launch: no wave ever decides on its own, Y stays one function of the verdict word.
__device__ __forceinline__ int sample_verdict(unsigned long long* ctr) {
    int v = 0;   // (wave-uniform: lane 0's reading)
    for (unsigned spins = 0; spins < (1u << 20); spins++) {
        v = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ctr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v != 0) break;
        __builtin_amdgcn_s_sleep(32);
    }
    if (v == 0) {   // timed out: the whole launch falls back to the box rule, unless a verdict lands first
        int r = 0;
        if ((threadIdx.x & 63u) == 0u) {
            unsigned long long expected = 0ull;
            const bool won = __hip_atomic_compare_exchange_strong(ctr + 3, &expected, 2ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r = won ? 2 : (int)expected;
        }
        v = __builtin_amdgcn_readfirstlane(r);
    }
    return v == 1;
}
__device__ __forceinline__ void sample_report(unsigned long long* ctr, unsigned long long far_pairs, unsigned long long redo_pairs, int sample_blocks) {
    __hip_atomic_fetch_add(ctr + 0, far_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(ctr + 1, redo_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long before = __hip_atomic_fetch_add(ctr + 2, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1ull == (unsigned long long)sample_blocks) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const unsigned long long farp = __hip_atomic_load(ctr + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long redo = __hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long expected = 0ull;   // at most 4 % of the sample's far pairs evaluated again -> the differential rule
        __hip_atomic_compare_exchange_strong(ctr + 3, &expected, (farp > 0ull && redo * 25ull <= farp) ? 1ull : 2ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifndef SHM_TIER_TX
#define SHM_TIER_TX 8
#endif
constexpr int kTierTX = SHM_TIER_TX, kTierTY = 64 / kTierTX;   // a wave's block of nodes is kTierTX x kTierTY x NPT (one z-column of NPT nodes per lane)
// The budget of the tiers on the normalised field Y (what tests/test_gpu_parity.py asserts against the C oracle at BASELINE.json's full sizes), and the
// calibrated relative error of a packed-fp32 term as it shows up in X: eps_far(u) = kTierEpsFar max(1, u / kTierU0) (see the header).  History: rounds 4-6 priced every
// term at a flat 3e-6 ("five times the largest max|dY| |X| / L1_far measured": 4.8e-7 on bunny_small 512^3, 6e-7 on SprayBottle.pc 1024^3 at lambda r = 28); round 4 had
// measured a price proportional to lambda' r ON TOP of that flat floor (SprayBottle.pc 1024^3 8.3e-9 -> 7.7e-9 for +9 % of its Step 1; profiles/r04_eps_sweep.txt) and not
// adopted it.  With the accumulation taken out of the number (kTierFlush) the floor is a third of it, the slope meets the old price at u = 72, and the same sweep reads
// SprayBottle.pc 1024^3 8.1e-9 -> 6.0e-9 for +3.5 % (fp64 solve), knot 1024^3 5.5e-9 -> 2.1e-9 for -1.8 %, the bunny 2.6e-9 -> 1.3e-9 for -2.8 % (profiles/r06_tier_calib.txt).
// A block where eps_far L1_far > budget |X| at any node is evaluated again, every source in fp64: 0.1 ... 0.8 % of the pairs on the shipped data.
constexpr double kTierBudget = 1.0e-8, kTierEpsFar = 1.0e-6;   // (eps_far at small exponents: see the header)
constexpr double kTierU0 = 24.0;  // exponent beyond which the a-posteriori test prices a packed-fp32 term at eps_far u / u0 (0: flat)
constexpr int kTierFlush = 256;   // packed-fp32 sources between two flushes of their sums into the fp64 accumulators (0: never)
constexpr int kTierCluster = 64;                    // sources per cluster = lanes per wave: one source per lane in the classification
constexpr int kTierChunk = 4;                       // clusters per LDS fill
constexpr int kTierFill = kTierCluster * kTierChunk;

// counters (optional, may be null): [0] (node, source) pairs evaluated in fp64, [1] in packed fp32 -- what `roofline.frac` is computed from --
// [2] pairs evaluated a second time in fp64 by the a-posteriori check (they are part of [0] as well).
//
// Work distribution: the unit of work is one wave's sub-tile (8 x 8 x NPT nodes).  Waves are independent -- each pulls the next unit from a global
// counter (one atomic per ~1 ms of work), scans the sources for its own bounds, stages 64 sources at a time into its own LDS region (lane l loads
// source l: the staging needs no workgroup barrier, a wave's LDS accesses execute in order) and walks its own masks.  The workgroup shares only the
// read-only exponential table.  Units cost between ~0.3 and 1 of the all-fp64 cost depending on how much of the object is near, so static
// assignment (and a barrier per LDS fill across four differently loaded waves) would leave the SIMDs idle for a fifth of the kernel.
#ifndef SHM_TIER_FAR_UNROLL
#define SHM_TIER_FAR_UNROLL 4
#endif
#ifndef SHM_TIER_LDS_FETCH
#define SHM_TIER_LDS_FETCH 1    // the next cluster's sources travel global -> LDS directly; 0: through 12 registers per lane (rounds 2-3)
#endif
#ifndef SHM_TIER_LOOKAHEAD
#define SHM_TIER_LOOKAHEAD 1
#endif
#ifndef SHM_TIER_NEAR_BATCH
#define SHM_TIER_NEAR_BATCH 4   // pairs whose e^{-lambda r}/r chains are interleaved stage by stage (round 5: all four of a lane's z-column; 2: +2.5 % Step 1)
#endif
#ifndef SHM_TIER_WAVES_PER_EU
#define SHM_TIER_WAVES_PER_EU 2
#endif
// TY = double, CHECK = true: Step 1 of the fp64 solve (near tier fp64, far tier packed fp32 under the enforced budget).
// TY = float, CHECK = false (round 5): Step 1 of the fp32 solve -- the host sets the far threshold to -infinity, so every source that is neither dropped nor
// outside the far tier's exponent range goes through the packed-fp32 body (no L1 sums, no second pass; the few out-of-range sources keep the fp64 body), Y is
// stored in fp32.  What the fp32 solve gains over conv_normalize_kernel<float> is the culling per (8 x 8 x NPT block, SOURCE) instead of per (8 x 8 x 32 tile,
// cluster of 32) -- SprayBottle.pc at 256^3: 2 x -- and this kernel's cheaper far body.
// Register budget: <= 184 (two of its waves + one wave of a set-up kernel per SIMD; tests/test_abi_and_host.py pins it from the compiler's report -- the
// amdgpu_num_vgpr attribute is not honoured beside waves_per_eu, so the budget is kept by construction: 171 / 168 registers).
template <int NPT, typename TY, bool CHECK>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(SHM_TIER_WAVES_PER_EU, SHM_TIER_WAVES_PER_EU))) void conv_tiered_kernel(
    ConvParams P, const double* __restrict__ src /* [S][6]: pos xyz, wn xyz */, const float* __restrict__ clusters,
    const double* __restrict__ exp_tab_g /* [2048]: 2^(j/2048) */, TY* __restrict__ Y0, TY* __restrict__ Y1, TY* __restrict__ Y2,
    unsigned long long* __restrict__ counters, unsigned* __restrict__ next_unit /* [8] queue heads, zeroed before the launch */) {
    static_assert(NPT % 2 == 0, "the far tier handles a lane's nodes in packed pairs");
    constexpr int kWaves = kBlock / kWave;
    constexpr int kFarUnroll = SHM_TIER_FAR_UNROLL;
    constexpr int kPad = kFarUnroll;   // staged entries behind the last real one: zero weight, far away -- they pad the
    __shared__ double stage64[kWaves][(kTierCluster + kPad) * 6];              // compacted lists to whole groups of sources in flight
    // far list (round 5: three arrays instead of one 32-byte record): A = x', y', wx, wy | B = (z_e - z')^2 for the block's NPT planes -- the planes are the same for
    // every lane, so the classifying lane squares them once per source and the far loop gets d^2 of two nodes by ONE packed add (it used to take a packed add and a
    // packed fma per two nodes, and a subtraction per source) | C = wz, |w|_1.  Weights sit in aligned pairs: either half is broadcast by op_sel (pk_fma_lo / _hi)
    __shared__ __attribute__((aligned(16))) float farA[kWaves][(kTierCluster + kPad) * 4];
    __shared__ __attribute__((aligned(16))) float farB[kWaves][(kTierCluster + kPad) * NPT];
    __shared__ __attribute__((aligned(16))) float farC[kWaves][(kTierCluster + kPad) * 2];
    // the next cluster's sources travel global -> LDS directly (global_load_lds_dwordx4: three 16-byte pieces of every lane's 48-byte record, each piece
    // landing at wave base + lane * 16), not through 12 registers per lane held across the two loops: those registers are what the per-node L1 sums of the
    // a-posteriori test live in (round 4; the kernel must stay within 184 VGPRs for the set-up kernels to run beside it)
#if SHM_TIER_LDS_FETCH
    __shared__ double2 raw[kWaves][3][kTierCluster];
#endif
    __shared__ uint2v exp_tab[2048];   // bits of 2^(j/2048) with (j << 9) taken off the high word (see yukawa_near)
    __shared__ float4 star_stash[kWaves];   // the block's reference source (scaled position, squared weight), parked here between the block's set-up and its a-posteriori test
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    double* const tile = stage64[wave];
    float* const tA = farA[wave];
    float* const tB = farB[wave];
    float* const tC = farC[wave];
    for (int a = threadIdx.x; a < 2048; a += kBlock) {
        uint2v tv = __builtin_bit_cast(uint2v, exp_tab_g[a]);
        tv.y -= (unsigned)a << 9;
        exp_tab[a] = tv;
    }
    __syncthreads();   // the only workgroup barrier: the table
    const int n = P.n;
    const size_t plane = (size_t)n * n;
    const float lam_l2 = (float)(P.lambda * 1.4426950408889634);     // lambda in powers of two per unit length
    const float g_l2 = P.tier_log * 1.4426950408889634f;
    const bool drop_on = P.drop_eps_soft > 0.f;   // (wave-uniform: kernel argument)
    const float cellq = (float)(P.cell * (P.lambda * 1.4426950408889634));   // the cell size in the far tier's scaled units
    const float lws = uniform_f32((float)log2(P.wscale));                      // log2 of the weights' scale factor (a power of two)
    constexpr double kHalfZ = 0.5 * (NPT - 1);
    constexpr double kHalfX = 0.5 * (kTierTX - 1), kHalfY = 0.5 * (kTierTY - 1);
    const float rt_w = (float)(sqrt(kHalfX * kHalfX + kHalfY * kHalfY + kHalfZ * kHalfZ) * P.cell) * 1.000001f;
    const float hx = uniform_f32((float)(kHalfX * P.cell) * 1.000001f), hy = uniform_f32((float)(kHalfY * P.cell) * 1.000001f),
                hz = uniform_f32((float)(kHalfZ * P.cell) * 1.000001f);   // half extents of a block
    unsigned long long cnt_near = 0, cnt_far = 0, cnt_redo = 0;
    int decided = -1;   // the sample's verdict, once this wave has needed it
    // Eight queue heads, one per XCD (workgroup b runs on XCD b % 8): the units -- x fastest, then y, then z -- are cut into eight contiguous ranges, so
    // that the two 64-byte halves of a 128-byte line of Y (x-adjacent blocks) are written through the same L2 and leave it as one line; an XCD
    // whose range is exhausted takes units from the others' (work stealing keeps the end of the kernel balanced).
    const unsigned n_units = (unsigned)P.n_tiles;   // n_tiles counts sub-tiles here
    const unsigned per_q = (n_units + 7u) / 8u;
    const unsigned my_q = blockIdx.x & 7u;
    unsigned q_off = 0;   // queues tried so far: my_q, my_q + 1, ... (mod 8)
    for (;;) {
        unsigned unit = 0xffffffffu;
        while (q_off < 8u) {
            const unsigned q = (my_q + q_off) & 7u;
            const unsigned lo = q * per_q, hi = min(n_units, lo + per_q);
            unsigned t = 0;
            if (lane == 0) t = atomicAdd(next_unit + q, 1u);
            t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
            if (lo + t < hi) {
                unit = lo + t;
                break;
            }
            q_off++;
        }
        if (unit == 0xffffffffu) break;
        const int tzq = (int)(unit / (unsigned)(P.tiles_x * P.tiles_y)), trem = (int)unit - tzq * (P.tiles_x * P.tiles_y);
        const int tz = P.layer_order ? P.layer_order[tzq] : tzq;   // (the queues run from the layers at the grid's centre to the ones at its faces: launch_conv)
        // Which far rule this block classifies with (see Solver::far_rule_plan): P.far_rule for every block, or per layer of the queue order -- 1: the SAMPLE (differential
        // rule; its counters decide), 0: box rule whatever the sample says (the layers every queue works through while the sample finishes), 2: what the sample earned.
        bool use_diff = false;
        int rule_sel = -1;
        if constexpr (CHECK) {
            use_diff = P.far_rule != 0;
            if (P.unit_rule) {
                rule_sel = (int)P.unit_rule[tzq];
                if (rule_sel == 2) {
                    if (decided < 0) decided = sample_verdict(P.sample_ctr);   // (once per wave; one word, written once per launch: every wave reads the same verdict)
                    use_diff = decided != 0;
                } else {
                    use_diff = rule_sel == 1;
                }
            }
        }
        const unsigned long long far_before = cnt_far, redo_before = cnt_redo;
        const int ty = trem / P.tiles_x, tx = trem - ty * P.tiles_x;
        const int i0 = tx * kTierTX, j0 = ty * kTierTY, kk0 = P.kk_begin + tz * NPT;

        double pz0 = 0., ax[NPT], ay[NPT], az[NPT];
        float qz0 = 0.f;
        float2v fx[NPT / 2], fy[NPT / 2], fz[NPT / 2];   // far sums: nodes (2p, 2p + 1) of the lane's z-column in one register pair
        const int li = i0 + (lane % kTierTX), lj = j0 + (lane / kTierTX);
        const int ci = min(li, n - 1), cj = min(lj, n - 1);
        // indicesToNodePosition: (i,j,k)*cellSize + bboxMin, evaluated in double like the reference (:510-514)
        const double px = ci * P.cell + P.bbox_min[0], py = cj * P.cell + P.bbox_min[1];
        const float qx = (float)(px * (P.lambda * 1.4426950408889634)), qy = (float)(py * (P.lambda * 1.4426950408889634));   // far tier: scaled coordinates (see the far loop)
#pragma unroll
        for (int e = 0; e < NPT; e++) {
            ax[e] = ay[e] = az[e] = 0.;
            fx[e / 2] = fy[e / 2] = fz[e / 2] = float2v{0.f, 0.f};
        }
        // the wave's nodes form a compact kTierTX x kTierTY x NPT block: a lane keeps the z of its first node, the others follow by the cell size (planes past the
        // end of the launch are evaluated like the others and not stored)
        pz0 = (P.k0 + kk0 - 1) * P.cell + P.bbox_min[2];
        {   // (the same for every lane of the wave: a scalar register pair, not two vector registers held through the loops)
            const long long b = __double_as_longlong(pz0);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
            pz0 = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
        qz0 = uniform_f32((float)(pz0 * (P.lambda * 1.4426950408889634)));
        const bool live_xy = li < n && lj < n;
        // nearest source of the block's centre (and its weight): one source per lane and step, butterfly minimum
        // (wave-uniform values computed by vector instructions are moved to scalar registers by hand: the compiler would keep a copy per lane)
        const float cx = uniform_f32((float)((i0 + kHalfX) * P.cell + P.bbox_min[0])), cy = uniform_f32((float)((j0 + kHalfY) * P.cell + P.bbox_min[1]));
        const float cz = uniform_f32((float)((P.k0 + kk0 - 1 + kHalfZ) * P.cell + P.bbox_min[2]));
        float dmin = 3.0e38f, wnear = 0.f, nx = 0.f, ny = 0.f, nz = 0.f;   // nearest source: squared distance, squared weight, offset from the centre
        for (int s = lane; s < P.S; s += kWave) {
            const double* q = src + (size_t)s * 6;
            const float dx = cx - (float)q[0], dy = cy - (float)q[1], dz = cz - (float)q[2];
            const float w0 = (float)(q[3] * P.wscale), w1 = (float)(q[4] * P.wscale), w2c = (float)(q[5] * P.wscale);   // (scaled: see the staging below)
            const float d2 = dx * dx + dy * dy + dz * dz, w2 = w0 * w0 + w1 * w1 + w2c * w2c;
            if (d2 < dmin || (d2 == dmin && w2 > wnear)) {   // ties (the zero-weight padding repeats a source) go to the larger weight
                dmin = d2;
                wnear = w2;
                nx = dx;
                ny = dy;
                nz = dz;
            }
        }
        {   // the lane that holds the nearest source (ties: the larger weight, then the lower lane), then its five values by v_readlane.  Minimum / maximum over the wave
            // by ds_swizzle butterflies within the halves + two v_readlane: no per-step lane index held in a register (the __shfl_xor form kept six of them live
            // through the whole kernel), a sixth of the instructions of a five-value butterfly
            auto wave_min_u = [&](unsigned v) {
                v = min(v, (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (1 << 10)));    // bit-mask mode: lane ^ 1, 2, 4, 8, 16 within each half
                v = min(v, (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (2 << 10)));
                v = min(v, (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (4 << 10)));
                v = min(v, (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (8 << 10)));
                v = min(v, (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (16 << 10)));
                return min((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 32));
            };
            // (squared distances and squared weights are >= 0: their bit patterns order like the values)
            const unsigned dbits = wave_min_u(__float_as_uint(dmin));
            const unsigned wkey = __float_as_uint(dmin) == dbits ? ~__float_as_uint(wnear) : 0xffffffffu;
            const unsigned wbits = wave_min_u(wkey);
            const unsigned long long hit = __ballot(wkey == wbits);
            const int win = (int)__builtin_ctzll(hit);
            dmin = __uint_as_float(dbits);
            wnear = __uint_as_float(~wbits);
            nx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nx), win));
            ny = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ny), win));
            nz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nz), win));
        }
        dmin = sqrtf(dmin);
        // every node of the block has a source at most r_hi_w away: the block's farthest corner from the source nearest to its centre (the block is the box
        // centre +- (hx, hy, hz); tighter than dmin + rt_w unless that source lies along a diagonal)
        const float fxn = fabsf(nx) + hx, fyn = fabsf(ny) + hy, fzn = fabsf(nz) + hz;
        const float r_hi_w = uniform_f32(sqrtf(fxn * fxn + fyn * fyn + fzn * fzn) * 1.000001f);
        const float lnear_w = uniform_f32(0.5f * __log2f(fmaxf(wnear, 1e-37f)) - 1e-5f);         // log2 of that source's weight, rounded down
        const float d0_w = uniform_f32(fmaxf(0.f, dmin * 0.999999f - rt_w));                      // no source is closer than this to any node of the block
        const float coff = uniform_f32(lam_l2 * d0_w);   // far tier: e^{-lambda (r - d0)} = 2^(coff - lambda log2 e r), folded back in by e^{-lambda d0} at the end

        // the block's own exponent (see yukawa_near): 2^k0 ~ e^{-lambda d0}; r >= d0 for every pair of the block, so the scaled near terms are at most 2^1
        const int k0 = __builtin_amdgcn_readfirstlane((int)floor((double)d0_w * P.cexp * (1.0 / 2048.0)));
        const double m1 = 6755399441055744.0 - 2048.0 * (double)k0;   // 1.5 * 2^52 - 2048 k0: exact
        // the far sums hold w_scaled 2^{-lambda' (r - d0)} / (lambda' r): times e^{-lambda d0} lambda' / wscale they are terms of X; here in the block's scale 2^-k0
        // e^{-lambda d0} / 2^k0 = 2^((d0 c - 2048 k0) / 2048), exponent in [0, 2048): from the near tier's table.  (Evaluated where it is used, after the loops -- the
        // empty asm keeps the compiler from hoisting it above them, where it would hold two registers throughout.)
        auto far_scale = [&]() {
            float d = d0_w;
            asm volatile("" : "+v"(d));
            return exp2_tab(fma((double)d, P.cexp, -2048.0 * (double)k0), exp_tab) * (P.lambda * 1.4426950408889634) / P.wscale;
        };
        float2v fl[NPT / 2];   // per node: sum over the far sources of |w|_1 e^{-lambda (r - d0)} / (lambda' r) -- the L1 norm of what the packed-fp32 tier contributed
#pragma unroll
        for (int e = 0; e < NPT / 2; e++) fl[e] = float2v{0.f, 0.f};
        // constant of the exponent-range test of a far source: lambda' (dist + 2 rt_w) - coff - log2(w_s / w_max) <= 113  (see the header)
        const float range_c = uniform_f32(2.f * rt_w * lam_l2 - 113.f - coff);
        // the differential far rule's block constants: unit vector from the nearest source s* to the centre, its distance from the centre (rounded up) and the reciprocal of
        // its distance from the block's box (inf when the block holds it: the rule then never fires)
        float star_ux = 0.f, star_uy = 0.f, star_uz = 0.f, star_dc = 0.f, star_inv_d = 0.f;
        if constexpr (CHECK) {
            // What a packed-fp32 term's error is grows with its exponent: the scaled distance u = lambda' r carries its rounding, ~1e-7 u, into 2^-u.  Far from the sources (u in
            // the hundreds: point clouds with lambda r ~ 1e2 ... 1e3) every term of a node is off by several 1e-6; under the box rule the far tier's share of |X| is so small
            // there that it does not show (6.0e-9 of the 1e-8 budget at worst, measured, with the test's price growing with u -- see the header), but the differential rule
            // triples that share: SprayBottle.pc read 1.8e-8 against the all-fp64 kernel with it (profiles/r05_far_rule.txt).  So the rule is used only in blocks whose
            // nearest terms stay below u = 72 (lambda' (d0 + 2 rt) <= 72, lambda r <~ 50: the whole grid of a mesh whose mean edge is a few cells, the neighbourhood of a
            // dense point cloud); everywhere else the box rule and its measured margin stand.
            use_diff = use_diff && coff + 2.f * rt_w * lam_l2 <= 72.f;
        }
        // (round 6: the drop rule reads the same differential bound in every block -- it decides what is evaluated at all, not in which arithmetic)
        if (use_diff || drop_on) {
            const float inv = dmin > 0.f ? 1.f / dmin : 0.f;
            star_ux = uniform_f32(nx * inv);
            star_uy = uniform_f32(ny * inv);
            star_uz = uniform_f32(nz * inv);
            star_dc = uniform_f32(dmin * 1.000001f);
            const float bx = fmaxf(fabsf(nx) - hx, 0.f), by = fmaxf(fabsf(ny) - hy, 0.f), bz = fmaxf(fabsf(nz) - hz, 0.f);
            const float db = sqrtf(bx * bx + by * by + bz * bz);
            star_inv_d = uniform_f32(dmin > 0.f && db > 0.f ? 1.f / db : __builtin_huge_valf());
        }
        // ---- The drop rule by ACCUMULATED bound (round 6) ----
        // For a source s and the block's reference source s* (the one nearest to the block's centre), at every node x of the block
        //     |term_s(x)| / |term_s*(x)| = (|w_s| / |w_s*|) e^{-lambda (r_s(x) - r_s*(x))} r_s*(x) / r_s(x)  <=  b_s := (|w_s| / |w_s*|) e^{-lambda gap_s} r_hi / d_s
        // with gap_s the larger of the two lower bounds of r_s - r_s* over the block -- the box rule's d_s - r_hi (d_s: distance from s to the block's box, r_hi: farthest corner
        // from s*) and the differential rule's (see the classification below) -- and r_s* <= r_hi, r_s >= d_s.  Rounds 3-5 dropped s when b_s <= eps / S: all S sources together then
        // stay below eps of |term_s*(x)| <= the node's dominant term.  That is the same inequality with a factor S that is loose by 4-10 e-folds (most sources are orders of magnitude
        // below the threshold; what sits near it is a ring of a few hundred).  Now: the block keeps the running sum R of the bounds of what it has dropped, in the fixed order of the
        // scan (clusters in storage order, a cluster's candidates together -- Y stays bit-identical from run to run), and a candidate (b_s <= tau = eps_soft / K, K a per-problem
        // estimate of the ring's population: Solver::drop_rule_plan) is dropped while R + (the cluster's candidates) <= eps_soft = 7/8 eps.  K only decides how well the budget is
        // used; the sum is what makes the rule sound.  A source with b_s <= tau_hard = (eps / 8) / S is dropped whatever R (the old rule on an eighth of the budget: it bounds how
        // far an evaluated source can lie from the block -- the exponent span of yukawa_near -- and keeps the new rule from ever doing much worse than the old one).
        // Whole clusters go the same way with their bounding sphere, largest weight (candidate test) and weight sum (bound).
        // fp64 solve: R enters the a-posteriori test (R |term_s*(x)| against budget |X(x)|, beside the packed-fp32 tier's L1 sums), and the second pass evaluates the dropped
        // sources too: where the sheets of the geometry cancel (|X| << dominant term) the rule of rounds 3-5 had no such guard.
        float R_soft = 0.f, R_hard = 0.f;   // wave-uniform
        if constexpr (CHECK) {   // (four registers less through the loops; a wave's LDS accesses execute in order)
            if (drop_on && lane == 0) star_stash[wave] = float4{(cx - nx) * lam_l2, (cy - ny) * lam_l2, (cz - nz) * lam_l2, wnear};
        }
        const float span_c = uniform_f32(2.f * rt_w * lam_l2 - coff - 960.f);   // pass 1: a dropped source is evaluated only where its exponent stays inside the block's span
        // pass 0: near sources in fp64, far ones in packed fp32.  pass 1 (only when the a-posteriori test failed): every source in fp64, from cleared accumulators.
        int far_pending = 0;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
        // the lane's source of cluster c: fetched into the wave's raw LDS region one cluster ahead (issued once cluster c - 1 has been read out of it, in flight
        // while that cluster's two loops run; a skipped cluster's fetch is simply overwritten by the next: loads return in order)
#if SHM_TIER_LDS_FETCH
        auto fetch_cluster = [&](int c) {
            // (scalar base + one 32-bit lane offset: the three pieces share the lane's offset register instead of holding three 64-bit addresses)
            const char* g = reinterpret_cast<const char*>(src) + (size_t)c * (kTierCluster * 48);
            const unsigned lo = (unsigned)lane * 48u;
            __builtin_amdgcn_global_load_lds(g + lo, &raw[wave][0][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds(g + 16 + lo, &raw[wave][1][0], 16, 0, 0);
            __builtin_amdgcn_global_load_lds(g + 32 + lo, &raw[wave][2][0], 16, 0, 0);
        };
#else
        double nq[6];
        auto fetch_cluster = [&](int c) {
            const double* qn = src + ((size_t)c * kTierCluster + lane) * 6;
#pragma unroll
            for (int a = 0; a < 6; a++) nq[a] = qn[a];
        };
#endif
        // ---- whole clusters against the block (bounding spheres), SIXTY-FOUR AT A TIME (round 6) ----
        // Rounds 3-5 tested one cluster per trip of a scalar loop -- six scalar loads, a dozen vector instructions on wave-uniform values, a branch: a dependent chain per
        // cluster, 216 of them per block on rocker, 817 on SprayBottle.pc, most of them for clusters that are dropped.  Now a lane tests ONE cluster of a batch of 64 (its
        // record by three 8-byte loads), ballots turn the 64 answers into scalar masks, and the walk below steps from set bit to set bit.  The accumulated drop rule takes a
        // batch's candidates together -- all of them while the running sum allows, else only the ones below the hard threshold --, in the fixed order of the batches; pass 1
        // repeats the masks and walks what pass 0 dropped as well (`cdrop_mask` tells the classification that every source of such a cluster was dropped).
        int cur_batch = -1;
        unsigned long long kept_mask = 0ull, cdrop_mask = 0ull;
        auto load_batch = [&](int b) {
            cur_batch = b;
            const int ci = b * kWave + lane;
            const bool in = ci < P.n_clusters;
            const unsigned long long validm = __ballot(in);
            cdrop_mask = 0ull;
            if (drop_on) {
                const float2* rec = reinterpret_cast<const float2*>(clusters + (size_t)min(ci, P.n_clusters - 1) * kConvClusterRec);
                const float2 r0 = rec[0], r1 = rec[1], r2 = rec[2];   // centre xy | centre z, radius | ln of the largest weight, ln of the sum of the weights (UNscaled, rounded up)
                const float gdx = cx - r0.x, gdy = cy - r0.y, gdz = cz - r1.x;
                const float gap = sqrtf(gdx * gdx + gdy * gdy + gdz * gdz) * 0.999999f - rt_w - r1.y - r_hi_w;
                const float lrel = lws - lnear_w - gap * lam_l2;
                const float lbmax = fmaf(r2.x, 1.4426950408889634f, lrel) + 2e-5f;   // log2 of the largest bound of any of its sources
                const bool cand = in && gap > 0.f && lbmax <= P.drop_ltau;
                const unsigned long long candm = __ballot(cand);
                if (candm != 0ull) {
                    // a candidate cluster's sources together: (sum of weights / |w_*|) e^{-lambda gap} r_hi / (r_hi + gap)
                    float bc = cand ? __builtin_amdgcn_exp2f(fmaf(r2.y, 1.4426950408889634f, lrel)) * (r_hi_w * __builtin_amdgcn_rcpf(r_hi_w + gap)) * 1.0001f : 0.f;
                    bc += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bc), 0x1f | (1 << 10)));
                    bc += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bc), 0x1f | (2 << 10)));
                    bc += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bc), 0x1f | (4 << 10)));
                    bc += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bc), 0x1f | (8 << 10)));
                    bc += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bc), 0x1f | (16 << 10)));
                    const float sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bc), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bc), 32));
                    const float r_new = uniform_f32(R_soft + sum);
                    if (__builtin_amdgcn_readfirstlane((int)(r_new <= P.drop_eps_soft))) {
                        R_soft = r_new;
                        cdrop_mask = candm;
                    } else {
                        cdrop_mask = __ballot(cand && lbmax <= P.drop_ltau_hard);
                        R_hard = uniform_f32(R_hard + (float)(__builtin_popcountll(cdrop_mask) * kTierCluster) * P.drop_tau_hard);
                    }
                }
            }
            kept_mask = pass == 1 ? validm : validm & ~cdrop_mask;
        };
        // the next cluster >= `from` this pass walks (P.n_clusters: none); cdrop: pass 0 dropped it as a whole
        auto next_kept = [&](int from, bool& cdrop) {
            cdrop = false;
#pragma unroll 1
            while (from < P.n_clusters) {
                const int b = from >> 6;
                if (b != cur_batch) load_batch(b);
                const unsigned long long m = kept_mask & (~0ull << (from & 63));
                if (m != 0ull) {
                    const int bit = (int)__builtin_ctzll(m);
                    cdrop = ((cdrop_mask >> bit) & 1ull) != 0ull;
                    return (b << 6) + bit;
                }
                from = (b + 1) << 6;
            }
            return P.n_clusters;
        };
        // The fetch runs one KEPT cluster ahead (round 5 in the fp64 solve; the fp32 solve kept the plain order then because the scalar scan for the next kept cluster sat between
        // a cluster's fetch and its use -- with the scan a few scalar instructions on a mask, round 6, both solves look ahead).
        R_soft = 0.f;   // (the second pass repeats the first one's scan and with it its sums and decisions)
        R_hard = 0.f;
        bool cdrop_cur = false, cdrop_next = false;
        int c = next_kept(0, cdrop_cur);
        if (c < P.n_clusters) fetch_cluster(c);
#pragma unroll 1
        while (c < P.n_clusters) {
            const int c_next = next_kept(c + 1, cdrop_next);   // (the scan overlaps the fetch in flight)
            const int c_fetch = c_next;
            double q[6];
#if SHM_TIER_LDS_FETCH
            {
                __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the fetch of this cluster has landed in LDS
                const double2 r0 = raw[wave][0][lane], r1 = raw[wave][1][lane], r2 = raw[wave][2][lane];
                q[0] = r0.x; q[1] = r0.y; q[2] = r1.x; q[3] = r1.y; q[4] = r2.x; q[5] = r2.y;
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the reads are done before the next fetch may overwrite the region
            }
#else
#pragma unroll
            for (int a = 0; a < 6; a++) q[a] = nq[a];
#endif
            if (c_fetch < P.n_clusters) fetch_cluster(c_fetch);
            // one source per lane: near / far / dropped for this wave's block of nodes
            unsigned long long nearmask, farmask;
            {
                // (the fp32 copies carry the weights relative to the largest one -- times P.wscale, a power of two: exact -- so that the exponent-range test
                // below and the packed-fp32 sums see weights in (0, 1] whatever the mesh's units; the fp64 list keeps the reference's own weights)
                float q32[6];
#pragma unroll
                for (int a = 0; a < 6; a++) q32[a] = a < 3 ? (float)q[a] : (float)(q[a] * P.wscale);
                // no node of the block is closer to the source than the source is to the block's box
                const float dx = fmaxf(fabsf(cx - q32[0]) - hx, 0.f), dy = fmaxf(fabsf(cy - q32[1]) - hy, 0.f), dz = fmaxf(fabsf(cz - q32[2]) - hz, 0.f);
                const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                const float w2 = q32[3] * q32[3] + q32[4] * q32[4] + q32[5] * q32[5];
                const float lw = 0.5f * __log2f(w2);                                       // log2(|w_s| / w_max) <= 0 (w = 0: -inf)
                const float rel = lw + 1e-5f - lnear_w;                                    // log2(|w_s| / |w_near|), rounded up
                const float lhs = (dist * 0.999999f - r_hi_w) * lam_l2;                    // lower bound of lambda (r(x, s) - r_near(x)) / ln 2
                const bool valid = w2 > 0.f;                                              // the zero-weight padding is never evaluated
                const bool in_range = fmaf(dist, lam_l2, range_c) <= lw;                   // every term of the source stays a normal fp32 number over the block
                float lhs_far = lhs, lhs_drop = lhs;
                {
                    // The drop rule needs the differential bound only where it can change a decision: it exceeds the box rule's bound by at most 2 rt lambda' (|r_s - d_s| and
                    // |r_hi - r_s*| are each at most a block radius), so a source the box rule already makes a candidate stays one, and one more than that slack above the
                    // threshold cannot become one.  Evaluated for the whole cluster when ANY lane sits in the window (wave-uniform branch): on inputs whose sources all
                    // matter (the bunny) almost never, and the fp32 solve -- whose far rule is not the differential one -- is back to its round-5 classification cost.
                    bool want_diff = use_diff;
                    if (!want_diff && drop_on && !cdrop_cur) {
                        const float lb_box = rel - lhs;
                        want_diff = __ballot(valid && lb_box > P.drop_ltau && lb_box <= P.drop_ltau + 2.02f * rt_w * lam_l2) != 0ull;
                    }
                    if (want_diff) {
                        // Differential rule (round 5, late; chosen per problem: Solver::far_rule_plan / far_rule_now): with s* the source nearest to the block's centre c,
                        //   r_s(x) - r_near(x) >= f(x) := r_s(x) - r_s*(x) >= f(c) - rt (|u_s(c) - u_s*(c)| + rt (1 / d_s + 1 / d_s*))      for every x of the block
                        // (u: unit vectors towards c; d: distances to the block's box -- the gradient of f is u_s - u_s*, a unit vector turns by at most |x - c| / d).  Two
                        // sources in similar directions keep their DIFFERENCE of distances over the block although each distance varies by the block's diameter: the
                        // box rule loses 2 rt lambda e-folds there.  More sources go to the packed-fp32 tier (bunny 256^3: 0.50 -> 0.60 of the pairs); the a-posteriori
                        // test below is what keeps the budget, whatever rule filled the tier.
                        const float ex = cx - q32[0], ey = cy - q32[1], ez = cz - q32[2];
                        const float dc2 = ex * ex + ey * ey + ez * ez, rdc = __builtin_amdgcn_rsqf(dc2), dc = dc2 * rdc;
                        const float dot = (ex * star_ux + ey * star_uy + ez * star_uz) * rdc;
                        const float du = sqrtf(fmaxf(0.f, 2.f - 2.f * dot));
                        const float lip = du + rt_w * (__builtin_amdgcn_rcpf(dist) + star_inv_d);
                        const float lhs2 = (dc * 0.999999f - star_dc - rt_w * lip * 1.00001f) * lam_l2;
                        lhs_drop = fmaxf(lhs, lhs2);   // (NaN -- the source at the centre -- leaves the box rule)
                        if (use_diff) lhs_far = lhs_drop;
                    }
                }
                const bool far = lhs_far > g_l2 + rel && in_range;
                // the drop rule by accumulated bound (see the block's header above): candidates of this cluster together, or only the ones below the hard threshold
                bool drop = valid && cdrop_cur;   // (pass 1 walks a cluster that pass 0 dropped as a whole)
                if (drop_on && !cdrop_cur) {
                    const float lb = rel - lhs_drop + 2e-5f;   // log2 of b_s without its geometric factor, rounded up
                    const bool cand = valid && lb <= P.drop_ltau;
                    if (__ballot(cand) != 0ull) {
                        float bs = cand ? __builtin_amdgcn_exp2f(lb) * (r_hi_w * __builtin_amdgcn_rcpf(dist)) * 1.0001f : 0.f;   // (a source inside the block: inf -- never dropped by the sum)
                        bs += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bs), 0x1f | (1 << 10)));   // butterfly sum within the halves of the wave: fixed order
                        bs += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bs), 0x1f | (2 << 10)));
                        bs += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bs), 0x1f | (4 << 10)));
                        bs += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bs), 0x1f | (8 << 10)));
                        bs += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(bs), 0x1f | (16 << 10)));
                        const float sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bs), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bs), 32));
                        const float r_new = uniform_f32(R_soft + sum);
                        if (__builtin_amdgcn_readfirstlane((int)(r_new <= P.drop_eps_soft))) {   // (inf / NaN: false)
                            R_soft = r_new;
                            drop = cand;
                        } else {
                            drop = cand && lb <= P.drop_ltau_hard;
                            R_hard = uniform_f32(R_hard + (float)__builtin_popcountll(__ballot(drop)) * P.drop_tau_hard);
                        }
                    }
                }
                // pass 1 evaluates what pass 0 dropped as well, where the term's exponent stays inside the span of the block's scale (beyond it the term is < 2^-900 of the scale)
                const bool span_ok = fmaf(dist, lam_l2, span_c) < 0.f;
                const bool to64 = valid && (pass == 0 ? (!drop && !far) : (drop ? span_ok : true));   // (drop first: a source outside the fp32 exponent range is not "far", but it may well be dropped)
                const bool to32 = valid && pass == 0 && far && !drop;
                nearmask = __ballot(to64);
                farmask = __ballot(to32);
                // stage the cluster for the broadcast reads below (wave-private region: no barrier), COMPACTED: the near sources in fp64 and the far ones in fp32
                // each as a dense list in mask order, so that the two loops below walk consecutive entries with a plain counter (a bit scan per source cost
                // ~10 scalar instructions on the wave's in-order instruction stream)
                if (to64) {
                    const int rnk = mask_rank(nearmask);
#pragma unroll
                    for (int a = 0; a < 6; a++) tile[rnk * 6 + a] = q[a];
                } else if (to32) {
                    const int rnk = mask_rank(farmask);
                    // positions in units of 1 / (lambda log2 e): the far loop then gets lambda r log2 e = d2' rsq(d2') without a multiplication of its own
                    *reinterpret_cast<float4*>(&tA[rnk * 4]) = float4{q32[0] * lam_l2, q32[1] * lam_l2, q32[3], q32[4]};
                    const float dz0 = qz0 - q32[2] * lam_l2;
                    float dz2[NPT];
#pragma unroll
                    for (int e = 0; e < NPT; e++) {
                        const float dz = e * cellq + dz0;   // (consecutive planes: the block's first scaled z, the others by the scaled cell size)
                        dz2[e] = dz * dz;
                    }
                    if constexpr (NPT == 4) *reinterpret_cast<float4*>(&tB[rnk * 4]) = float4{dz2[0], dz2[1], dz2[2], dz2[3]};
                    else *reinterpret_cast<float2*>(&tB[rnk * 2]) = float2{dz2[0], dz2[1]};
                    *reinterpret_cast<float2*>(&tC[rnk * 2]) = float2{q32[5], fabsf(q32[3]) + fabsf(q32[4]) + fabsf(q32[5])};
                }
            }
            const int nnear = __builtin_popcountll(nearmask), nfar = __builtin_popcountll(farmask);
            if (lane < kPad) {
#pragma unroll
                for (int a = 0; a < 6; a++) tile[(nnear + lane) * 6 + a] = a < 3 ? P.pad_pos[a] : 0.0;   // the padding entry: >= one grid side from every node, zero weight
                // (the far list holds SCALED positions; an unscaled padding point would land inside the grid, where coff - r' > 0 can overflow exp2f
                // and 0 * inf = NaN reaches the sums.  1e18 in the scaled x and y, 1e36 as the squared z offsets: d2 ~ 3e36 is finite, r' = 1.7e18 > coff always, 2^(coff - r') = 0)
                *reinterpret_cast<float4*>(&tA[(nfar + lane) * 4]) = float4{1.0e18f, 1.0e18f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < NPT; e++) tB[(nfar + lane) * NPT + e] = 1.0e36f;
                *reinterpret_cast<float2*>(&tC[(nfar + lane) * 2]) = float2{0.f, 0.f};
            }
            cnt_near += (unsigned)nnear;
            cnt_far += (unsigned)nfar;
            if (pass) cnt_redo += (unsigned)nnear;
            // one near source against the lane's z-column; breadth-first over the pairs in flight: every stage of e^{-lambda r}/r for all of them before the next stage, so
            // that the dependent chain of one pair (rsq -> Newton -> exponent -> table -> exponent insertion) is covered by the others' independent work
            auto near_source = [&](const double (&rec)[6]) {
                const double dx = px - rec[0], dy = py - rec[1];
                const double dxy2 = dx * dx + dy * dy;
                const double dz0 = pz0 - rec[2];
                double x[NPT], g[NPT];
#pragma unroll
                for (int e = 0; e < NPT; e++) {
                    const double dz = e ? dz0 + e * P.cell : dz0;   // (the block's nodes are consecutive planes: one z per lane, the others by the cell size)
                    x[e] = fma(dz, dz, dxy2);
                }
                constexpr int kB = SHM_TIER_NEAR_BATCH < NPT ? SHM_TIER_NEAR_BATCH : NPT;
#pragma unroll
                for (int b0 = 0; b0 < NPT; b0 += kB) yukawa_near_batch<kB>(x + b0, P.cexp, m1, exp_tab, g + b0);
#pragma unroll
                for (int e = 0; e < NPT; e++) {
                    ax[e] = fma(rec[3], g[e], ax[e]);
                    ay[e] = fma(rec[4], g[e], ay[e]);
                    az[e] = fma(rec[5], g[e], az[e]);
                }
            };
            // ---- near tier: fp64 ----
            for (int i0 = 0; i0 < nnear; i0++) {
                double rec[6];
#pragma unroll
                for (int a = 0; a < 6; a++) rec[a] = tile[6 * i0 + a];
                near_source(rec);
            }
            // ---- far tier: packed fp32 (two nodes per instruction), kFarUnroll sources in flight (the transcendentals' results arrive late) ----
            // (written stage by stage over the kFarUnroll x NPT / 2 packed pairs in flight: the order the scheduler starts from keeps every transcendental's consumer a full
            // stage behind it.  Round 6: with the per-source form the fp32 solve's loop came out with 24 s_nop of trans-use hazards -- 142 instructions where round 5's
            // build had 116 -- once unrelated code around it changed; this form does not depend on the scheduler finding the interleaving.)
            for (int i0f = 0; i0f < nfar; i0f += kFarUnroll) {
                constexpr int kP = kFarUnroll * (NPT / 2);
                float2v wxy[kFarUnroll], pcw[kFarUnroll], d2[kP], rinv[kP], g[kP];
                float dxy2[kFarUnroll];
#pragma unroll
                for (int u = 0; u < kFarUnroll; u++) {
                    const int s = i0f + u;
                    const float4 pa = *reinterpret_cast<const float4*>(&tA[4 * s]);
                    pcw[u] = *reinterpret_cast<const float2v*>(&tC[2 * s]);
                    wxy[u] = float2v{pa.z, pa.w};
                    const float dx = qx - pa.x, dy = qy - pa.y;   // (scaled coordinates: qx, qy and the staged positions are x lambda log2 e)
                    dxy2[u] = dx * dx + dy * dy;
                    if constexpr (NPT == 4) {
                        const float4 pb = *reinterpret_cast<const float4*>(&tB[4 * s]);
                        d2[2 * u] = float2v{pb.x, pb.y};
                        d2[2 * u + 1] = float2v{pb.z, pb.w};
                    } else {
                        d2[u] = *reinterpret_cast<const float2v*>(&tB[2 * s]);
                    }
                }
#pragma unroll
                for (int a = 0; a < kP; a++) d2[a] += float2v{dxy2[a / (NPT / 2)], dxy2[a / (NPT / 2)]};
#pragma unroll
                for (int a = 0; a < kP; a++) rinv[a] = float2v{__builtin_amdgcn_rsqf(d2[a].x), __builtin_amdgcn_rsqf(d2[a].y)};   // 1 / (lambda log2 e r)
#pragma unroll
                for (int a = 0; a < kP; a++) d2[a] = __builtin_elementwise_fma(-d2[a], rinv[a], float2v{coff, coff});              // -lambda log2 e (r - d0)
#pragma unroll
                for (int a = 0; a < kP; a++) g[a] = float2v{__builtin_amdgcn_exp2f(d2[a].x), __builtin_amdgcn_exp2f(d2[a].y)};
#pragma unroll
                for (int a = 0; a < kP; a++) g[a] *= rinv[a];
#pragma unroll
                for (int a = 0; a < kP; a++) {
                    const int u = a / (NPT / 2), h = a % (NPT / 2);
                    pk_fma_lo(fx[h], wxy[u], g[a]);
                    pk_fma_hi(fy[h], wxy[u], g[a]);
                    pk_fma_lo(fz[h], pcw[u], g[a]);
                    if constexpr (CHECK) pk_fma_hi(fl[h], pcw[u], g[a]);
                }
            }
            if constexpr (CHECK) {
                // the packed-fp32 sums go to the fp64 accumulators every P.tier_flush far sources (0: never): an fp32 accumulator that has taken N terms carries ~6e-8 sqrt(N) of its
                // partial sums, which the test's price of a far TERM does not know about
                far_pending += nfar;
                if (P.tier_flush > 0 && far_pending >= P.tier_flush) {
                    far_pending = 0;
                    const double e0f = far_scale();
#pragma unroll
                    for (int e = 0; e < NPT; e++) {
                        ax[e] = fma((double)fx[e / 2][e & 1], e0f, ax[e]);
                        ay[e] = fma((double)fy[e / 2][e & 1], e0f, ay[e]);
                        az[e] = fma((double)fz[e / 2][e & 1], e0f, az[e]);
                    }
#pragma unroll
                    for (int e = 0; e < NPT / 2; e++) fx[e] = fy[e] = fz[e] = float2v{0.f, 0.f};
                }
            }
            c = c_next;
            cdrop_cur = cdrop_next;
        }
        if (!CHECK) break;
        if (pass == 0) {
            // a-posteriori test of the far tier's contribution (see the header): eps_far L1_far <= budget |X| at every node of the block, or the far sources
            // -- near, far and dropped -- are evaluated again in fp64 (pass 1) and the sums of pass 0 discarded
            bool fail = false;
            const double e0 = far_scale();
            // what the block dropped, against |X| as well (round 6): its accumulated bound R is relative to the reference source's term at the node, evaluated here once per node
            // (in the block's scale, like the sums); R / eps_far puts it on the scale of the packed-fp32 tier's L1 sums
            // (fp32 is plenty for a bound that carries a 1 % margin, and it keeps the test out of the kernel's register peak)
            const float r_drop = R_soft + R_hard;
            // (everything here is invariant in the pass loop; the reference source comes back from LDS, the rest is pinned behind the loops by the empty asm -- computed ahead of
            // them it held 16 registers through them)
            const float4 st = star_stash[wave];
            float chk = P.drop_check, k0_here = (float)k0;
            asm volatile("" : "+v"(chk), "+v"(k0_here));
            const float rs = r_drop * chk * 1.01f * sqrtf(st.w) * __builtin_amdgcn_exp2f(-lws) * lam_l2;   // (lambda': the reciprocal below is of the SCALED distance)
            const float sdx = qx - st.x, sdy = qy - st.y, sdz0 = qz0 - st.z;   // scaled coordinates, like the far tier's
            // the price of a packed-fp32 term grows with its exponent (its scaled distance carries ~1e-7 u of rounding into 2^-u): beyond u = P.tier_u0 the block's far terms
            // -- at least coff + g_l2 powers of two below 1 -- are priced at eps_far u / u0 (a wave-uniform fp32 factor on the threshold: a scalar register; u0 = 0: flat)
            const float ratio_f = uniform_f32((float)P.far_redo_ratio * (P.tier_u0 > 0.f ? fminf(1.f, P.tier_u0 / (coff + g_l2)) : 1.f));
            const float sxy2 = sdx * sdx + sdy * sdy;
#pragma unroll
            for (int e = 0; e < NPT; e++) {
                const double x0 = ax[e] + (double)fx[e / 2][e & 1] * e0, x1 = ay[e] + (double)fy[e / 2][e & 1] * e0, x2 = az[e] + (double)fz[e / 2][e & 1] * e0;
                // |w_*| e^{-lambda r} / r in the block's scale 2^-k0:  2^(-lambda' r - k0) lambda' / (lambda' r)      (the node on s*: inf * 0 = NaN -- compares false, X is NaN there itself)
                const float dz = e * cellq + sdz0, d2 = fmaf(dz, dz, sxy2), rinv = __builtin_amdgcn_rsqf(d2);
                const float tstar = r_drop > 0.f ? rs * __builtin_amdgcn_exp2f(-d2 * rinv - k0_here) * rinv : 0.f;
                fail = fail || (live_xy && kk0 + e < P.kk_end && fma((double)fl[e / 2][e & 1], e0, (double)tstar) > (double)ratio_f * sqrt(x0 * x0 + x1 * x1 + x2 * x2));
            }
            if (__ballot(fail) == 0ull) break;
#pragma unroll
            for (int e = 0; e < NPT / 2; e++) fx[e] = fy[e] = fz[e] = float2v{0.f, 0.f};
            // (the fp64 accumulators hold flushed packed-fp32 sums: the second pass starts from nothing and takes every source in fp64)
#pragma unroll
            for (int e = 0; e < NPT; e++) ax[e] = ay[e] = az[e] = 0.;
        }
        }  // pass loop
        const double e0 = far_scale();
        // (the lane's node indices are formed again here rather than held through the loops: two registers)
        int lane_here = lane;
        asm volatile("" : "+v"(lane_here));
        const int si = min(i0 + (lane_here % kTierTX), n - 1), sj = min(j0 + (lane_here / kTierTX), n - 1);
#pragma unroll
        for (int e = 0; e < NPT; e++) {
            if (!(live_xy && kk0 + e < P.kk_end)) continue;
            // the block's scale back on (exact unless the reference's own sum underflows: 2^k0 ~ e^{-lambda d0} < 1e-308 only > 700 / lambda from every source)
            // (fp32 solve: the direction is formed in the block's own scale -- finite however far the block is from the sources, like the per-tile offset of the fp32 kernel)
            const int kback = sizeof(TY) == 8 ? k0 : 0;
            const double x0 = __builtin_amdgcn_ldexp(ax[e] + (double)fx[e / 2][e & 1] * e0, kback), x1 = __builtin_amdgcn_ldexp(ay[e] + (double)fy[e / 2][e & 1] * e0, kback),
                         x2 = __builtin_amdgcn_ldexp(az[e] + (double)fz[e / 2][e & 1] * e0, kback);
            const double nrm = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
            const size_t vi = (size_t)(kk0 + e) * plane + (size_t)sj * n + si;   // (live: kk0 + e is a plane of the launch)
            // 0/0 -> NaN exactly like X /= X.norm() (:61).  A wave writes 64-byte row segments (8 nodes: half lines); the x-adjacent block is a neighbouring
            // unit of the same XCD's queue, so the two halves meet in that XCD's L2: PMC, kernel alone (tools/conv_pmc.sh): 412 MB written for 384 MB of
            // output (non-temporal stores: 497 MB; 16 x 4 x NPT blocks: 387 MB but 7 % slower -- the wider block classifies fewer sources as far)
            Y0[vi] = (TY)(x0 / nrm);
            Y1[vi] = (TY)(x1 / nrm);
            Y2[vi] = (TY)(x2 / nrm);
        }
        if constexpr (CHECK) {
            if (rule_sel == 1 && lane == 0)   // a sample block reports: far pairs, pairs evaluated again, then that it is done; the last one to report publishes the verdict
                sample_report(P.sample_ctr, (cnt_far - far_before) * (unsigned long long)(64 * NPT), (cnt_redo - redo_before) * (unsigned long long)(64 * NPT), P.sample_blocks);
        }
    }  // unit loop
    if (counters && lane == 0) {
        atomicAdd(&counters[0], cnt_near * (unsigned long long)(64 * NPT));
        atomicAdd(&counters[1], cnt_far * (unsigned long long)(64 * NPT));
        if (cnt_redo) atomicAdd(&counters[2], cnt_redo * (unsigned long long)(64 * NPT));
    }
}

}  // namespace shm
