// Explicit Schur complement S = A K^+ A^T of the dual solver (signed_heat_grid_solver.cpp:101-107 replaced by CG on S, see the block comment above
// dual_init_mu_kernel) for moderate constraint counts.
//
// Applying S through the grid costs five transform sweeps per iteration (256^3: 0.1 ms, 512^3: 0.8 ms) although p and S p live on m ~ 3000 rows.
// K^+ is the Neumann Green's function of the 7-point Laplacian, diagonal in the DCT-II basis, and the product formula
// cos a cos b = (cos(a - b) + cos(a + b)) / 2 turns its kernel into a sum of eight images of ONE table on the integer lattice:
//
//     K^+(x, y) = sum_{sigma in {0,1}^3} T(d^sigma),   d_a^0 = |x_a - y_a|,   d_a^1 = x_a + y_a + 1  (folded: d -> 2n - d beyond n),
//     T(d) = sum_{k != 0} prod_a [ gamma(k_a) cos(pi k_a d_a / n) ] / lambda_k,   gamma(0) = 1/(2n), gamma(k > 0) = 1/n,   d_a = 0 .. n.
//
// T is three dense cosine contractions of the symbol gamma gamma gamma / lambda (6 n^4 flop: fp64 matrix cores, dgemm_rm_kernel), built once per
// solve on the set-up stream beside Step 1.  A constraint row is a trilinear stencil (cell X_i, weights (1 - t_a, t_a) per axis), so
//
//     S_ij = sum over the 6 x 6 x 6 combinations of per-axis (index, weight) pairs of  w_x w_y w_z T[i_x][i_y][i_z],
//
// three pairs per axis for the difference image (|X_i - X_j + {-1, 0, 1}|) and three for the sum image (X_i + X_j + 1 + {0, 1, 2}): 216 table
// reads per entry, m (m + 1) / 2 entries (schur_assemble_kernel).  The dual iteration then needs one dense m x m mat-vec instead of scatter +
// five sweeps + gather: 256^3 (m = 2842) 0.17 -> 0.07 ms, 512^3 0.96 -> 0.07 ms per iteration.  Same operator up to rounding: same iteration counts.
#pragma once
#include "shm_kernels.hip.h"

namespace shm {

// W0[k1][k2][k3] = gamma gamma gamma / lambda_k  (0 for k = 0);  lam1[k] = (2 - 2 cos(pi k / n)) / h^2 is the table the transforms use.
// One (k1, k2) row of n entries per workgroup step, k3 across the threads: no 64-bit division by n per element (rounds 2-4 took 0.43 ms at 256^3 for 134 MB of
// output -- and that much of the vector pipes away from the Step-1 waves it runs beside).  The quotient itself is an IEEE division, as before.
static __global__ __launch_bounds__(kBlock) void green_symbol_kernel(int n, const double* __restrict__ lam1, double* __restrict__ W0) {
    const double g0 = 0.5 / n, g1 = 1.0 / n;
    for (int row = blockIdx.x; row < n * n; row += gridDim.x) {
        const int k1 = row / n, k2 = row - k1 * n;
        const double l12 = lam1[k1] + lam1[k2];
        const double g12 = (k1 ? g1 : g0) * (k2 ? g1 : g0);
        double* const out = W0 + (size_t)row * n;
        for (int k3 = threadIdx.x; k3 < n; k3 += kBlock) {
            const double lam = l12 + lam1[k3];
            const double g = g12 * (k3 ? g1 : g0);
            out[k3] = (row | k3) ? g / lam : 0.;
        }
    }
}

// Cm[d][k] = cos(pi k d / n)  ((n + 1) x n, leading dimension n)  and  Ct[k][d]  (n x (n + 1), leading dimension P) from the 2n values
// ctab[r] = cos(pi r / n): the argument is reduced exactly in integers
static __global__ __launch_bounds__(kBlock) void cosine_tables_kernel(int n, int P, const double* __restrict__ ctab, double* __restrict__ Cm, double* __restrict__ Ct) {
    const int total = (n + 1) * n;
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < total; v += gridDim.x * kBlock) {
        const int d = v / n, k = v - d * n;
        const double c = ctab[(int)(((long long)d * k) % (2 * n))];
        Cm[(size_t)d * n + k] = c;
        Ct[(size_t)k * P + d] = c;
    }
}

// C (M x N, ldc) = A (M x K, lda) B (K x N, ldb), row-major, batched over blockIdx.z with element strides sA, sB, sC; any M, N, K (edges are
// zero-filled / masked).  128 x 128 tile per workgroup, one 64 x 64 quadrant per wave as 4 x 4 v_mfma_f64_16x16x4_f64 accumulators, K in LDS chunks
// of 16 (operand layout of gj_update_kernel), the next chunk's operands prefetched into registers under the matrix instructions: 41 TFLOP/s at n = 512
// (a 64 x 64-tile version without the prefetch: 13).
// WN = 4: the shape above.  WN = 1: 128 x 32 tiles, a 64 x 16 quadrant per wave -- a quarter of the accumulators; the whole kernel stays under 128 registers,
// so that its workgroups fit on a SIMD beside two waves of the tiered fp64 Step-1 kernel (184 registers each: 144 are left) and the table is built WHILE
// Step 1 runs rather than in the gaps between its launches.  (WN = 2, 160 registers, served the first, 176-register version of that kernel.)
constexpr int kGemmT = 128, kGemmK = 16;
template <int WN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WN == 4 ? 2 : WN == 2 ? 3 : 4, WN == 4 ? 2 : WN == 2 ? 3 : 4))) void dgemm_rm_kernel(int M, int N, int K, const double* __restrict__ A, int lda, long long sA, const double* __restrict__ B, int ldb,
                                                          long long sB, double* __restrict__ C, int ldc, long long sC, int prio) {
    constexpr int kTN = 32 * WN;                // columns of the workgroup tile
    if (prio) __builtin_amdgcn_s_setprio(3);              // (see gj_panels_kernel)
    __shared__ double as[kGemmT][kGemmK + 1];   // A chunk [i][k]
    __shared__ double bs[kGemmK][kTN + 1];      // B chunk [k][j]
    A += (long long)blockIdx.z * sA;
    B += (long long)blockIdx.z * sB;
    C += (long long)blockIdx.z * sC;
    const int i0 = blockIdx.y * kGemmT, j0 = blockIdx.x * kTN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    gj_f64x4 acc[4][WN];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < WN; b++) acc[a][b] = gj_f64x4{0., 0., 0., 0.};
    // the next chunk's operands travel from global memory into registers while the matrix cores work on the current one
    constexpr int kPerA = kGemmT * kGemmK / kBlock, kPerB = kTN * kGemmK / kBlock;   // elements of each operand per thread and chunk
    double ra[kPerA], rb[kPerB];
    auto fetch = [&](int kc) {
#pragma unroll
        for (int u = 0; u < kPerA; u++) {
            const int t = threadIdx.x + u * kBlock;
            const int k = t & (kGemmK - 1), i = t >> 4;   // 16 consecutive k of one row of A
            ra[u] = (i0 + i < M && kc + k < K) ? A[(size_t)(i0 + i) * lda + kc + k] : 0.;
        }
#pragma unroll
        for (int u = 0; u < kPerB; u++) {
            const int t = threadIdx.x + u * kBlock;
            const int j = t & (kTN - 1), k2 = t / kTN;  // kTN consecutive columns of one row of B
            rb[u] = (kc + k2 < K && j0 + j < N) ? B[(size_t)(kc + k2) * ldb + j0 + j] : 0.;
        }
    };
    fetch(0);
    for (int kc = 0; kc < K; kc += kGemmK) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPerA; u++) {
            const int t = threadIdx.x + u * kBlock;
            as[t >> 4][t & (kGemmK - 1)] = ra[u];
        }
#pragma unroll
        for (int u = 0; u < kPerB; u++) {
            const int t = threadIdx.x + u * kBlock;
            bs[t / kTN][t & (kTN - 1)] = rb[u];
        }
        __syncthreads();
        if (kc + kGemmK < K) fetch(kc + kGemmK);
#pragma unroll
        for (int kk = 0; kk < kGemmK; kk += 4) {
            double af[4], bf[WN];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = as[wr * 64 + a * 16 + l15][kk + l4];
#pragma unroll
            for (int b = 0; b < WN; b++) bf[b] = bs[kk + l4][wc * 16 * WN + b * 16 + l15];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < WN; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < WN; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = i0 + wr * 64 + a * 16 + l4 + 4 * r, col = j0 + wc * 16 * WN + b * 16 + l15;
                if (row < M && col < N) C[(size_t)row * ldc + col] = acc[a][b][r];
            }
}

// S[i][j] = S[j][i] = sum over the eight images (difference or sum window per axis) of sum_{p, q, r < 3} w_x w_y w_z T[i_x[p]][i_y[q]][i_z[r]]
// (T: (n + 1) x (n + 1) x P, last index fastest).  One thread per entry of a 16 x 16 tile of the Morton-sorted rows (neighbouring cells: neighbouring
// table entries, served by L1 / L2 -- staging a tile's index windows in LDS first was measured and is no faster: 1.79 against 1.76 ms at m = 2842);
// tiles below the diagonal are left to their mirror images.
__device__ __forceinline__ int schur_fold(int e, int n) { return e <= n ? e : 2 * n - e; }
typedef double schur_f64x2 __attribute__((ext_vector_type(2), aligned(8)));
static __global__ __launch_bounds__(kBlock) void schur_assemble_kernel(int m, int ld, int n, int P, const int* __restrict__ rowX /* [m][4]: cell i, j, k and the row index, in Morton order of the cells */,
                                                                const double* __restrict__ rowT /* [m][3] */, const double* __restrict__ T, double* __restrict__ S, int prio) {
    if (blockIdx.x < blockIdx.y) return;
    if (prio) __builtin_amdgcn_s_setprio(3);   // runs on the SIMDs Step 1 occupies (see gj_panels_kernel)
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4), j = blockIdx.x * 16 + (threadIdx.x & 15);
    if (i >= m || j >= m || j < i) return;   // (diagonal tiles: the upper entry writes its mirror image too, so S is exactly symmetric)
    // Round 5 (late): everything that depends on ONE axis and its image flag is formed once per entry -- the three table indices of the x and y windows as
    // element offsets (24-bit multiplications: indices <= n + 1 <= 1025, row pitches < 2^24), and for the z window its first index and the weights of the three
    // consecutive table entries [zb, zb + 2] it covers (a window |D - 1|, |D|, |D + 1| or its folded counterpart: ascending, descending, or folded onto two
    // entries) -- instead of once per image and term.  The image loop is unrolled, so every index below is static: 2000 -> 700 vector instructions per entry, no
    // scratch (the weights used to be indexed by the image's flags), no per-term selects.  The kernel runs beside Step 1, which loses what it issues
    // (profiles/r05_schur_assembly.txt).  Sums are formed in a different order than in rounds 3-4 (last-bit differences in S).
    unsigned ox[2][3], oy[2][3], oz[2];    // [image flag][window term]
    double wx[2][3], wy[2][3], wz[2][3];   // wz: per table entry zb + s, not per window term
    const unsigned pitch_y = (unsigned)P, pitch_x = (unsigned)(n + 1) * (unsigned)P;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int Xi = rowX[4 * i + a], Xj = rowX[4 * j + a];
        const double ti = rowT[3 * i + a], tj = rowT[3 * j + a];
        const double wi0 = 1. - ti, wi1 = ti, wj0 = 1. - tj, wj1 = tj;
        const int D = Xi - Xj, E = Xi + Xj + 1;
        const double wd[3] = {wi0 * wj1, wi0 * wj0 + wi1 * wj1, wi1 * wj0};   // difference window, terms |D - 1|, |D|, |D + 1|
        const double ws[3] = {wi0 * wj0, wi0 * wj1 + wi1 * wj0, wi1 * wj1};   // sum window, terms fold(E), fold(E + 1), fold(E + 2)
        int pd[3], ps[3];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            pd[e] = abs(D - 1 + e);
            ps[e] = schur_fold(E + e, n);
        }
        if (a == 0) {
#pragma unroll
            for (int e = 0; e < 3; e++) {
                ox[0][e] = __umul24((unsigned)pd[e], pitch_x);
                ox[1][e] = __umul24((unsigned)ps[e], pitch_x);
                wx[0][e] = wd[e];
                wx[1][e] = ws[e];
            }
        } else if (a == 1) {
#pragma unroll
            for (int e = 0; e < 3; e++) {
                oy[0][e] = __umul24((unsigned)pd[e], pitch_y);
                oy[1][e] = __umul24((unsigned)ps[e], pitch_y);
                wy[0][e] = wd[e];
                wy[1][e] = ws[e];
            }
        } else {
#pragma unroll
            for (int f = 0; f < 2; f++) {
                const int* pz = f ? ps : pd;
                const double* wv = f ? ws : wd;
                const int zb = min(pz[0], min(pz[1], pz[2]));
                oz[f] = (unsigned)zb;
#pragma unroll
                for (int sl = 0; sl < 3; sl++) {
                    double w = 0.;
#pragma unroll
                    for (int e = 0; e < 3; e++) w += pz[e] - zb == sl ? wv[e] : 0.;
                    wz[f][sl] = w;
                }
            }
        }
    }
    double acc = 0.;
#pragma unroll
    for (int combo = 0; combo < 8; combo++) {   // image (sigma_x, sigma_y, sigma_z): 27 terms each
        const int fx = combo & 1, fy = (combo >> 1) & 1, fz = combo >> 2;
        const double* base = T + oz[fz];
        double sc = 0.;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            double sp = 0.;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                // the three entries along the last axis lie in [zb, zb + 2]: two loads (16 + 8 bytes, 8-byte aligned) instead of three -- the kernel is bound by the
                // cache lines its loads touch per instruction (round 4: 1.8 -> 1.1 ms at m = 2842, profiles/r04_schur_assemble.txt); rows are P = n + 8 long, so
                // zb + 2 stays inside the row
                const double* row = base + (ox[fx][p] + oy[fy][q]);
                const schur_f64x2 lo = *reinterpret_cast<const schur_f64x2*>(row);
                const double hi = row[2];
                sp += wy[fy][q] * (wz[fz][0] * lo.x + wz[fz][1] * lo.y + wz[fz][2] * hi);
            }
            sc += wx[fx][p] * sp;
        }
        acc += sc;
    }
    const int ri = rowX[4 * i + 3], rj = rowX[4 * j + 3];
    S[(size_t)ri * ld + rj] = acc;
    S[(size_t)rj * ld + ri] = acc;
}

// M[i][i] = v for i in [i0, i1)  (identity tail of a padded SPD matrix)
static __global__ __launch_bounds__(kBlock) void set_diagonal_kernel(double* __restrict__ M, int ld, int i0, int i1, double v) {
    const int i = i0 + blockIdx.x * kBlock + threadIdx.x;
    if (i < i1) M[(size_t)i * ld + i] = v;
}

// Direct dual solve, bordered system [[S, 1], [1^T, 0]] [mu; c] = [rhs; sigma]:  with u = S^-1 rhs and v = S^-1 1,  c = (1^T u - sigma) / (1^T v),
// mu (+)= u - c v.  sigma = *sigma_ptr (nullptr: 0).  One workgroup.
static __global__ __launch_bounds__(kDualBlock) void dual_bordered_kernel(int m, const double* __restrict__ u, const double* __restrict__ v, const double* __restrict__ sigma_ptr,
                                                                  int accumulate, double* __restrict__ mu) {
    __shared__ double lds[17];
    double su = 0., sv = 0.;
    for (int a = threadIdx.x; a < m; a += kDualBlock) {
        su += u[a];
        sv += v[a];
    }
    su = block_sum_1024(su, lds);
    sv = block_sum_1024(sv, lds);
    const double c = (su - (sigma_ptr ? *sigma_ptr : 0.)) / sv;
    for (int a = threadIdx.x; a < m; a += kDualBlock) mu[a] = (accumulate ? mu[a] : 0.) + u[a] - c * v[a];
}

static __global__ __launch_bounds__(kBlock) void fill_kernel(double* __restrict__ x, int count, double v) {
    const int a = blockIdx.x * kBlock + threadIdx.x;
    if (a < count) x[a] = v;
}

}  // namespace shm
