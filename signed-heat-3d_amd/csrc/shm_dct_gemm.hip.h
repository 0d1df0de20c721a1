// Fast Poisson solve K^+ v for grids whose side is NOT a power of two (signed_heat_grid_solver.cpp:24: nx = (size_t)(2 * 2^(hCoef + 3)) with a float
// hCoef the demo's slider edits freely, src/main.cpp:146-148 -- hCoef 4.5 gives n = 362 = 2 * 181).
//
// K = -L (the 7-point Neumann graph Laplacian, :278-334) is diagonal in the 3-D DCT-II basis for EVERY n; only the O(n log n) line transforms of
// shm_dct.hip.h need n = 2^k.  For the other sizes the three forward and three inverse line transforms are applied as dense products with the
// orthonormal DCT-II matrix C (n x n) on the fp64 matrix cores (dgemm_rm_kernel, shm_schur.hip.h):
//     x:  W[(k,j)][a] = sum_i X[(k,j)][i] C[a][i]          one n^2 x n x n product (B = C^T)
//     y:  W[k][b][a]  = sum_j C[b][j] W[k][j][a]           n products n x n x n (batched over k)
//     z:  W[c][(b,a)] = sum_k C[c][k] W[k][(b,a)]          one n x n^2 x n product
//     scale by 1 / (lam_a + lam_b + lam_c)  (0 for the constant mode: the pseudo-inverse), then the same three with C^T.
// 12 n^4 flop per application (n = 362: 2.1e11, ~5 ms at the measured 41 TFLOP/s of the kernel) against 10 N T bytes for the FFT sweeps -- 20x the
// arithmetic, but independent of the factorisation of n (181 is prime) and still a small part of a solve whose Step 1 costs N S pair evaluations
// (n = 362, bunny_small: ~75 ms).  Same operator to rounding: test_preconditioner_is_the_dct_pseudo_inverse covers both paths.
#pragma once
#include "shm_kernels.hip.h"

namespace shm {

// Cm[k][i] = s_k cos(pi k (2 i + 1) / (2 n)), s_0 = sqrt(1/n), s_k = sqrt(2/n)  (orthonormal DCT-II: Cm Cm^T = I) and its transpose Ct[i][k];
// ctab[r] = cos(pi r / (2 n)), r < 4 n, computed on the host: the argument is reduced exactly in integers
static __global__ __launch_bounds__(kBlock) void dct_matrix_kernel(int n, const double* __restrict__ ctab, double* __restrict__ Cm, double* __restrict__ Ct) {
    const int total = n * n;
    const double s0 = sqrt(1.0 / n), s1 = sqrt(2.0 / n);
    for (int v = blockIdx.x * kBlock + threadIdx.x; v < total; v += gridDim.x * kBlock) {
        const int k = v / n, i = v - k * n;
        const double c = (k ? s1 : s0) * ctab[(int)(((long long)k * (2 * i + 1)) % (4 * n))];
        Cm[(size_t)k * n + i] = c;
        Ct[(size_t)i * n + k] = c;
    }
}

// W[c][b][a] *= 1 / (lam[a] + lam[b] + lam[c]); the constant mode (a = b = c = 0) -> 0
static __global__ __launch_bounds__(kBlock) void spectral_scale_kernel(int n, const double* __restrict__ lam, double* __restrict__ W) {
    const size_t N = (size_t)n * n * n;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < N; v += (size_t)gridDim.x * kBlock) {
        const int a = (int)(v % n), b = (int)((v / n) % n), c = (int)(v / ((size_t)n * n));
        const double l = lam[a] + lam[b] + lam[c];
        W[v] = (a | b | c) ? W[v] / l : 0.;
    }
}

// partial sums of u . v over `count` elements (fixed order: deterministic), one partial per workgroup
template <typename T>
__global__ __launch_bounds__(kBlock) void dot_partial_kernel(size_t count, const T* __restrict__ u, const T* __restrict__ v, double* __restrict__ partials) {
    __shared__ double red[8];
    double acc = 0.;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock) acc += (double)u[i] * (double)v[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

}  // namespace shm
