// Exact two-level (one-step nested dissection) inverse of G = A A^T for large constraint sets.
//
// The dense (A A^T)^-1 of the small cases costs m^3 flops to build and m^2 bytes per application (SprayBottle 1024^3, m = 48 893: 7.6 s
// of Gauss-Jordan and 9.6 GB per mat-vec).  G is sparse with the geometry of the source surface: rows = distinct source cells
// (signed_heat_grid_solver.cpp:80-98), two rows couple iff their cells share a grid node, i.e. the cells are 26-neighbours.  Cells
// whose i, j or k is a multiple of the box size b form a separator Sigma (planes one cell thick); the remaining cells fall into
// b^3-boxes whose interiors cannot touch each other.  With the rows ordered [interiors I_1..I_P | Sigma]:
//
//        G = [ D   E ]      D = blockdiag(D_a) (a few hundred rows each),   S = F - sum_a E_a^T D_a^-1 E_a  (|Sigma| ~ 3 m / b rows),
//            [ E^T F ]
//        u_I = t - T u_S ,  u_S = S^-1 (w_S - E^T t) ,  t = D^-1 w_I ,  T_a = D_a^-1 E_a .
//
// Set-up: batched blocked Gauss-Jordan of the D_a (all boxes at once), T_a for all boxes in one launch, the Schur update per colour (boxes of one colour -- box
// coordinates of equal parity -- touch disjoint separator rows, so eight passes need no atomics and the sums have a fixed order),
// then the existing blocked Gauss-Jordan on the |Sigma| x |Sigma| Schur complement: (3/b)^3 of the flops of the full inversion.
// Application, five launches (tl_rows_kernel: row-parallel mat-vecs with D_a^-1; tl_cols_kernel: column-parallel ones with E_a^T; tl_gather_sep_kernel; the dense S_Sigma^-1
// mat-vec; tl_finish_kernel: row-parallel mat-vecs with T_a).  A three-launch form ([t | y = T^T w] in one launch, the S^-1 mat-vec's workgroups gathering w_S - T^T w_I
// first) was built in round 5 behind SHM_TL_MERGED3 and measured slower (profiles/r05_projection.txt): five it stays.  Exact up to rounding, so the projector and the dual preconditioner are the same operators as with the dense
// inverse (same iteration counts); matrices in double for the projector, an fp32 copy for the dual preconditioner.
#pragma once
#include "shm_kernels.hip.h"

namespace shm {

struct TlBoxes {               // device views shared by the kernels below (all arrays indexed by box a unless noted)
    const int* ptrI;           // [P+1] interior slots of box a: [ptrI[a], ptrI[a+1])
    const int* ptrS;           // [P+1] separator columns of box a (slots of the y buffer)
    const size_t* offD;        // [P] offset of D_a (s_a x s_a, row-major)
    const size_t* offE;        // [P] offset of E_a and T_a (s_a x c_a, row-major)
    const int* rowsI;          // [nI] global row of an interior slot
    const int* colsS;          // [sum c_a] separator index (0..nS) of a box-local column
};

// Set-up kernels (round 4).  The D_a are stored padded to whole 64-row blocks (leading dimension ld_a = tl_ld(s_a), identity on the padded diagonal) and inverted
// ALL AT ONCE by the blocked Gauss-Jordan kernels of shm_kernels.hip.h in their batched form (GjBatch: 3 launches per 64 rows of the largest box).  Rounds 2-3
// inverted each box with one workgroup walking its matrix in global memory (s dependent steps of s^2 / 256 read-modify-writes per thread: 2 ms for the 126-row
// boxes of rocker at 512^3, 45 ms for the 280-row boxes of the same mesh at 256^3) and formed T_a and the Schur update with one workgroup per box as well
// (3.8 / 41 ms) -- together the 6 ms of set-up an fp32 solve of configs[2] waits for after its Step 1, 41 ms for configs[4], and 88 ms behind an 84 ms Step 1
// for rocker at 256^3 in fp64.
constexpr int kTlMaxBox = 2048;  // rows / separator columns of a box (the application kernels stage them in LDS)
__host__ __device__ __forceinline__ int tl_ld(int s) { return (s + 63) / 64 * 64; }

// T_a = D_a^-1 E_a for all boxes: a workgroup per (box, 16 rows of T), a wave per 4 rows, lanes = consecutive columns (coalesced rows of E)
constexpr int kTlRowsPerWg = 16;
static __global__ __launch_bounds__(kBlock) void tl_T_kernel(TlBoxes B, const int* __restrict__ chunkBox, const int* __restrict__ chunkRow, const double* __restrict__ Dinv,
                                                      const double* __restrict__ E, double* __restrict__ Tm, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);   // beside the tiered Step 1 (see gj_panels_kernel)
    const int a = chunkBox[blockIdx.x], r0 = chunkRow[blockIdx.x];
    const int s = B.ptrI[a + 1] - B.ptrI[a], c = B.ptrS[a + 1] - B.ptrS[a], ld = tl_ld(s);
    const double* Di = Dinv + B.offD[a];
    const double* Ea = E + B.offE[a];
    double* Ta = Tm + B.offE[a];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kR = kTlRowsPerWg / (kBlock / kWave);   // rows per wave
    for (int l = lane; l < c; l += kWave) {
        double acc[kR];
#pragma unroll
        for (int q = 0; q < kR; q++) acc[q] = 0.;
        for (int j = 0; j < s; j++) {
            const double e = Ea[(size_t)j * c + l];
#pragma unroll
            for (int q = 0; q < kR; q++) {
                const int i = r0 + wave * kR + q;
                acc[q] += (i < s ? Di[(size_t)i * ld + j] : 0.) * e;
            }
        }
#pragma unroll
        for (int q = 0; q < kR; q++) {
            const int i = r0 + wave * kR + q;
            if (i < s) Ta[(size_t)i * c + l] = acc[q];
        }
    }
}

// S[Sigma_a, Sigma_a] -= E_a^T T_a for the boxes of one colour (disjoint separator rows: plain read-modify-write): a workgroup per (box, 16 rows p of the
// update), a wave per 4 rows, lanes = consecutive columns q
static __global__ __launch_bounds__(kBlock) void tl_schur_kernel(TlBoxes B, const int* __restrict__ chunkBox, const int* __restrict__ chunkRow, const double* __restrict__ E,
                                                          const double* __restrict__ Tm, double* __restrict__ S, int ldS, int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int a = chunkBox[blockIdx.x], p0 = chunkRow[blockIdx.x];
    const int s = B.ptrI[a + 1] - B.ptrI[a], c = B.ptrS[a + 1] - B.ptrS[a];
    const double* Ea = E + B.offE[a];
    const double* Ta = Tm + B.offE[a];
    const int* cols = B.colsS + B.ptrS[a];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kR = kTlRowsPerWg / (kBlock / kWave);
    for (int q = lane; q < c; q += kWave) {
        double acc[kR];
#pragma unroll
        for (int u = 0; u < kR; u++) acc[u] = 0.;
        for (int i = 0; i < s; i++) {
            const double t = Ta[(size_t)i * c + q];
#pragma unroll
            for (int u = 0; u < kR; u++) {
                const int p = p0 + wave * kR + u;
                acc[u] += (p < c ? Ea[(size_t)i * c + p] : 0.) * t;
            }
        }
#pragma unroll
        for (int u = 0; u < kR; u++) {
            const int p = p0 + wave * kR + u;
            if (p < c) S[(size_t)cols[p] * ldS + cols[q]] -= acc[u];
        }
    }
}

// Application, step 1a (round 4):  t = D_a^-1 w_I  ->  tbuf, a WAVE per interior row over all boxes at once (rowBox: box of an interior slot).
// (Rounds 2-3 ran steps 1 and 4 with one workgroup per box: ~80 workgroups on 256 CUs for rocker at 512^3, 50 + 30 us per application -- 0.13 ms of
// projection per stencil-CG iteration, which held the loop of configs[2] at 0.64-0.68 of the HBM roofline; profiles/r04_bench_default.json.)
template <typename TM>
__global__ __launch_bounds__(kBlock) void tl_rows_kernel(TlBoxes B, const int* __restrict__ rowBox, int nI, const TM* __restrict__ Dinv, const double* __restrict__ w,
                                                         double* __restrict__ tbuf) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (i >= nI) return;
    const int a = rowBox[i], i0 = B.ptrI[a], s = B.ptrI[a + 1] - i0;
    const TM* Di = Dinv + B.offD[a] + (size_t)(i - i0) * tl_ld(s);
    const int* rows = B.rowsI + i0;
    double acc = 0.;
    for (int j = lane; j < s; j += kWave) acc += (double)Di[j] * w[rows[j]];
    acc = wave_sum(acc);
    if (lane == 0) tbuf[i] = acc;
}

// step 1b:  y = E_a^T t  ->  ybuf.  One workgroup per (box, chunk of 64 separator columns): lanes = consecutive columns (coalesced rows of E), the four
// waves split the box's rows and their partial sums meet in LDS in a fixed order.
template <typename TM>
__global__ __launch_bounds__(kBlock) void tl_cols_kernel(TlBoxes B, const int* __restrict__ chunkBox, const int* __restrict__ chunkCol, const TM* __restrict__ E,
                                                         const double* __restrict__ tbuf, double* __restrict__ ybuf) {
    __shared__ double part[kBlock / kWave][kWave];
    const int a = chunkBox[blockIdx.x], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = B.ptrI[a], s = B.ptrI[a + 1] - i0, c0 = B.ptrS[a], c = B.ptrS[a + 1] - c0;
    const int l = chunkCol[blockIdx.x] + lane;
    const TM* Ea = E + B.offE[a];
    double acc = 0.;
    if (l < c) {
        // four of the wave's rows in flight (round 5, late: one load per trip left the kernel at a row's memory latency per four rows of the box -- 14 us on an idle
        // device, 60 us beside the x update of the stencil CG, whose iteration then waited for the projection; profiles/r05_projection.txt)
        constexpr int W = kBlock / kWave;
        double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
        int i = wave;
        for (; i + 3 * W < s; i += 4 * W) {
            const double e0 = (double)Ea[(size_t)i * c + l], e1 = (double)Ea[(size_t)(i + W) * c + l], e2 = (double)Ea[(size_t)(i + 2 * W) * c + l],
                         e3 = (double)Ea[(size_t)(i + 3 * W) * c + l];
            a0 += e0 * tbuf[i0 + i];
            a1 += e1 * tbuf[i0 + i + W];
            a2 += e2 * tbuf[i0 + i + 2 * W];
            a3 += e3 * tbuf[i0 + i + 3 * W];
        }
        for (; i < s; i += W) a0 += (double)Ea[(size_t)i * c + l] * tbuf[i0 + i];
        acc = (a0 + a1) + (a2 + a3);
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && l < c) ybuf[c0 + l] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// Steps 1a + 1b in ONE launch (round 5; the projection of the stencil CG is a chain of short dependent launches, each worth 5-10 us beside a 0.28 ms sweep):
//   workgroups [0, grows):                t = D_a^-1 w_I, a wave per interior row (tl_rows_kernel's body);
//   workgroups [grows, grows + nChunks):  y = T_a^T w_I = E_a^T D_a^-1 w_I  (D_a is symmetric, T_a = D_a^-1 E_a is stored for step 4) -- from w, not from t, so the two
//                                         parts do not depend on each other (tl_cols_kernel's body with T for E).
// (A version whose last workgroup to arrive also gathered v_S was measured at 0.27 instead of 0.08 ms per projection: the device-scope fence every workgroup needs
// before it takes its ticket writes the L2 back each time.  v_S is gathered by the workgroups of the S^-1 mat-vec instead: tl_sep_matvec_kernel.)
template <typename TM>
__global__ __launch_bounds__(kBlock) void tl_rows_cols_kernel(TlBoxes B, const int* __restrict__ rowBox, int nI, int grows, const TM* __restrict__ Dinv, const int* __restrict__ chunkBox,
                                                              const int* __restrict__ chunkCol, const TM* __restrict__ Tm, const double* __restrict__ w, double* __restrict__ tbuf,
                                                              double* __restrict__ ybuf) {
    __shared__ double part[kBlock / kWave][kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < grows) {
        const int i = blockIdx.x * (kBlock / kWave) + wave;
        if (i >= nI) return;
        const int a = rowBox[i], i0 = B.ptrI[a], s = B.ptrI[a + 1] - i0;
        const TM* Di = Dinv + B.offD[a] + (size_t)(i - i0) * tl_ld(s);
        const int* rows = B.rowsI + i0;
        double acc = 0.;
        for (int j = lane; j < s; j += kWave) acc += (double)Di[j] * w[rows[j]];
        acc = wave_sum(acc);
        if (lane == 0) tbuf[i] = acc;
        return;
    }
    const int ch = (int)blockIdx.x - grows;
    const int a = chunkBox[ch];
    const int i0 = B.ptrI[a], s = B.ptrI[a + 1] - i0, c0 = B.ptrS[a], c = B.ptrS[a + 1] - c0;
    const int l = chunkCol[ch] + lane;
    const TM* Ta = Tm + B.offE[a];
    const int* rows = B.rowsI + i0;
    double acc = 0.;
    if (l < c)
        for (int i = wave; i < s; i += kBlock / kWave) acc += (double)Ta[(size_t)i * c + l] * w[rows[i]];
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && l < c) ybuf[c0 + l] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// Steps 2 + 3 in one launch: u_S = S^-1 v_S with v_S = w_S - sum of the y of the boxes that border the row gathered by the mat-vec's own workgroups -- each takes
// kTlSepRows rows of S^-1 and first builds ALL of v_S in LDS (nS values, a handful of loads each; the per-row sums then run in ginv_matvec_kernel's order).
constexpr int kTlSepRows = 8;
template <typename TM>
__global__ __launch_bounds__(kBlock) void tl_sep_matvec_kernel(int nS, int ld, const TM* __restrict__ Sinv, const int* __restrict__ sepRow, const int* __restrict__ adj_ptr,
                                                               const int* __restrict__ adj_idx, const double* __restrict__ w, const double* __restrict__ ybuf,
                                                               double* __restrict__ uS) {
    extern __shared__ double tl_v[];   // [ld]: v_S, zero in the padded tail
    __shared__ double lds[8];
    for (int g = threadIdx.x; g < ld; g += kBlock) {
        double v = 0.;
        if (g < nS) {
            v = w[sepRow[g]];
            for (int e = adj_ptr[g]; e < adj_ptr[g + 1]; e++) v -= ybuf[adj_idx[e]];
        }
        tl_v[g] = v;
    }
    __syncthreads();
    for (int q = 0; q < kTlSepRows; q++) {
        const int row = blockIdx.x * kTlSepRows + q;
        if (row >= nS) break;
        const TM* g = Sinv + (size_t)row * ld;
        double s = 0.;
        for (int c = threadIdx.x * 4; c < nS; c += kBlock * 4) {   // (ld is a multiple of 64: the vector loads stay inside the row; v_S is zero beyond nS)
            if (sizeof(TM) == 4) {
                const float4 gv = *reinterpret_cast<const float4*>(g + c);
                s += (double)gv.x * tl_v[c] + (double)gv.y * tl_v[c + 1] + (double)gv.z * tl_v[c + 2] + (double)gv.w * tl_v[c + 3];
            } else {
                const double2 g0 = *reinterpret_cast<const double2*>(g + c), g1 = *reinterpret_cast<const double2*>(g + c + 2);
                s += g0.x * tl_v[c] + g0.y * tl_v[c + 1] + g1.x * tl_v[c + 2] + g1.y * tl_v[c + 3];
            }
        }
        s = block_sum(s, lds);
        if (threadIdx.x == 0) uS[row] = s;
        __syncthreads();
    }
}

// step 2: v_S = w_S - E^T t, gathered per separator row from the boxes that border it (fixed order); the padded tail stays 0
static __global__ __launch_bounds__(kBlock) void tl_gather_sep_kernel(int nS, const int* __restrict__ sepRow, const int* __restrict__ adj_ptr, const int* __restrict__ adj_idx,
                                                               const double* __restrict__ w, const double* __restrict__ ybuf, double* __restrict__ vS) {
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= nS) return;
    double v = w[sepRow[g]];
    for (int e = adj_ptr[g]; e < adj_ptr[g + 1]; e++) v -= ybuf[adj_idx[e]];
    vS[g] = v;
}

// step 4: u_I = t - T_a u_S, a wave per interior row (workgroups [0, ceil(nI / 4))); u_S copied to its rows (the workgroups beyond)
template <typename TM>
__global__ __launch_bounds__(kBlock) void tl_finish_kernel(TlBoxes B, const int* __restrict__ rowBox, int nI, int nS, const int* __restrict__ sepRow, const TM* __restrict__ Tm,
                                                           const double* __restrict__ tbuf, const double* __restrict__ uS, double* __restrict__ u) {
    const int nrb = (nI + kBlock / kWave - 1) / (kBlock / kWave);
    if ((int)blockIdx.x >= nrb) {
        const int g = ((int)blockIdx.x - nrb) * kBlock + threadIdx.x;
        if (g < nS) u[sepRow[g]] = uS[g];
        return;
    }
    const int lane = threadIdx.x & 63, i = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (i >= nI) return;
    const int a = rowBox[i], i0 = B.ptrI[a], s = B.ptrI[a + 1] - i0, c0 = B.ptrS[a], c = B.ptrS[a + 1] - c0;
    (void)s;
    const TM* Ti = Tm + B.offE[a] + (size_t)(i - i0) * c;
    const int* cols = B.colsS + c0;
    double acc = 0.;
    for (int l = lane; l < c; l += kWave) acc += (double)Ti[l] * uS[cols[l]];
    acc = wave_sum(acc);
    if (lane == 0) u[B.rowsI[i]] = tbuf[i] - acc;
}

}  // namespace shm
