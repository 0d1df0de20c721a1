// The image-sum Green's table of the dual solver (shm_schur.hip.h) by FFT instead of by dense products (round 5).
//   reference: the operator being tabulated is K^+ of signed_heat_grid_solver.cpp:278-334 (laplacian()), see shm_schur.hip.h.
//
//     T(d1, d2, d3) = sum_k W0[k1][k2][k3] cos(pi k1 d1 / n) cos(pi k2 d2 / n) cos(pi k3 d3 / n),      k_a = 0 .. n-1,   d_a = 0 .. n
//
// is three passes of the 1-D transform  y[d] = sum_{k<n} w[k] cos(pi k d / n), d = 0 .. n, which rounds 2-4 applied as three dense products with the
// (n+1) x n cosine matrix: 6 n^4 flop on the fp64 matrix cores -- 1.2 ms at 256^3, 10 ms at 512^3, and because the fp64 matrix instruction shares the vector
// pipes' datapath (profiles/r02_mfma_f64_probe.txt) that is machine time taken from the Step-1 kernel it runs beside.  The transform splits by the parity of d:
//     y[2j]     = sum_k w[k] cos(2 pi k j / n)              = Re DFT_n(w)[j],                      j = 0 .. n/2
//     y[2j + 1] = sum_k w[k] cos(2 pi k j / n + pi k / n)   = Re DFT_n(w[k] e^{-i pi k / n})[j],   j = 0 .. n/2 - 1
// -- two complex FFTs of length n per PAIR of real lines a, b:  Z = DFT(a + i b) gives Re A[j] = (Re Z[j] + Re Z[n-j]) / 2, Re B[j] = (Im Z[j] + Im Z[n-j]) / 2
// (A, B spectra of real sequences), and Q = DFT((a + i b) omega), omega_k = e^{-i pi k / n}, gives the odd outputs the same way with the partner index n-1-j
// (DFT(a omega)[n-1-j] = conj DFT(a omega)[j] for real a).  O(n^3 log n) instead of O(n^4): the three passes move 0.8 GB at 256^3.
// The FFT itself is the Stockham core of the transform sweeps (shm_fft_core.h), here always with 4 complex lines (8 real lines) per tile: the table's last
// index is padded to n + 8, a multiple of 8 for every n = 2^k >= 16.
#pragma once
#include "shm_dct.hip.h"

namespace shm {

constexpr int kCosiLC = 4, kCosiL = 2 * kCosiLC;   // complex / real lines per tile

struct CosiParams {
    int ntiles, tiles_a;
    DctAddr in, out;       // element k of line l of tile t: see dct_addr (no segments)
};

template <int LOG2N, int R, int NS>
__device__ __forceinline__ void cosi_fft_pass(Cplx<double>* buf, const Cplx<double>* tw, int tid) {
    constexpr int items = PassGeom<LOG2N, R, kCosiLC>::items;
    constexpr int IPT = (items + kBlock - 1) / kBlock;
    Cplx<double> v[IPT][R];
#pragma unroll
    for (int a = 0; a < IPT; a++) {
        const int w = tid + a * kBlock;
        if (items % kBlock == 0 || w < items) pass_load<double, LOG2N, R, NS, -1, kCosiLC, true>(buf, tw, w, v[a]);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < IPT; a++) {
        const int w = tid + a * kBlock;
        if (items % kBlock == 0 || w < items) pass_store<double, LOG2N, R, NS, kCosiLC>(buf, w, v[a]);
    }
    __syncthreads();
}
template <int LOG2N>
__device__ __forceinline__ void cosi_fft(Cplx<double>* buf, const Cplx<double>* tw, int tid) {
    typedef FftPlan<LOG2N> P;
    cosi_fft_pass<LOG2N, P::R0, 1>(buf, tw, tid);
    if constexpr (P::npass > 1) cosi_fft_pass<LOG2N, P::R1, P::R0>(buf, tw, tid);
    if constexpr (P::npass > 2) cosi_fft_pass<LOG2N, P::R2, P::R0 * P::R1>(buf, tw, tid);
    if constexpr (P::npass > 3) cosi_fft_pass<LOG2N, P::R3, P::R0 * P::R1 * P::R2>(buf, tw, tid);
}

// XPASS: the elements of a line are contiguous (lanes run along k); otherwise consecutive LINES are contiguous (lanes run along the line pair: 16-byte accesses)
template <int LOG2N, bool XPASS>
__global__ __launch_bounds__(kBlock) void cosi_lines_kernel(CosiParams P, const double* __restrict__ in, double* __restrict__ out, const Cplx<double>* __restrict__ tw_g /* e^{-2 pi i t/n} */,
                                                            const Cplx<double>* __restrict__ om_g /* e^{-i pi k/n} */, int prio) {
    constexpr int n = 1 << LOG2N, ROW = kCosiLC + 1;
    __shared__ Cplx<double> buf[n * ROW];
    if (prio) __builtin_amdgcn_s_setprio(3);   // beside the tiered Step 1 (see gj_panels_kernel)
    const int tid = threadIdx.x;
    auto item = [&](int idx, int len, int& c, int& k) {   // work item -> (line pair c, element k) with the lanes along the contiguous direction
        if (XPASS) {
            k = idx % len;
            c = idx / len;
        } else {
            c = idx % kCosiLC;
            k = idx / kCosiLC;
        }
    };
    for (int t = blockIdx.x; t < P.ntiles; t += gridDim.x) {
        const long long bin = (long long)(t % P.tiles_a) * P.in.a_stride + (long long)(t / P.tiles_a) * P.in.b_stride;
        const long long bout = (long long)(t % P.tiles_a) * P.out.a_stride + (long long)(t / P.tiles_a) * P.out.b_stride;
#pragma unroll 1
        for (int odd = 0; odd < 2; odd++) {
            for (int idx = tid; idx < n * kCosiLC; idx += kBlock) {
                int c, k;
                item(idx, n, c, k);
                const double a = in[dct_addr(P.in, bin, 2 * c, k)], b = in[dct_addr(P.in, bin, 2 * c + 1, k)];
                Cplx<double> z{a, b};
                if (odd) z = cmul(z, om_g[k]);
                buf[k * ROW + c] = z;
            }
            __syncthreads();
            cosi_fft<LOG2N>(buf, tw_g, tid);
            const int nout = odd ? n / 2 : n / 2 + 1;
            for (int idx = tid; idx < nout * kCosiLC; idx += kBlock) {
                int c, j;
                item(idx, nout, c, j);
                const Cplx<double> z0 = buf[j * ROW + c], z1 = buf[(odd ? n - 1 - j : (n - j) & (n - 1)) * ROW + c];
                const int d = 2 * j + odd;
                out[dct_addr(P.out, bout, 2 * c, d)] = 0.5 * (z0.x + z1.x);
                out[dct_addr(P.out, bout, 2 * c + 1, d)] = 0.5 * (z0.y + z1.y);
            }
            __syncthreads();
        }
    }
}

}  // namespace shm
