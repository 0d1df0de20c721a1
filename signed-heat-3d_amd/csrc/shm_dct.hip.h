// Fast Poisson preconditioner for the projected CG: M^-1 = C3^T D C3, the exact pseudo-inverse of the
// reference's 7-point Neumann Laplacian K = -L (laplacian(): signed_heat_grid_solver.cpp:278-334), which the
// 3-D DCT-II diagonalises (eigenvalues (2-2cos(pi k/n))/h^2 per axis).  Used sandwiched between two
// constraint projections, P M^-1 P (SURVEY 7.3 "preconditioning"): the answer of the KKT system is unchanged,
// only the iteration count drops (~25-30x at 128^3..256^3).
//
// One kernel template does a batch of 1-D transforms along any axis, in LDS:
//   tile = L real lines (two real lines share one complex FFT), radix-2 in place, 256 threads;
//   DCT-II  = Makhoul permutation on load -> DIT FFT -> twiddle e^{-i pi k/2n}        (mode FWD)
//   DCT-III = pair pre-twiddle -> DIF inverse FFT -> inverse permutation on store     (mode INV)
//   FUSED   = FWD, spectral scaling D(kx,ky,kz), INV without leaving LDS (the z axis)
// Each pass reads and writes every element once: x-fwd, y-fwd, z-fused, y-inv, x-inv = 10 array sweeps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shm_kernels.hip.h"

namespace shm {

template <typename TP> struct Cplx { TP x, y; };
template <typename TP> __device__ __forceinline__ Cplx<TP> cmul(Cplx<TP> a, Cplx<TP> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename TP> __device__ __forceinline__ Cplx<TP> cadd(Cplx<TP> a, Cplx<TP> b) { return {a.x + b.x, a.y + b.y}; }
template <typename TP> __device__ __forceinline__ Cplx<TP> csub(Cplx<TP> a, Cplx<TP> b) { return {a.x - b.x, a.y - b.y}; }
template <typename TP> __device__ __forceinline__ Cplx<TP> cconj(Cplx<TP> a) { return {a.x, -a.y}; }

enum DctMode : int { DCT_FWD = 0, DCT_INV = 1, DCT_FUSED = 2 };
constexpr int kDctLines = 16;  // real lines per tile -> 8 complex FFTs (the index math below hard-codes 16 / 8)

struct DctParams {
    int n, log2n;
    int l_fastest;            // 1: consecutive threads take consecutive lines (y/z passes); 0: consecutive elements (x pass)
    long long elem_stride;    // distance between consecutive elements of a line
    long long line_stride;    // distance between consecutive lines of a tile
    int tiles_a;              // tile t -> base = (t % tiles_a) * a_stride + (t / tiles_a) * b_stride
    long long a_stride, b_stride;
    long long in_off, out_off;  // element offsets added to the in/out pointers (ghost plane of the CG vectors)
    // FUSED only: spectral coordinates of line l of tile t: kx = (t % tiles_a)*L + l ; ky = t / tiles_a
    double inv_n3_8;          // 8/n^3 (product of the three 2/n normalisations; k=0 factors handled per axis)
};

// lam[k] = (2-2cos(pi k/n))/h^2 ; tw[k] = e^{-2 pi i k/n}, k<n/2 ; om[k] = e^{-i pi k/(2n)}, k<n
template <typename TP, typename TIn, typename TOut, int MODE, bool DOT>
__global__ __launch_bounds__(kBlock) void dct_lines_kernel(DctParams P, const TIn* __restrict__ in, TOut* __restrict__ out,
                                                           const Cplx<TP>* __restrict__ tw_g, const Cplx<TP>* __restrict__ om_g,
                                                           const TP* __restrict__ lam_g, const TOut* __restrict__ dot_with,
                                                           double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int L = kDctLines, LC = L / 2;
    const int n = P.n, lg = P.log2n, half_n = n >> 1;
    Cplx<TP>* buf = reinterpret_cast<Cplx<TP>*>(smem);       // [n][LC]
    Cplx<TP>* tw = buf + (size_t)n * LC;                      // [n/2]
    double* red = reinterpret_cast<double*>(tw + half_n);     // [8] block-reduction scratch (all LDS in the dynamic region)
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    const long long base = (long long)(t % P.tiles_a) * P.a_stride + (long long)(t / P.tiles_a) * P.b_stride;
    for (int a = tid; a < half_n; a += kBlock) tw[a] = tw_g[a];

    // ---------------- load ----------------
    const int total = n * L;
    if (MODE == DCT_FWD || MODE == DCT_FUSED) {
        // Makhoul: v[j/2] = x[j] (j even), v[n-1-(j-1)/2] = x[j] (j odd); stored at the bit-reversed slot for the DIT FFT
        for (int idx = tid; idx < total; idx += kBlock) {
            int l, j;
            if (P.l_fastest) { l = idx & (L - 1); j = idx >> 4; } else { j = idx & (n - 1); l = idx >> lg; }
            const TP v = (TP)in[P.in_off + base + (long long)l * P.line_stride + (long long)j * P.elem_stride];
            const int pj = (j & 1) ? (n - 1 - (j >> 1)) : (j >> 1);
            const int slot = (int)(__brev((unsigned)pj) >> (32 - lg));
            TP* dst = reinterpret_cast<TP*>(&buf[slot * LC + (l >> 1)]);
            dst[l & 1] = v;
        }
    } else {
        for (int idx = tid; idx < total; idx += kBlock) {
            int l, j;
            if (P.l_fastest) { l = idx & (L - 1); j = idx >> 4; } else { j = idx & (n - 1); l = idx >> lg; }
            const TP v = (TP)in[P.in_off + base + (long long)l * P.line_stride + (long long)j * P.elem_stride];
            TP* dst = reinterpret_cast<TP*>(&buf[j * LC + (l >> 1)]);
            dst[l & 1] = v;  // raw spectrum X_a[k], X_b[k] side by side
        }
    }
    __syncthreads();

    const int nbf = half_n * LC;  // butterflies per stage
    // ---------------- forward FFT (DIT, bit-reversed input, natural output) ----------------
    if (MODE == DCT_FWD || MODE == DCT_FUSED) {
        for (int s = 1; s <= lg; s++) {
            const int half = 1 << (s - 1);
            const int tstep = n >> s;
            for (int b = tid; b < nbf; b += kBlock) {
                const int c = b & (LC - 1), bf = b >> 3;
                const int pos = bf & (half - 1);
                const int i0 = ((bf >> (s - 1)) << s) + pos, i1 = i0 + half;
                const Cplx<TP> w = tw[pos * tstep];
                const Cplx<TP> u = buf[i0 * LC + c], v = cmul(w, buf[i1 * LC + c]);
                buf[i0 * LC + c] = cadd(u, v);
                buf[i1 * LC + c] = csub(u, v);
            }
            __syncthreads();
        }
    }

    // ---------------- spectral step on (k, n-k) pairs ----------------
    if (MODE == DCT_FWD) {
        // X[k] = Re(om_k V[k]); V_a = (Z_k + conj Z_{n-k})/2, V_b = -i (Z_k - conj Z_{n-k})/2 ; write straight to global
        for (int idx = tid; idx < total; idx += kBlock) {
            int l, k;
            if (P.l_fastest) { l = idx & (L - 1); k = idx >> 4; } else { k = idx & (n - 1); l = idx >> lg; }
            const int c = l >> 1;
            const Cplx<TP> zk = buf[k * LC + c], zn = cconj(buf[((n - k) & (n - 1)) * LC + c]);
            Cplx<TP> v;
            if ((l & 1) == 0) v = {(TP)0.5 * (zk.x + zn.x), (TP)0.5 * (zk.y + zn.y)};
            else v = {(TP)0.5 * (zk.y - zn.y), (TP)-0.5 * (zk.x - zn.x)};
            const Cplx<TP> om = om_g[k];
            out[P.out_off + base + (long long)l * P.line_stride + (long long)k * P.elem_stride] = (TOut)(om.x * v.x - om.y * v.y);
        }
        return;
    }
    if (MODE == DCT_FUSED || MODE == DCT_INV) {
        // pairs (k, n-k), k = 0..n/2 ; builds Z'[k] = H_a[k] + i H_b[k], H[k] = 1/2 conj(om_k) (X[k] - i X[n-k]), H[0] = X[0]
        const int kx0 = (t % P.tiles_a) * L, ky = t / P.tiles_a;
        for (int b = tid; b < (half_n + 1) * LC; b += kBlock) {
            const int c = b & (LC - 1), k = b >> 3;
            const int nk = (n - k) & (n - 1);
            TP xa_k, xb_k, xa_n, xb_n;
            if (MODE == DCT_FUSED) {
                const Cplx<TP> zk = buf[k * LC + c], znk = buf[nk * LC + c];
                const Cplx<TP> cz = cconj(znk), czk = cconj(zk);
                const Cplx<TP> omk = om_g[k], omn = om_g[nk];
                // spectrum at k
                const Cplx<TP> va = {(TP)0.5 * (zk.x + cz.x), (TP)0.5 * (zk.y + cz.y)}, vb = {(TP)0.5 * (zk.y - cz.y), (TP)-0.5 * (zk.x - cz.x)};
                xa_k = omk.x * va.x - omk.y * va.y;
                xb_k = omk.x * vb.x - omk.y * vb.y;
                // spectrum at n-k
                const Cplx<TP> wa = {(TP)0.5 * (znk.x + czk.x), (TP)0.5 * (znk.y + czk.y)}, wb = {(TP)0.5 * (znk.y - czk.y), (TP)-0.5 * (znk.x - czk.x)};
                xa_n = omn.x * wa.x - omn.y * wa.y;
                xb_n = omn.x * wb.x - omn.y * wb.y;
                // D(kx,ky,kz) = sx sy sz / (lam_x + lam_y + lam_z), s_0 = 1/n, s_k = 2/n ; zero mode -> 0
                const int kxa = kx0 + 2 * c, kxb = kxa + 1;
                const TP lxy_a = lam_g[kxa] + lam_g[ky], lxy_b = lam_g[kxb] + lam_g[ky];
                const TP sy = ky == 0 ? (TP)0.5 : (TP)1;
                const TP sa = (kxa == 0 ? (TP)0.5 : (TP)1) * sy * (TP)P.inv_n3_8, sb = (kxb == 0 ? (TP)0.5 : (TP)1) * sy * (TP)P.inv_n3_8;
                const TP lk = lam_g[k], ln = lam_g[nk];
                const TP szk = k == 0 ? (TP)0.5 : (TP)1, szn = nk == 0 ? (TP)0.5 : (TP)1;
                const TP da_k = lxy_a + lk, db_k = lxy_b + lk, da_n = lxy_a + ln, db_n = lxy_b + ln;
                xa_k = da_k > (TP)0 ? xa_k * sa * szk / da_k : (TP)0;
                xb_k = db_k > (TP)0 ? xb_k * sb * szk / db_k : (TP)0;
                xa_n = da_n > (TP)0 ? xa_n * sa * szn / da_n : (TP)0;
                xb_n = db_n > (TP)0 ? xb_n * sb * szn / db_n : (TP)0;
            } else {
                const Cplx<TP> rk = buf[k * LC + c], rn = buf[nk * LC + c];
                xa_k = rk.x; xb_k = rk.y; xa_n = rn.x; xb_n = rn.y;
            }
            if (k == 0) {
                // nk == 0 as well: H[0] = X[0]
                buf[c] = {xa_k, xb_k};
            } else {
                const Cplx<TP> ok = cconj(om_g[k]), on = cconj(om_g[nk]);
                // H_a[k] = 1/2 ok (xa_k - i xa_n) ; H_b[k] likewise ; Z'[k] = H_a[k] + i H_b[k]
                const Cplx<TP> ha_k = cmul(ok, Cplx<TP>{(TP)0.5 * xa_k, (TP)-0.5 * xa_n}), hb_k = cmul(ok, Cplx<TP>{(TP)0.5 * xb_k, (TP)-0.5 * xb_n});
                const Cplx<TP> zk2 = {ha_k.x - hb_k.y, ha_k.y + hb_k.x};
                if (nk != k) {
                    const Cplx<TP> ha_n = cmul(on, Cplx<TP>{(TP)0.5 * xa_n, (TP)-0.5 * xa_k}), hb_n = cmul(on, Cplx<TP>{(TP)0.5 * xb_n, (TP)-0.5 * xb_k});
                    buf[nk * LC + c] = {ha_n.x - hb_n.y, ha_n.y + hb_n.x};
                }
                buf[k * LC + c] = zk2;
            }
        }
        __syncthreads();
        // ---------------- inverse FFT (DIF, natural input, bit-reversed output), e^{+2 pi i jk/n} ----------------
        for (int s = lg; s >= 1; s--) {
            const int half = 1 << (s - 1);
            const int tstep = n >> s;
            for (int b = tid; b < nbf; b += kBlock) {
                const int c = b & (LC - 1), bf = b >> 3;
                const int pos = bf & (half - 1);
                const int i0 = ((bf >> (s - 1)) << s) + pos, i1 = i0 + half;
                const Cplx<TP> w = cconj(tw[pos * tstep]);
                const Cplx<TP> u = buf[i0 * LC + c], v = buf[i1 * LC + c];
                buf[i0 * LC + c] = cadd(u, v);
                buf[i1 * LC + c] = cmul(w, csub(u, v));
            }
            __syncthreads();
        }
        // ---------------- store: x[2j] = v[j], x[2j+1] = v[n-1-j]; v[j] sits at slot bitrev(j) ----------------
        double acc = 0.;
        for (int idx = tid; idx < total; idx += kBlock) {
            int l, j;
            if (P.l_fastest) { l = idx & (L - 1); j = idx >> 4; } else { j = idx & (n - 1); l = idx >> lg; }
            const int pj = (j & 1) ? (n - 1 - (j >> 1)) : (j >> 1);
            const int slot = (int)(__brev((unsigned)pj) >> (32 - lg));
            const TP* srcp = reinterpret_cast<const TP*>(&buf[slot * LC + (l >> 1)]);
            const TP v = srcp[l & 1];
            const long long o = base + (long long)l * P.line_stride + (long long)j * P.elem_stride;
            out[P.out_off + o] = (TOut)v;
            if (DOT) acc += (double)v * (double)dot_with[P.out_off + o];
        }
        if (DOT) {
            acc = block_sum(acc, red);
            if (tid == 0) partials[blockIdx.x] = acc;
        }
    }
}

}  // namespace shm
