// In-LDS Stockham FFT building blocks shared by the device kernels (shm_dct.hip.h) and a host unit test
// (tests/native/test_fft_core.cpp, compiled with g++): everything here is plain C++ marked SHM_HD.
//
// A batch of LC complex lines (template parameter; 8, or 4 for n >= 512) of length n lives in a tile  buf[j * ROW + c]  (j = element, c = line, ROW = LC+1:
// one complex of padding per row keeps the x-pass scatter off a single bank).  One pass of radix R reads R elements
// per work item into registers, multiplies by the inter-pass twiddles, does a register DFT_R and writes back in
// place (all reads of the tile happen before the barrier, all writes after it):
//     for jj in [0, n/R):  k = jj % NS;  v[r] = buf[jj + r n/R] * w^{r k n/(NS R)};  v = DFT_R(v);
//                          buf[(jj/NS) NS R + k + r NS] = v[r]                      (NS = product of earlier radices)
// Natural order in, natural order out, no bit reversal (Stockham autosort).
#pragma once

#ifdef __HIPCC__
#define SHM_HD __host__ __device__ __forceinline__
#else
#define SHM_HD inline
#endif

namespace shm {

template <typename TP> struct Cplx { TP x, y; };
template <typename TP> SHM_HD Cplx<TP> cmul(Cplx<TP> a, Cplx<TP> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename TP> SHM_HD Cplx<TP> cadd(Cplx<TP> a, Cplx<TP> b) { return {a.x + b.x, a.y + b.y}; }
template <typename TP> SHM_HD Cplx<TP> csub(Cplx<TP> a, Cplx<TP> b) { return {a.x - b.x, a.y - b.y}; }
template <typename TP> SHM_HD Cplx<TP> cconj(Cplx<TP> a) { return {a.x, -a.y}; }

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// 16th roots of unity: cos(2 pi k/16), sin(2 pi k/16), k = 0..7
template <typename TP> SHM_HD Cplx<TP> root16(int k, int sign /* -1 forward, +1 inverse */) {
    const TP c[8] = {(TP)1.0, (TP)0.92387953251128673848, (TP)0.70710678118654752440, (TP)0.38268343236508977173,
                     (TP)0.0, (TP)-0.38268343236508977173, (TP)-0.70710678118654752440, (TP)-0.92387953251128673848};
    const TP s[8] = {(TP)0.0, (TP)0.38268343236508977173, (TP)0.70710678118654752440, (TP)0.92387953251128673848,
                     (TP)1.0, (TP)0.92387953251128673848, (TP)0.70710678118654752440, (TP)0.38268343236508977173};
    return {c[k], sign < 0 ? -s[k] : s[k]};
}

// Register DFT of size R (2,4,8,16), natural order in and out, decimation in time by compile-time recursion.
template <typename TP, int R, int SIGN> struct RegDft {
    static SHM_HD void run(Cplx<TP> (&v)[R]) {
        Cplx<TP> e[R / 2], o[R / 2];
#pragma unroll
        for (int a = 0; a < R / 2; a++) {
            e[a] = v[2 * a];
            o[a] = v[2 * a + 1];
        }
        RegDft<TP, R / 2, SIGN>::run(e);
        RegDft<TP, R / 2, SIGN>::run(o);
#pragma unroll
        for (int k = 0; k < R / 2; k++) {
            Cplx<TP> t;
            if (k == 0) t = o[k];
            else if (4 * k == R) t = SIGN < 0 ? Cplx<TP>{o[k].y, -o[k].x} : Cplx<TP>{-o[k].y, o[k].x};  // * (-+ i)
            else t = cmul(root16<TP>(k * (16 / R), SIGN), o[k]);
            v[k] = cadd(e[k], t);
            v[k + R / 2] = csub(e[k], t);
        }
    }
};
template <typename TP, int SIGN> struct RegDft<TP, 1, SIGN> {
    static SHM_HD void run(Cplx<TP> (&)[1]) {}
};

// Radix plan per transform length: R[p] for p = 0..npass-1.
template <int LOG2N> struct FftPlan;
template <> struct FftPlan<4> { static constexpr int npass = 1; static constexpr int R0 = 16, R1 = 1, R2 = 1, R3 = 1; };
template <> struct FftPlan<5> { static constexpr int npass = 2; static constexpr int R0 = 8, R1 = 4, R2 = 1, R3 = 1; };
template <> struct FftPlan<6> { static constexpr int npass = 2; static constexpr int R0 = 8, R1 = 8, R2 = 1, R3 = 1; };
template <> struct FftPlan<7> { static constexpr int npass = 2; static constexpr int R0 = 16, R1 = 8, R2 = 1, R3 = 1; };
#ifndef SHM_FFT_PLAN8_BALANCED
#define SHM_FFT_PLAN8_BALANCED 1   // n = 256 as 8*8*4 (all four waves carry butterflies) instead of 16*16 (two of four)
#endif
#if SHM_FFT_PLAN8_BALANCED == 2
template <> struct FftPlan<8> { static constexpr int npass = 4; static constexpr int R0 = 4, R1 = 4, R2 = 4, R3 = 4; };
#elif SHM_FFT_PLAN8_BALANCED == 3
template <> struct FftPlan<8> { static constexpr int npass = 3; static constexpr int R0 = 4, R1 = 8, R2 = 8, R3 = 1; };
#elif SHM_FFT_PLAN8_BALANCED
template <> struct FftPlan<8> { static constexpr int npass = 3; static constexpr int R0 = 8, R1 = 8, R2 = 4, R3 = 1; };
#else
template <> struct FftPlan<8> { static constexpr int npass = 2; static constexpr int R0 = 16, R1 = 16, R2 = 1, R3 = 1; };
#endif
template <> struct FftPlan<9> { static constexpr int npass = 3; static constexpr int R0 = 8, R1 = 8, R2 = 8, R3 = 1; };
template <> struct FftPlan<10> { static constexpr int npass = 3; static constexpr int R0 = 16, R1 = 8, R2 = 8, R3 = 1; };

// Work items of one pass: (n/R) butterflies x LC lines; item w -> line c = w % LC, butterfly jj = w / LC.
template <int LOG2N, int R, int LC> struct PassGeom {
    static constexpr int n = 1 << LOG2N;
    static constexpr int items = (n / R) * LC;
};

// Phase 1 of a pass (before the barrier): gather + twiddle + register DFT for work item w.
// POW = false: the R-1 inter-pass twiddles w^{r k n/(NS R)} are read from the table.  POW = true: only w^{k n/(NS R)} is read and
// the other powers come from a multiplication tree of depth <= 4 (w2 = w1^2, w3 = w2 w1, w4 = w2^2, ...; a few ulp): one table access
// per work item instead of R-1, for the lengths whose table lives in global memory.
template <typename TP, int LOG2N, int R, int NS, int SIGN, int LC, bool POW = false>
SHM_HD void pass_load(const Cplx<TP>* buf, const Cplx<TP>* tw /* [n]: e^{-2 pi i t/n} */, int w, Cplx<TP> (&v)[R]) {
    constexpr int n = 1 << LOG2N;
    const int c = w & (LC - 1), jj = w >> ilog2(LC);
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = buf[(jj + r * (n / R)) * (LC + 1) + c];
    if (NS > 1) {
        const int k = jj & (NS - 1);
        if (POW) {
            Cplx<TP> p[R];
            p[1] = tw[k * (n / (NS * R))];
            if (SIGN > 0) p[1] = cconj(p[1]);
#pragma unroll
            for (int r = 2; r < R; r++) p[r] = cmul(p[r / 2], p[r - r / 2]);
#pragma unroll
            for (int r = 1; r < R; r++) v[r] = cmul(v[r], p[r]);
        } else {
#pragma unroll
            for (int r = 1; r < R; r++) {
                Cplx<TP> t = tw[r * k * (n / (NS * R))];
                if (SIGN > 0) t = cconj(t);
                v[r] = cmul(v[r], t);
            }
        }
    }
    RegDft<TP, R, SIGN>::run(v);
}

// Phase 2 of a pass (after the barrier): scatter.
template <typename TP, int LOG2N, int R, int NS, int LC>
SHM_HD void pass_store(Cplx<TP>* buf, int w, const Cplx<TP> (&v)[R]) {
    const int c = w & (LC - 1), jj = w >> ilog2(LC);
    const int k = jj & (NS - 1);
    const int j0 = (jj / NS) * NS * R + k;
#pragma unroll
    for (int r = 0; r < R; r++) buf[(j0 + r * NS) * (LC + 1) + c] = v[r];
}

// Makhoul permutation: DCT-II of x = Re(om_k FFT(v)_k) with v[pj(j)] = x[j].
SHM_HD int makhoul_slot(int j, int n) { return (j & 1) ? (n - 1 - (j >> 1)) : (j >> 1); }


// DCT-II post step for a pair of real lines packed as z = v_a + i v_b:  from Z[k], Z[(n-k)%n] to X_a[k], X_b[k]
//   V_a = (Z_k + conj Z_{n-k})/2, V_b = -i (Z_k - conj Z_{n-k})/2, X[k] = Re(om_k V[k]), om_k = e^{-i pi k/(2n)}.
template <typename TP> SHM_HD void dct_fwd_post(Cplx<TP> zk, Cplx<TP> znk, Cplx<TP> omk, TP& xa, TP& xb) {
    const Cplx<TP> cz = cconj(znk);
    const Cplx<TP> va = {(TP)0.5 * (zk.x + cz.x), (TP)0.5 * (zk.y + cz.y)};
    const Cplx<TP> vb = {(TP)0.5 * (zk.y - cz.y), (TP)-0.5 * (zk.x - cz.x)};
    xa = omk.x * va.x - omk.y * va.y;
    xb = omk.x * vb.x - omk.y * vb.y;
}

// DCT-III pre step: spectrum of the packed pair, Z'[k] = H_a[k] + i H_b[k], H[k] = 1/2 conj(om_k) (X[k] - i X[n-k]),
// H[0] = X[0]  (then v = IFFT_unnormalised(Z'), x[j] = v[makhoul_slot(j)]).
template <typename TP> SHM_HD Cplx<TP> dct_inv_pre(int k, TP xa_k, TP xa_nk, TP xb_k, TP xb_nk, Cplx<TP> omk) {
    if (k == 0) return {xa_k, xb_k};
    const Cplx<TP> ok = cconj(omk);
    const Cplx<TP> ha = cmul(ok, Cplx<TP>{(TP)0.5 * xa_k, (TP)-0.5 * xa_nk});
    const Cplx<TP> hb = cmul(ok, Cplx<TP>{(TP)0.5 * xb_k, (TP)-0.5 * xb_nk});
    return {ha.x - hb.y, ha.y + hb.x};
}

}  // namespace shm
