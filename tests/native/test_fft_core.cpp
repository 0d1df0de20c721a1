// Host unit test of signed-heat-3d_amd/csrc/shm_fft_core.h (the exact code the device kernels run): Stockham passes
// against a naive DFT, and the full packed DCT-II / DCT-III pipeline against the O(n^2) cosine sums.
// Build+run:  g++ -O2 -std=c++17 tests/native/test_fft_core.cpp -o /tmp/test_fft_core && /tmp/test_fft_core
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../signed-heat-3d_amd/csrc/shm_fft_core.h"

using namespace shm;
typedef Cplx<double> C;
#ifndef TEST_LC
#define TEST_LC 8
#endif
#ifndef TEST_POW
#define TEST_POW false   // true: inter-pass twiddles from the power tree (what the device uses for n >= 512)
#endif
constexpr int kFftLC = TEST_LC, kFftRow = TEST_LC + 1;

template <int LOG2N, int R, int NS, int SIGN> static void run_pass(std::vector<C>& buf, const std::vector<C>& tw) {
    constexpr int items = PassGeom<LOG2N, R, kFftLC>::items;
    std::vector<C> regs((size_t)items * R);
    for (int w = 0; w < items; w++) {  // "before the barrier"
        C v[R];
        pass_load<double, LOG2N, R, NS, SIGN, kFftLC, TEST_POW>(buf.data(), tw.data(), w, v);
        for (int r = 0; r < R; r++) regs[(size_t)w * R + r] = v[r];
    }
    for (int w = 0; w < items; w++) {  // "after the barrier"
        C v[R];
        for (int r = 0; r < R; r++) v[r] = regs[(size_t)w * R + r];
        pass_store<double, LOG2N, R, NS, kFftLC>(buf.data(), w, v);
    }
}

template <int LOG2N, int SIGN> static void fft_tile(std::vector<C>& buf, const std::vector<C>& tw) {
    typedef FftPlan<LOG2N> P;
    run_pass<LOG2N, P::R0, 1, SIGN>(buf, tw);
    if (P::npass > 1) run_pass<LOG2N, P::R1, P::R0, SIGN>(buf, tw);
    if (P::npass > 2) run_pass<LOG2N, P::R2, P::R0 * P::R1, SIGN>(buf, tw);
    if (P::npass > 3) run_pass<LOG2N, P::R3, P::R0 * P::R1 * P::R2, SIGN>(buf, tw);
}

template <int LOG2N> static double test_n() {
    constexpr int n = 1 << LOG2N;
    const double pi = 3.14159265358979323846;
    std::vector<C> tw(n), om(n);
    for (int t = 0; t < n; t++) {
        tw[t] = {std::cos(-2 * pi * t / n), std::sin(-2 * pi * t / n)};
        om[t] = {std::cos(-pi * t / (2. * n)), std::sin(-pi * t / (2. * n))};
    }
    double worst = 0;
    // ---- plain FFT, both signs
    for (int sign = -1; sign <= 1; sign += 2) {
        std::vector<C> buf((size_t)n * kFftRow), ref((size_t)n * kFftLC);
        std::vector<C> in((size_t)n * kFftLC);
        for (auto& z : in) z = {drand48() - 0.5, drand48() - 0.5};
        for (int j = 0; j < n; j++)
            for (int c = 0; c < kFftLC; c++) buf[(size_t)j * kFftRow + c] = in[(size_t)j * kFftLC + c];
        if (sign < 0) fft_tile<LOG2N, -1>(buf, tw);
        else fft_tile<LOG2N, +1>(buf, tw);
        for (int c = 0; c < kFftLC; c += 3)
            for (int k = 0; k < n; k += (n > 64 ? 7 : 1)) {
                double sr = 0, si = 0;
                for (int j = 0; j < n; j++) {
                    const double a = sign * 2 * pi * (double)((long long)j * k % n) / n;
                    const C z = in[(size_t)j * kFftLC + c];
                    sr += z.x * std::cos(a) - z.y * std::sin(a);
                    si += z.x * std::sin(a) + z.y * std::cos(a);
                }
                const C g = buf[(size_t)k * kFftRow + c];
                worst = std::fmax(worst, std::fmax(std::fabs(g.x - sr), std::fabs(g.y - si)) / std::sqrt((double)n));
            }
    }
    // ---- packed DCT-II then DCT-III: 16 real lines
    constexpr int NL = 2 * kFftLC;
    std::vector<double> x((size_t)NL * n);
    for (auto& v : x) v = drand48() - 0.5;
    std::vector<C> buf((size_t)n * kFftRow);
    for (int l = 0; l < NL; l++)
        for (int j = 0; j < n; j++) {
            double* d = reinterpret_cast<double*>(&buf[(size_t)makhoul_slot(j, n) * kFftRow + (l >> 1)]);
            d[l & 1] = x[(size_t)l * n + j];
        }
    fft_tile<LOG2N, -1>(buf, tw);
    std::vector<double> X((size_t)NL * n);
    for (int c = 0; c < kFftLC; c++)
        for (int k = 0; k < n; k++) {
            double xa, xb;
            dct_fwd_post<double>(buf[(size_t)k * kFftRow + c], buf[(size_t)((n - k) & (n - 1)) * kFftRow + c], om[k], xa, xb);
            X[(size_t)(2 * c) * n + k] = xa;
            X[(size_t)(2 * c + 1) * n + k] = xb;
        }
    for (int l = 0; l < NL; l += 5)
        for (int k = 0; k < n; k += (n > 64 ? 5 : 1)) {
            double s = 0;
            for (int j = 0; j < n; j++) s += x[(size_t)l * n + j] * std::cos(pi * (2 * j + 1) * k / (2. * n));
            worst = std::fmax(worst, std::fabs(s - X[(size_t)l * n + k]) / std::sqrt((double)n));
        }
    // inverse (DCT-III, unnormalised): y_j = sum_k X_k cos(pi (2j+1) k / 2n)
    for (int c = 0; c < kFftLC; c++)
        for (int k = 0; k < n; k++) {
            const int nk = (n - k) & (n - 1);
            buf[(size_t)k * kFftRow + c] = dct_inv_pre<double>(k, X[(size_t)(2 * c) * n + k], X[(size_t)(2 * c) * n + nk], X[(size_t)(2 * c + 1) * n + k],
                                                              X[(size_t)(2 * c + 1) * n + nk], om[k]);
        }
    fft_tile<LOG2N, +1>(buf, tw);
    for (int l = 0; l < NL; l += 3)
        for (int j = 0; j < n; j += (n > 64 ? 3 : 1)) {
            double s = 0;
            for (int k = 0; k < n; k++) s += X[(size_t)l * n + k] * std::cos(pi * (2 * j + 1) * k / (2. * n));
            const double* d = reinterpret_cast<const double*>(&buf[(size_t)makhoul_slot(j, n) * kFftRow + (l >> 1)]);
            worst = std::fmax(worst, std::fabs(s - d[l & 1]) / n);
        }
    printf("n=%4d worst scaled error %.3e\n", n, worst);
    return worst;
}

int main() {
    double w = 0;
    w = std::fmax(w, test_n<4>());
    w = std::fmax(w, test_n<5>());
    w = std::fmax(w, test_n<6>());
    w = std::fmax(w, test_n<7>());
    w = std::fmax(w, test_n<8>());
    w = std::fmax(w, test_n<9>());
    w = std::fmax(w, test_n<10>());
    if (!(w < 1e-12)) {
        printf("FAIL\n");
        return 1;
    }
    printf("OK\n");
    return 0;
}
