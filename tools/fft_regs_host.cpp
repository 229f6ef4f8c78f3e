// Host check of pyslice_amd/csrc/fft_regs.h: the register FFT network and the four-step
// (32 lanes x 32 registers) index algebra used by fft_pow2.h, against a float64 DFT.
//   g++ -O2 -std=c++17 -o tools/bin/fft_regs_host tools/fft_regs_host.cpp && tools/bin/fft_regs_host
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pyslice_amd/csrc/fft_regs.h"
using namespace msl;
using cd = std::complex<double>;

static std::vector<cd> dft(const std::vector<cd>& x, int sign) {
    int n = (int)x.size();
    std::vector<cd> y(n);
    for (int k = 0; k < n; ++k) {
        cd acc = 0;
        for (int j = 0; j < n; ++j) acc += x[j] * std::polar(1.0, sign * 2.0 * M_PI * (double)((long long)j * k % n) / n);
        y[k] = acc;
    }
    return y;
}

template <int N, bool INV>
static double check_regs() {
    cf v[N];
    std::vector<cd> x(N);
    for (int i = 0; i < N; ++i) { x[i] = cd(drand48() - 0.5, drand48() - 0.5); v[i] = mk((float)x[i].real(), (float)x[i].imag()); x[i] = cd(v[i].x, v[i].y); }
    fft_regs<N, INV>(v);
    auto y = dft(x, INV ? +1 : -1);
    double num = 0, den = 0;
    for (int i = 0; i < N; ++i) { num += std::norm(cd(v[i].x, v[i].y) - y[i]); den += std::norm(y[i]); }
    return std::sqrt(num / den);
}

// decimation-in-time network with fused multiply-adds; WMODE 1 / 2: weights (conjugated) on the inputs, folded into the leaf level
template <int N, bool INV, int WMODE>
static double check_dit() {
    cf v[N], in[N], w[N];
    std::vector<cd> x(N);
    for (int i = 0; i < N; ++i) {
        in[i] = mk((float)(drand48() - 0.5), (float)(drand48() - 0.5));
        const double a = 6.283185307179586 * drand48();
        w[i] = mk((float)std::cos(a), (float)std::sin(a));
        x[i] = cd(in[i].x, in[i].y);
        if (WMODE == 1) x[i] *= cd(w[i].x, w[i].y);
        if (WMODE == 2) x[i] *= cd(w[i].x, -w[i].y);
    }
    fft_regs_dit<N, INV, WMODE>(in, v, w);
    auto y = dft(x, INV ? +1 : -1);
    double num = 0, den = 0;
    for (int i = 0; i < N; ++i) { num += std::norm(cd(v[i].x, v[i].y) - y[i]); den += std::norm(y[i]); }
    return std::sqrt(num / den);
}

// the slice loop's four-step form on the DIT network: register FFT, exchange, register FFT with the twiddles as leaf weights
template <int R, bool INV>
static double check_fourstep_dit() {
    constexpr int N = R * R;
    std::vector<cd> x(N);
    static cf regs[R][R], lds[R][R];
    for (int n = 0; n < N; ++n) {
        float a = (float)(drand48() - 0.5), b = (float)(drand48() - 0.5);
        x[n] = cd(a, b);
        regs[n % R][n / R] = mk(a, b);
    }
    for (int n2 = 0; n2 < R; ++n2) {
        cf v[R], in[R];
        for (int i = 0; i < R; ++i) in[i] = regs[n2][i];
        fft_regs_dit<R, INV, 0>(in, v);
        for (int k1 = 0; k1 < R; ++k1) lds[k1][n2] = v[k1];
    }
    std::vector<cd> out(N);
    for (int k1 = 0; k1 < R; ++k1) {
        cf v[R], in[R], w[R];
        for (int n2 = 0; n2 < R; ++n2) {
            in[n2] = lds[k1][n2];
            const double ang = -2.0 * M_PI * (double)(n2 * k1) / N;          // forward table; the inverse conjugates it (WMODE 2)
            w[n2] = mk((float)std::cos(ang), (float)std::sin(ang));
        }
        fft_regs_dit<R, INV, INV ? 2 : 1>(in, v, w);
        for (int k2 = 0; k2 < R; ++k2) out[k1 + R * k2] = cd(v[k2].x, v[k2].y);
    }
    auto y = dft(x, INV ? +1 : -1);
    double num = 0, den = 0;
    for (int i = 0; i < N; ++i) { num += std::norm(out[i] - y[i]); den += std::norm(y[i]); }
    return std::sqrt(num / den);
}

// four-step N = R1*R2 on L=R2 lanes x R1 registers:  element n = n1*R2 + n2 (lane n2, reg n1)
//   fft over n1 -> k1 ; multiply W_N^{n2 k1} ; transpose ; fft over n2 -> k2 ; output k = k1 + R1*k2 (lane k1, reg k2)
template <int R1, int R2, bool INV>
static double check_fourstep() {
    constexpr int N = R1 * R2;
    std::vector<cd> x(N);
    static cf regs[R2][R1];                     // [lane n2][reg n1]
    for (int n = 0; n < N; ++n) {
        float a = (float)(drand48() - 0.5), b = (float)(drand48() - 0.5);
        x[n] = cd(a, b);
        regs[n % R2][n / R2] = mk(a, b);
    }
    static cf lds[R1][R2];                      // [k1][n2]
    for (int n2 = 0; n2 < R2; ++n2) {
        cf v[R1];
        for (int i = 0; i < R1; ++i) v[i] = regs[n2][i];
        fft_regs<R1, INV>(v);
        for (int k1 = 0; k1 < R1; ++k1) {
            double ang = (INV ? 2.0 : -2.0) * M_PI * (double)(n2 * k1) / N;
            cf w = mk((float)std::cos(ang), (float)std::sin(ang));
            lds[k1][n2] = mk(v[k1].x * w.x - v[k1].y * w.y, v[k1].x * w.y + v[k1].y * w.x);
        }
    }
    std::vector<cd> out(N);
    for (int k1 = 0; k1 < R1; ++k1) {           // lane k1 now holds regs n2
        cf v[R2];
        for (int n2 = 0; n2 < R2; ++n2) v[n2] = lds[k1][n2];
        fft_regs<R2, INV>(v);
        for (int k2 = 0; k2 < R2; ++k2) out[k1 + R1 * k2] = cd(v[k2].x, v[k2].y);
    }
    auto y = dft(x, INV ? +1 : -1);
    double num = 0, den = 0;
    for (int i = 0; i < N; ++i) { num += std::norm(out[i] - y[i]); den += std::norm(y[i]); }
    return std::sqrt(num / den);
}

int main() {
    int bad = 0;
    auto rep = [&](const char* name, double e, double tol) { printf("%-28s rel-l2 %.3e %s\n", name, e, e < tol ? "ok" : "FAIL"); if (!(e < tol)) ++bad; };
    rep("fft_regs<2> fwd", check_regs<2, false>(), 1e-6);
    rep("fft_regs<4> fwd", check_regs<4, false>(), 1e-6);
    rep("fft_regs<8> fwd", check_regs<8, false>(), 1e-6);
    rep("fft_regs<16> fwd", check_regs<16, false>(), 1e-6);
    rep("fft_regs<32> fwd", check_regs<32, false>(), 1e-6);
    rep("fft_regs<64> fwd", check_regs<64, false>(), 1e-6);
    rep("fft_regs<8> inv", check_regs<8, true>(), 1e-6);
    rep("fft_regs<16> inv", check_regs<16, true>(), 1e-6);
    rep("fft_regs<32> inv", check_regs<32, true>(), 1e-6);
    rep("fft_regs<64> inv", check_regs<64, true>(), 1e-6);
    // mixed radix (2, 3, 4, 5): the per-lane time transform of tacaw_time.h
    rep("fft_regs<3> fwd", check_regs<3, false>(), 1e-6);
    rep("fft_regs<5> fwd", check_regs<5, false>(), 1e-6);
    rep("fft_regs<5> inv", check_regs<5, true>(), 1e-6);
    rep("fft_regs<9> inv", check_regs<9, true>(), 1e-6);
    rep("fft_regs<12> fwd", check_regs<12, false>(), 1e-6);
    rep("fft_regs<15> fwd", check_regs<15, false>(), 1e-6);
    rep("fft_regs<25> fwd", check_regs<25, false>(), 1e-6);
    rep("fft_regs<40> fwd", check_regs<40, false>(), 1e-6);
    rep("fft_regs<45> inv", check_regs<45, true>(), 1e-6);
    rep("fft_regs<96> fwd", check_regs<96, false>(), 1e-6);
    rep("fft_regs<100> fwd", check_regs<100, false>(), 1e-6);
    rep("fft_regs<100> inv", check_regs<100, true>(), 1e-6);
    rep("fft_regs<120> fwd", check_regs<120, false>(), 1e-6);
    rep("fft_regs<125> fwd", check_regs<125, false>(), 1e-6);
    rep("fft_regs<128> fwd", check_regs<128, false>(), 1e-6);
    // radix 7: the mixed-radix slice-loop passes of rowtm_pass.h (7-smooth factors up to 32)
    rep("fft_regs<7> fwd", check_regs<7, false>(), 1e-6);
    rep("fft_regs<7> inv", check_regs<7, true>(), 1e-6);
    rep("fft_regs<14> fwd", check_regs<14, false>(), 1e-6);
    rep("fft_regs<21> inv", check_regs<21, true>(), 1e-6);
    rep("fft_regs<21> fwd", check_regs<21, false>(), 1e-6);
    rep("fft_regs<28> fwd", check_regs<28, false>(), 1e-6);
    rep("fft_regs<28> inv", check_regs<28, true>(), 1e-6);
    rep("fft_regs<49> fwd", check_regs<49, false>(), 1e-6);
    rep("dit<2> fwd", check_dit<2, false, 0>(), 1e-6);
    rep("dit<4> inv", check_dit<4, true, 0>(), 1e-6);
    rep("dit<8> fwd", check_dit<8, false, 0>(), 1e-6);
    rep("dit<8> inv w", check_dit<8, true, 1>(), 1e-6);
    rep("dit<16> fwd", check_dit<16, false, 0>(), 1e-6);
    rep("dit<16> inv", check_dit<16, true, 0>(), 1e-6);
    rep("dit<16> fwd w", check_dit<16, false, 1>(), 1e-6);
    rep("dit<16> inv conj w", check_dit<16, true, 2>(), 1e-6);
    rep("dit<32> fwd", check_dit<32, false, 0>(), 1e-6);
    rep("dit<32> inv", check_dit<32, true, 0>(), 1e-6);
    rep("dit<32> fwd w", check_dit<32, false, 1>(), 1e-6);
    rep("dit<32> inv w", check_dit<32, true, 1>(), 1e-6);
    rep("dit<32> fwd conj w", check_dit<32, false, 2>(), 1e-6);
    rep("dit<32> inv conj w", check_dit<32, true, 2>(), 1e-6);
    rep("dit<64> fwd w", check_dit<64, false, 1>(), 1e-6);
    rep("dit<64> inv", check_dit<64, true, 0>(), 1e-6);
    rep("fourstep dit 32x32 fwd", check_fourstep_dit<32, false>(), 1e-6);
    rep("fourstep dit 32x32 inv", check_fourstep_dit<32, true>(), 1e-6);
    rep("fourstep dit 16x16 fwd", check_fourstep_dit<16, false>(), 1e-6);
    rep("fourstep dit 16x16 inv", check_fourstep_dit<16, true>(), 1e-6);
    rep("fourstep 32x32 fwd", check_fourstep<32, 32, false>(), 1e-6);
    rep("fourstep 32x32 inv", check_fourstep<32, 32, true>(), 1e-6);
    rep("fourstep 16x16 fwd", check_fourstep<16, 16, false>(), 1e-6);
    rep("fourstep 16x32 fwd", check_fourstep<16, 32, false>(), 1e-6);
    rep("fourstep 32x64 inv", check_fourstep<32, 64, true>(), 1e-6);
    return bad;
}
