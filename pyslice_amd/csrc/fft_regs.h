// Register-resident radix-4 / radix-2 FFT butterflies with compile-time twiddles.
//
// fft_regs<N>(v, inverse): in-place N-point DFT of float2 v[N] held in registers, natural
// order in and out (the decimation-in-frequency network leaves digit-reversed order; the final
// permutation is a compile-time renaming).  Used by the four-step power-of-two kernels in
// fft_pow2.h where a 1024-point line is 32 lanes x 32 registers.
//
// The header compiles for the device (hipcc) and, for the CPU unit test of the index algebra
// (tests/test_fft_regs_host.py builds tools/fft_regs_host.cpp with g++), for the host.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MSL_HD __host__ __device__ __forceinline__
namespace msl { using cf = float2; }
#else
#define MSL_HD inline
namespace msl { struct cf { float x, y; }; }
#endif

namespace msl {

MSL_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }

// constexpr cos/sin of 2*pi*k/n in double (Taylor on a reduced argument); exact to double rounding
constexpr double cx_pi = 3.14159265358979323846264338327950288;
constexpr double cx_sin_taylor(double x) {       // |x| <= pi/4
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i < 12; ++i) { term *= -x2 / ((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
constexpr double cx_cos_taylor(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i < 12; ++i) { term *= -x2 / ((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
// cos(2 pi k / n), sin(2 pi k / n) for 0 <= k < n, via octant reduction (exact symmetries)
constexpr double cx_cos2pi(int k, int n) {
    k %= n;
    if (8 * k <= n) return cx_cos_taylor(2 * cx_pi * k / n);
    if (8 * k <= 3 * n) return -cx_sin_taylor(2 * cx_pi * k / n - cx_pi / 2);
    if (8 * k <= 5 * n) return -cx_cos_taylor(2 * cx_pi * k / n - cx_pi);
    if (8 * k <= 7 * n) return cx_sin_taylor(2 * cx_pi * k / n - 3 * cx_pi / 2);
    return cx_cos_taylor(2 * cx_pi * k / n - 2 * cx_pi);
}
constexpr double cx_sin2pi(int k, int n) {
    k %= n;
    if (8 * k <= n) return cx_sin_taylor(2 * cx_pi * k / n);
    if (8 * k <= 3 * n) return cx_cos_taylor(2 * cx_pi * k / n - cx_pi / 2);
    if (8 * k <= 5 * n) return -cx_sin_taylor(2 * cx_pi * k / n - cx_pi);
    if (8 * k <= 7 * n) return -cx_cos_taylor(2 * cx_pi * k / n - 3 * cx_pi / 2);
    return cx_sin_taylor(2 * cx_pi * k / n - 2 * cx_pi);
}

constexpr int bitrev(int i, int n) {
    int r = 0;
    for (int b = 1; b < n; b <<= 1) { r = (r << 1) | (i & 1); i >>= 1; }
    return r;
}

// t * W_N^k with W = exp(-+ 2 pi i / N); INV selects the conjugate (inverse transform)
template <int N, int K, bool INV>
MSL_HD cf twiddle_mul(cf t) {
    if constexpr (K % N == 0) {
        return t;
    } else if constexpr (2 * K == N) {            // -1
        return mk(-t.x, -t.y);
    } else if constexpr (4 * K == N) {            // -i (forward) / +i (inverse)
        return INV ? mk(-t.y, t.x) : mk(t.y, -t.x);
    } else if constexpr (4 * K == 3 * N) {        // +i (forward) / -i (inverse)
        return INV ? mk(t.y, -t.x) : mk(-t.y, t.x);
    } else if constexpr (8 * K == N) {            // (1 -+ i)/sqrt2
        constexpr float h = 0.70710678118654752440f;
        return INV ? mk((t.x - t.y) * h, (t.x + t.y) * h) : mk((t.x + t.y) * h, (t.y - t.x) * h);
    } else if constexpr (8 * K == 3 * N) {        // (-1 -+ i)/sqrt2
        constexpr float h = 0.70710678118654752440f;
        return INV ? mk((-t.x - t.y) * h, (t.x - t.y) * h) : mk((t.y - t.x) * h, (-t.x - t.y) * h);
    } else {
        constexpr float c = (float)cx_cos2pi(K % N, N);
        constexpr float s = (float)cx_sin2pi(K % N, N);   // W = c - i s (forward), c + i s (inverse)
        if constexpr (INV) return mk(t.x * c - t.y * s, t.y * c + t.x * s);
        else return mk(t.x * c + t.y * s, t.y * c - t.x * s);
    }
}

// Radix plan of the decimation-in-frequency network: radix 4 while the length allows it, then one radix-2 stage
// (32 = 4.4.2, 16 = 4.4).  Half as many twiddle multiplications on every input-to-output path as a pure radix-2
// network: the rounding error of a transform, which accumulates linearly over the slices of a multislice run, halves.
constexpr int fft_radix(int n) { return (n % 4 == 0) ? 4 : 2; }

// frequency index held at position i after dif<N>: block q = i / (N/r) holds the sub-transform of the outputs r m + q
constexpr int dif_out_index(int i, int n) {
    if (n == 1) return 0;
    const int r = fft_radix(n), m = n / r;
    return r * dif_out_index(i % m, m) + i / m;
}

template <int N, int S, bool INV, int K>
MSL_HD void dif2_level(cf* v) {
    if constexpr (K < N / 2) {
        cf a = v[K * S], b = v[(K + N / 2) * S];
        v[K * S] = mk(a.x + b.x, a.y + b.y);
        v[(K + N / 2) * S] = twiddle_mul<N, K, INV>(mk(a.x - b.x, a.y - b.y));
        dif2_level<N, S, INV, K + 1>(v);
    }
}

template <int N, int S, bool INV, int K>
MSL_HD void dif4_level(cf* v) {
    if constexpr (K < N / 4) {
        constexpr int Q = N / 4;
        const cf a0 = v[K * S], a1 = v[(K + Q) * S], a2 = v[(K + 2 * Q) * S], a3 = v[(K + 3 * Q) * S];
        const cf t0 = mk(a0.x + a2.x, a0.y + a2.y), t1 = mk(a0.x - a2.x, a0.y - a2.y);
        const cf t2 = mk(a1.x + a3.x, a1.y + a3.y), d = mk(a1.x - a3.x, a1.y - a3.y);
        const cf t3 = INV ? mk(-d.y, d.x) : mk(d.y, -d.x);         // -+ i (a1 - a3)
        v[K * S] = mk(t0.x + t2.x, t0.y + t2.y);
        v[(K + Q) * S] = twiddle_mul<N, K, INV>(mk(t1.x + t3.x, t1.y + t3.y));
        v[(K + 2 * Q) * S] = twiddle_mul<N, 2 * K, INV>(mk(t0.x - t2.x, t0.y - t2.y));
        v[(K + 3 * Q) * S] = twiddle_mul<N, 3 * K, INV>(mk(t1.x - t3.x, t1.y - t3.y));
        dif4_level<N, S, INV, K + 1>(v);
    }
}

template <int N, int S, bool INV, int Q>
MSL_HD void dif_blocks(cf* v);

// decimation in frequency on v[0], v[S], ..., v[(N-1)S]; position i ends up holding frequency dif_out_index(i, N)
template <int N, int S, bool INV>
MSL_HD void dif(cf* v) {
    if constexpr (N > 1) {
        if constexpr (fft_radix(N) == 4) dif4_level<N, S, INV, 0>(v); else dif2_level<N, S, INV, 0>(v);
        dif_blocks<N, S, INV, 0>(v);
    }
}

template <int N, int S, bool INV, int Q>
MSL_HD void dif_blocks(cf* v) {
    constexpr int R = fft_radix(N);
    if constexpr (Q < R) {
        dif<N / R, S, INV>(v + Q * (N / R) * S);
        dif_blocks<N, S, INV, Q + 1>(v);
    }
}

template <int N, int I>
MSL_HD void unscramble(const cf* src, cf* dst) {
    if constexpr (I < N) {
        constexpr int F = dif_out_index(I, N);       // forced compile-time: a run-time index would send the array to scratch
        dst[F] = src[I];
        unscramble<N, I + 1>(src, dst);
    }
}

// natural-order in, natural-order out, unnormalised
template <int N, bool INV>
MSL_HD void fft_regs(cf (&v)[N]) {
    dif<N, 1, INV>(v);
    cf t[N];
    unscramble<N, 0>(v, t);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = t[i];
}

}  // namespace msl
