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

// FENCE: a scheduling barrier after every butterfly.  A long transform held by ONE lane (tacaw_time.h: 100 complex registers) leaves
// the scheduler free to interleave dozens of independent butterflies, whose temporaries then no longer fit beside the data; in
// program order a butterfly needs its own operands and nothing else.  (The 16- / 32-point transforms of the slice loop stay unfenced.)
template <bool FENCE>
MSL_HD void fft_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
#endif
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
// Other lengths (the per-lane time transform of tacaw_time.h: 100 = 4.5.5 frames, 96 = 4.4.2.3 ...) continue with radix 5 and 3;
// a length with another prime factor has no plan (fft_smooth).
constexpr int fft_radix(int n) { return (n % 4 == 0) ? 4 : (n % 2 == 0) ? 2 : (n % 5 == 0) ? 5 : 3; }
constexpr bool fft_smooth(int n) {
    if (n < 1) return false;
    while (n % 2 == 0) n /= 2;
    while (n % 3 == 0) n /= 3;
    while (n % 5 == 0) n /= 5;
    return n == 1;
}

// frequency index held at position i after dif<N>: block q = i / (N/r) holds the sub-transform of the outputs r m + q
constexpr int dif_out_index(int i, int n) {
    if (n == 1) return 0;
    const int r = fft_radix(n), m = n / r;
    return r * dif_out_index(i % m, m) + i / m;
}

template <int N, int S, bool INV, int K, bool FENCE = false>
MSL_HD void dif2_level(cf* v) {
    if constexpr (K < N / 2) {
        cf a = v[K * S], b = v[(K + N / 2) * S];
        v[K * S] = mk(a.x + b.x, a.y + b.y);
        v[(K + N / 2) * S] = twiddle_mul<N, K, INV>(mk(a.x - b.x, a.y - b.y));
        fft_fence<FENCE>();
        dif2_level<N, S, INV, K + 1, FENCE>(v);
    }
}

template <int N, int S, bool INV, int K, bool FENCE = false>
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
        fft_fence<FENCE>();
        dif4_level<N, S, INV, K + 1, FENCE>(v);
    }
}

// radix 3: y0 = a0 + s, y1,2 = a0 - s/2 -+ i sin(2 pi/3) d   (s = a1 + a2, d = a1 - a2; signs for the forward transform)
template <int N, int S, bool INV, int K, bool FENCE = false>
MSL_HD void dif3_level(cf* v) {
    if constexpr (K < N / 3) {
        constexpr int Q = N / 3;
        constexpr float sn = 0.86602540378443864676f;
        const cf a0 = v[K * S], a1 = v[(K + Q) * S], a2 = v[(K + 2 * Q) * S];
        const cf s = mk(a1.x + a2.x, a1.y + a2.y), d = mk(a1.x - a2.x, a1.y - a2.y);
        const cf m = mk(a0.x - 0.5f * s.x, a0.y - 0.5f * s.y);
        const cf n = INV ? mk(-sn * d.y, sn * d.x) : mk(sn * d.y, -sn * d.x);      // -+ i sn d
        v[K * S] = mk(a0.x + s.x, a0.y + s.y);
        v[(K + Q) * S] = twiddle_mul<N, K, INV>(mk(m.x + n.x, m.y + n.y));
        v[(K + 2 * Q) * S] = twiddle_mul<N, 2 * K, INV>(mk(m.x - n.x, m.y - n.y));
        fft_fence<FENCE>();
        dif3_level<N, S, INV, K + 1, FENCE>(v);
    }
}

// radix 5 on the sums and differences of the pairs (1,4), (2,3): 36 instructions per butterfly
template <int N, int S, bool INV, int K, bool FENCE = false>
MSL_HD void dif5_level(cf* v) {
    if constexpr (K < N / 5) {
        constexpr int Q = N / 5;
        constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;     // cos(2 pi/5), cos(4 pi/5)
        constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;      // sin(2 pi/5), sin(4 pi/5)
        const cf a0 = v[K * S], a1 = v[(K + Q) * S], a2 = v[(K + 2 * Q) * S], a3 = v[(K + 3 * Q) * S], a4 = v[(K + 4 * Q) * S];
        const cf p1 = mk(a1.x + a4.x, a1.y + a4.y), d1 = mk(a1.x - a4.x, a1.y - a4.y);
        const cf p2 = mk(a2.x + a3.x, a2.y + a3.y), d2 = mk(a2.x - a3.x, a2.y - a3.y);
        const cf m1 = mk(a0.x + c1 * p1.x + c2 * p2.x, a0.y + c1 * p1.y + c2 * p2.y);
        const cf m2 = mk(a0.x + c2 * p1.x + c1 * p2.x, a0.y + c2 * p1.y + c1 * p2.y);
        const cf n1 = mk(s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y);
        const cf n2 = mk(s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y);
        // forward: y1 = m1 - i n1, y4 = m1 + i n1, y2 = m2 - i n2, y3 = m2 + i n2; the inverse swaps the signs
        const cf r1 = INV ? mk(-n1.y, n1.x) : mk(n1.y, -n1.x);
        const cf r2 = INV ? mk(-n2.y, n2.x) : mk(n2.y, -n2.x);
        v[K * S] = mk(a0.x + p1.x + p2.x, a0.y + p1.y + p2.y);
        v[(K + Q) * S] = twiddle_mul<N, K, INV>(mk(m1.x + r1.x, m1.y + r1.y));
        v[(K + 2 * Q) * S] = twiddle_mul<N, 2 * K, INV>(mk(m2.x + r2.x, m2.y + r2.y));
        v[(K + 3 * Q) * S] = twiddle_mul<N, 3 * K, INV>(mk(m2.x - r2.x, m2.y - r2.y));
        v[(K + 4 * Q) * S] = twiddle_mul<N, 4 * K, INV>(mk(m1.x - r1.x, m1.y - r1.y));
        fft_fence<FENCE>();
        dif5_level<N, S, INV, K + 1, FENCE>(v);
    }
}

template <int N, int S, bool INV, int Q, bool FENCE = false>
MSL_HD void dif_blocks(cf* v);

// decimation in frequency on v[0], v[S], ..., v[(N-1)S]; position i ends up holding frequency dif_out_index(i, N)
template <int N, int S, bool INV, bool FENCE = false>
MSL_HD void dif(cf* v) {
    if constexpr (N > 1) {
        if constexpr (fft_radix(N) == 4) dif4_level<N, S, INV, 0, FENCE>(v);
        else if constexpr (fft_radix(N) == 2) dif2_level<N, S, INV, 0, FENCE>(v);
        else if constexpr (fft_radix(N) == 5) dif5_level<N, S, INV, 0, FENCE>(v);
        else dif3_level<N, S, INV, 0, FENCE>(v);
        dif_blocks<N, S, INV, 0, FENCE>(v);
    }
}

template <int N, int S, bool INV, int Q, bool FENCE>
MSL_HD void dif_blocks(cf* v) {
    constexpr int R = fft_radix(N);
    if constexpr (Q < R) {
        dif<N / R, S, INV, FENCE>(v + Q * (N / R) * S);
        dif_blocks<N, S, INV, Q + 1, FENCE>(v);
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
