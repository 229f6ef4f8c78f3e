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
// Other lengths (the per-lane time transform of tacaw_time.h: 100 = 4.5.5 frames, 96 = 4.4.2.3 ...) continue with radix 5, 3 and 7
// (7: the mixed-radix slice-loop passes of rowtm_pass.h); a length with another prime factor has no plan.
// fft_smooth: factors 2, 3, 5 (what the time kernels are instantiated for); fft_smooth7: 7 as well.
constexpr int fft_radix(int n) { return (n % 4 == 0) ? 4 : (n % 2 == 0) ? 2 : (n % 5 == 0) ? 5 : (n % 3 == 0) ? 3 : 7; }
constexpr bool fft_smooth(int n) {
    if (n < 1) return false;
    while (n % 2 == 0) n /= 2;
    while (n % 3 == 0) n /= 3;
    while (n % 5 == 0) n /= 5;
    return n == 1;
}
constexpr bool fft_smooth7(int n) {
    if (n < 1) return false;
    while (n % 7 == 0) n /= 7;
    return fft_smooth(n);
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

// radix 7 on the sums and differences of the pairs (1,6), (2,5), (3,4): y_k = a0 + sum_j cos(2 pi j k / 7) p_j -+ i sum_j sin(2 pi j k / 7) d_j
template <int N, int S, bool INV, int K, bool FENCE = false>
MSL_HD void dif7_level(cf* v) {
    if constexpr (K < N / 7) {
        constexpr int Q = N / 7;
        constexpr float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;   // cos(2 pi k / 7)
        constexpr float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;    // sin(2 pi k / 7)
        const cf a0 = v[K * S], a1 = v[(K + Q) * S], a2 = v[(K + 2 * Q) * S], a3 = v[(K + 3 * Q) * S], a4 = v[(K + 4 * Q) * S],
                 a5 = v[(K + 5 * Q) * S], a6 = v[(K + 6 * Q) * S];
        const cf p1 = mk(a1.x + a6.x, a1.y + a6.y), d1 = mk(a1.x - a6.x, a1.y - a6.y);
        const cf p2 = mk(a2.x + a5.x, a2.y + a5.y), d2 = mk(a2.x - a5.x, a2.y - a5.y);
        const cf p3 = mk(a3.x + a4.x, a3.y + a4.y), d3 = mk(a3.x - a4.x, a3.y - a4.y);
        const cf m1 = mk(a0.x + c1 * p1.x + c2 * p2.x + c3 * p3.x, a0.y + c1 * p1.y + c2 * p2.y + c3 * p3.y);
        const cf m2 = mk(a0.x + c2 * p1.x + c3 * p2.x + c1 * p3.x, a0.y + c2 * p1.y + c3 * p2.y + c1 * p3.y);
        const cf m3 = mk(a0.x + c3 * p1.x + c1 * p2.x + c2 * p3.x, a0.y + c3 * p1.y + c1 * p2.y + c2 * p3.y);
        const cf n1 = mk(s1 * d1.x + s2 * d2.x + s3 * d3.x, s1 * d1.y + s2 * d2.y + s3 * d3.y);
        const cf n2 = mk(s2 * d1.x - s3 * d2.x - s1 * d3.x, s2 * d1.y - s3 * d2.y - s1 * d3.y);
        const cf n3 = mk(s3 * d1.x - s1 * d2.x + s2 * d3.x, s3 * d1.y - s1 * d2.y + s2 * d3.y);
        // forward: y_k = m_k - i n_k, y_{7-k} = m_k + i n_k; the inverse swaps the signs
        const cf r1 = INV ? mk(-n1.y, n1.x) : mk(n1.y, -n1.x);
        const cf r2 = INV ? mk(-n2.y, n2.x) : mk(n2.y, -n2.x);
        const cf r3 = INV ? mk(-n3.y, n3.x) : mk(n3.y, -n3.x);
        v[K * S] = mk(a0.x + p1.x + p2.x + p3.x, a0.y + p1.y + p2.y + p3.y);
        v[(K + Q) * S] = twiddle_mul<N, K, INV>(mk(m1.x + r1.x, m1.y + r1.y));
        v[(K + 2 * Q) * S] = twiddle_mul<N, 2 * K, INV>(mk(m2.x + r2.x, m2.y + r2.y));
        v[(K + 3 * Q) * S] = twiddle_mul<N, 3 * K, INV>(mk(m3.x + r3.x, m3.y + r3.y));
        v[(K + 4 * Q) * S] = twiddle_mul<N, 4 * K, INV>(mk(m3.x - r3.x, m3.y - r3.y));
        v[(K + 5 * Q) * S] = twiddle_mul<N, 5 * K, INV>(mk(m2.x - r2.x, m2.y - r2.y));
        v[(K + 6 * Q) * S] = twiddle_mul<N, 6 * K, INV>(mk(m1.x - r1.x, m1.y - r1.y));
        fft_fence<FENCE>();
        dif7_level<N, S, INV, K + 1, FENCE>(v);
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
        else if constexpr (fft_radix(N) == 3) dif3_level<N, S, INV, 0, FENCE>(v);
        else dif7_level<N, S, INV, 0, FENCE>(v);
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

// ---- decimation in time with fused multiply-adds (power-of-two lengths; the slice-loop kernels of fft_pow2.h) -----------------
// The decimation-in-frequency network above multiplies AFTER it adds, so a twiddle costs a full complex product (2 mul + 2 fma)
// on top of the butterfly's four additions.  Decimation in time multiplies BEFORE it adds, and then the product folds into the
// additions:  a + W b  is two chained FMAs per component, and  a - W b = 2 a - (a + W b)  one more -- 6 instructions per radix-2
// butterfly with a general twiddle instead of 8 (32 points: 388 instead of ~430, the operation count of a split-radix transform).
// Same for a pointwise product IN FRONT of a transform (inter-FFT twiddles, Fresnel factor, transmission function): the leaf level
// has trivial twiddles, so  (wa a) +- (wb b)  is 10 instructions per butterfly instead of 4 + 4 + 4 -- the products of the slice
// loop are passed to the leaf level as weights (dit_leaf_w) instead of being applied by a separate pass over the registers.
// Radix plan: 4 while the length allows it and a radix-2 leaf otherwise (32 = 4.4.2 from the top, 16 = 4.4).  A top-level radix-4
// butterfly with general twiddles applies W^k, W^2k, W^3k to its inputs directly (u0 = W^k a1 + W^3k a3 as product + two chained
// FMAs) rather than as two radix-2 levels in sequence: the same 24 instructions, but every path from input to output crosses one
// rounded twiddle constant per radix-4 level -- the constants' errors are systematic and add up linearly over the slices of a run.
constexpr int dit_radix(int n) { return (n % 4 == 0) ? 4 : 2; }
// position (in units of the stride) that holds frequency k after dit<N>: the sub-transform of the inputs r i + q leaves Y_q[k] at
// r pos(k) + q, and the combining butterfly over q writes X[k + (N/r) p] to r pos(k) + p
constexpr int dit_pos(int k, int n) {
    if (n == 1) return 0;
    const int r = dit_radix(n), m = n / r;
    return r * dit_pos(k % m, m) + k / m;
}
constexpr int dit_freq(int i, int n) {             // inverse map: frequency held at position i
    if (n == 1) return 0;
    const int r = dit_radix(n), m = n / r;
    return dit_freq(i / r, m) + m * (i % r);
}

MSL_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// (a, b) <- (a + W b, a - W b),  W = W_N^K = exp(-2 pi i K / N), conjugated when INV
template <int N, int K, bool INV>
MSL_HD void dit_bfly(cf& a, cf& b) {
    constexpr int KK = ((K % N) + N) % N;
    if constexpr (KK == 0) {
        const cf t = mk(a.x + b.x, a.y + b.y); b = mk(a.x - b.x, a.y - b.y); a = t;
    } else if constexpr (2 * KK == N) {            // W = -1
        const cf t = mk(a.x - b.x, a.y - b.y); b = mk(a.x + b.x, a.y + b.y); a = t;
    } else if constexpr ((4 * KK == N) != INV && (4 * KK == N || 4 * KK == 3 * N)) {      // W b = -i b = (b.y, -b.x)
        const cf t = mk(a.x + b.y, a.y - b.x); b = mk(a.x - b.y, a.y + b.x); a = t;
    } else if constexpr (4 * KK == N || 4 * KK == 3 * N) {                                // W b = +i b = (-b.y, b.x)
        const cf t = mk(a.x - b.y, a.y + b.x); b = mk(a.x + b.y, a.y - b.x); a = t;
    } else {
        constexpr float c = (float)cx_cos2pi(KK, N);
        constexpr float s = INV ? -(float)cx_sin2pi(KK, N) : (float)cx_sin2pi(KK, N);     // W = c - i s:  W b = (c b.x + s b.y, c b.y - s b.x)
        const cf t = mk(fma_(s, b.y, fma_(c, b.x, a.x)), fma_(-s, b.x, fma_(c, b.y, a.y)));
        b = mk(fma_(2.0f, a.x, -t.x), fma_(2.0f, a.y, -t.y));
        a = t;
    }
}

// radix-4 combination: inputs Y_q[k] in a_q, outputs X[k + (N/4) p] in a_p;  W_N^{qK} on input q
template <int N, int K, bool INV>
MSL_HD void dit_bfly4(cf& a0, cf& a1, cf& a2, cf& a3) {
    if constexpr (K % N == 0 || (8 * K) % N == 0) {      // trivial and W_8-family twiddles: two radix-2 levels are cheaper (16 / 20)
        dit_bfly<N, 2 * K, INV>(a0, a2);                 // t0, t1
        dit_bfly<N, 2 * K, INV>(a1, a3);                 // d = a1 + W^2K a3, e = a1 - W^2K a3
        dit_bfly<N, K, INV>(a0, a1);                     // y0, y2 = t0 +- W^K d
        dit_bfly<N, K + N / 4, INV>(a2, a3);             // y1, y3 = t1 -+ i W^K e
        const cf y2 = a1; a1 = a2; a2 = y2;
    } else {
        dit_bfly<N, 2 * K, INV>(a0, a2);                 // a0 = t0 = a0 + W^2K a2, a2 = t1
        constexpr float c1 = (float)cx_cos2pi(K % N, N), c3 = (float)cx_cos2pi((3 * K) % N, N);
        constexpr float s1 = INV ? -(float)cx_sin2pi(K % N, N) : (float)cx_sin2pi(K % N, N);
        constexpr float s3 = INV ? -(float)cx_sin2pi((3 * K) % N, N) : (float)cx_sin2pi((3 * K) % N, N);
        const cf p = mk(fma_(s1, a1.y, c1 * a1.x), fma_(-s1, a1.x, c1 * a1.y));                          // W^K a1
        const cf u0 = mk(fma_(s3, a3.y, fma_(c3, a3.x, p.x)), fma_(-s3, a3.x, fma_(c3, a3.y, p.y)));     // + W^3K a3
        const cf u1 = mk(fma_(2.0f, p.x, -u0.x), fma_(2.0f, p.y, -u0.y));                                // W^K a1 - W^3K a3
        const cf t0 = a0, t1 = a2;
        a0 = mk(t0.x + u0.x, t0.y + u0.y);
        a2 = mk(t0.x - u0.x, t0.y - u0.y);
        if constexpr (!INV) { a1 = mk(t1.x + u1.y, t1.y - u1.x); a3 = mk(t1.x - u1.y, t1.y + u1.x); }   // t1 -+ i u1
        else                { a1 = mk(t1.x - u1.y, t1.y + u1.x); a3 = mk(t1.x + u1.y, t1.y - u1.x); }
    }
}

// FENCE: a scheduling barrier after every FENCE-th butterfly (0: none), see fft_fence
template <int N, int S, bool INV, bool SKIP_LEAF, int K, int FENCE = 0>
MSL_HD void dit_combine(cf* v) {
    constexpr int R = dit_radix(N), M = N / R;
    if constexpr (K < M) {
        constexpr int B = R * dit_pos(K, M);
        if constexpr (R == 4) dit_bfly4<N, K, INV>(v[B * S], v[(B + 1) * S], v[(B + 2) * S], v[(B + 3) * S]);
        else dit_bfly<N, K, INV>(v[B * S], v[(B + 1) * S]);
        if constexpr (FENCE > 0) fft_fence<(K % FENCE) == FENCE - 1>();
        dit_combine<N, S, INV, SKIP_LEAF, K + 1, FENCE>(v);
    }
}

template <int N, int S, bool INV, bool SKIP_LEAF, int FENCE = 0>
MSL_HD void dit(cf* v);

template <int N, int S, bool INV, bool SKIP_LEAF, int Q, int FENCE = 0>
MSL_HD void dit_subs(cf* v) {
    constexpr int R = dit_radix(N);
    if constexpr (Q < R) {
        dit<N / R, S * R, INV, SKIP_LEAF, FENCE>(v + Q * S);
        if constexpr (FENCE > 0) fft_fence<true>();
        dit_subs<N, S, INV, SKIP_LEAF, Q + 1, FENCE>(v);
    }
}

// decimation in time on v[0], v[S], ..., v[(N-1)S], natural order in; position i ends up holding frequency dit_freq(i, N).
// SKIP_LEAF: the leaf level (the butterflies of the innermost radix, trivial twiddles) has been done by the caller (dit_leaf_*).
template <int N, int S, bool INV, bool SKIP_LEAF, int FENCE>
MSL_HD void dit(cf* v) {
    if constexpr (N > 1) {
        dit_subs<N, S, INV, SKIP_LEAF, 0, FENCE>(v);
        if constexpr (!(SKIP_LEAF && N == dit_radix(N))) dit_combine<N, S, INV, SKIP_LEAF, 0, FENCE>(v);
    }
}

constexpr int dit_leaf_radix(int n) { while (n > 4) n /= 4; return n; }            // 32 -> 2, 16 -> 4, 8 -> 2, 64 -> 4
constexpr int dit_leaf_count(int n) { return n / dit_leaf_radix(n); }              // leaf butterflies; butterfly I takes the elements I + q count

// leaf butterfly I of an N-point transform, out of place (in == out is fine), its inputs weighted:
//   WMODE 0: no weights;  1: in[I + q C] * w[q];  2: in[I + q C] * conj(w[q])          (C = dit_leaf_count(N), w = the butterfly's r weights)
template <int N, bool INV, int WMODE, int I>
MSL_HD void dit_leaf(const cf* in, cf* out, const cf* w) {
    constexpr int R = dit_leaf_radix(N), C = N / R;
    // product with the weight, and  p + w b  as two chained FMAs per component
    auto wmul = [](cf a, cf ww) {
        if constexpr (WMODE == 2) return mk(fma_(a.x, ww.x, a.y * ww.y), fma_(a.y, ww.x, -(a.x * ww.y)));
        else return mk(fma_(a.x, ww.x, -(a.y * ww.y)), fma_(a.y, ww.x, a.x * ww.y));
    };
    auto wfma = [](cf p, cf b, cf ww) {
        if constexpr (WMODE == 2) return mk(fma_(b.y, ww.y, fma_(b.x, ww.x, p.x)), fma_(-b.x, ww.y, fma_(b.y, ww.x, p.y)));
        else return mk(fma_(-b.y, ww.y, fma_(b.x, ww.x, p.x)), fma_(b.x, ww.y, fma_(b.y, ww.x, p.y)));
    };
    if constexpr (R == 2) {
        cf a = in[I], b = in[I + C];
        if constexpr (WMODE == 0) {
            dit_bfly<2, 0, INV>(a, b);
        } else {
            const cf p = wmul(a, w[0]);
            a = wfma(p, b, w[1]);
            b = mk(fma_(2.0f, p.x, -a.x), fma_(2.0f, p.y, -a.y));
        }
        out[I] = a; out[I + C] = b;
    } else {
        cf a0 = in[I], a1 = in[I + C], a2 = in[I + 2 * C], a3 = in[I + 3 * C];
        if constexpr (WMODE == 0) {
            dit_bfly4<4, 0, INV>(a0, a1, a2, a3);
        } else {
            const cf p0 = wmul(a0, w[0]), p1 = wmul(a1, w[1]);
            const cf t0 = wfma(p0, a2, w[2]), u0 = wfma(p1, a3, w[3]);
            const cf t1 = mk(fma_(2.0f, p0.x, -t0.x), fma_(2.0f, p0.y, -t0.y));
            const cf u1 = mk(fma_(2.0f, p1.x, -u0.x), fma_(2.0f, p1.y, -u0.y));
            a0 = mk(t0.x + u0.x, t0.y + u0.y);
            a2 = mk(t0.x - u0.x, t0.y - u0.y);
            if constexpr (!INV) { a1 = mk(t1.x + u1.y, t1.y - u1.x); a3 = mk(t1.x - u1.y, t1.y + u1.x); }
            else                { a1 = mk(t1.x - u1.y, t1.y + u1.x); a3 = mk(t1.x + u1.y, t1.y - u1.x); }
        }
        out[I] = a0; out[I + C] = a1; out[I + 2 * C] = a2; out[I + 3 * C] = a3;
    }
}

template <int N, bool INV, int I>
MSL_HD void dit_leaves_plain(const cf* in, cf* out) {
    if constexpr (I < dit_leaf_count(N)) {
        dit_leaf<N, INV, 0, I>(in, out, nullptr);
        dit_leaves_plain<N, INV, I + 1>(in, out);
    }
}
// all leaves with the weights in registers: w[n] on element n (conjugated for WMODE 2)
template <int N, bool INV, int WMODE, int I>
MSL_HD void dit_leaves_regs(const cf* in, cf* out, const cf* w) {
    if constexpr (I < dit_leaf_count(N)) {
        constexpr int R = dit_leaf_radix(N), C = N / R;
        cf ww[4];
        ww[0] = w[I]; ww[1] = w[I + C];
        if constexpr (R == 4) { ww[2] = w[I + 2 * C]; ww[3] = w[I + 3 * C]; }
        dit_leaf<N, INV, WMODE, I>(in, out, ww);
        dit_leaves_regs<N, INV, WMODE, I + 1>(in, out, w);
    }
}

template <int N, int I>
MSL_HD void dit_unscramble(const cf* src, cf* dst) {
    if constexpr (I < N) {
        constexpr int F = dit_freq(I, N);
        dst[F] = src[I];
        dit_unscramble<N, I + 1>(src, dst);
    }
}

// the levels above the leaves + the renaming to natural order
template <int N, bool INV, int FENCE = 0>
MSL_HD void dit_upper(cf (&v)[N]) {
    dit<N, 1, INV, true, FENCE>(v);
    cf t[N];
    dit_unscramble<N, 0>(v, t);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = t[i];
}

// natural-order in (array `in`, may be v itself), natural-order out, unnormalised; WMODE / w: weights on the inputs, see dit_leaf
template <int N, bool INV, int WMODE = 0, int FENCE = 0>
MSL_HD void fft_regs_dit(const cf (&in)[N], cf (&v)[N], const cf* w = nullptr) {
    if constexpr (WMODE == 0) dit_leaves_plain<N, INV, 0>(in, v);
    else dit_leaves_regs<N, INV, WMODE, 0>(in, v, w);
    fft_fence<(FENCE > 0)>();
    dit_upper<N, INV, FENCE>(v);
}

}  // namespace msl
