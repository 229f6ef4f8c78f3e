// Generic batched line FFT for gfx950: LDS-resident Stockham autosort, mixed radix
// {8,4,2,3,5,7,11,13}, arbitrary line stride.  This is the any-size path (non power-of-two
// grids such as the reference's 501x491 test grid need it) and the on-device
// cross-check for the register-resident power-of-two kernels in fft_pow2.h.
//
// One workgroup owns a tile of C lines of length N held as LDS[c][n] (float2).  A launch
// performs:  load -> [FFT a] -> [x M1] -> [FFT b] -> store(x M2, scale, index shift, mode)
// so the fused slice-loop passes (row: ifft_y, x t, fft_y, x Py;  column: fft_x, x Px, ifft_x)
// and the potential / TACAW epilogues are single launches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_regs.h"

namespace msl {

#define MSL_MAX_STAGES 16
#define MSL_GEN_E 16            // complex values held per thread between the two barriers of a stage
#define MSL_GEN_STEPS 4         // FFT / multiply steps of one launch (the one-pass slice kernel needs four)
#define MSL_GEN_HEADER 512      // bytes of per-line address bases in front of the LDS tile

// complex values per thread the block size is computed from (see stockham_stage); ceil5 selects the kernel variant
// whose radix-5 stages take a fourth, partly idle butterfly per thread so that the block stays C*N/16 threads
inline int gen_elems_per_thread(int r, bool ceil5) { return (r == 5 && ceil5) ? MSL_GEN_E : (MSL_GEN_E / r) * r; }

enum { MUL_NONE = 0, MUL_ARRAY = 1, MUL_VEC = 2 };
enum { STORE_C64 = 0, STORE_POTENTIAL = 1, STORE_INTENSITY = 2 };

struct LineJob {
    const float2* in;
    float2* out;
    float* out_real;            // STORE_POTENTIAL: V (may be null);  STORE_INTENSITY: intensity
    const float2* tw;           // W_N^j = exp(-2 pi i j / N), j < N (device)
    const float2* m2;           // at store
    long long n_lines;
    long long in_es, in_ls, in_is;     // element / line / image strides (float2 units)
    long long out_es, out_ls, out_is;
    long long m1_ls;            // MUL_ARRAY: m1[r*m1_ls + n]
    long long m2_ls;
    int N, C, lines_per_image;
    int contiguous_lines;       // 1: neighbouring lines are neighbouring addresses (column / time pass)
    // the launch is a short program on the LDS tile: step i = [FFT fft[i]] then [x mul[i]]; then the store (x m2, scale ...)
    int n_steps;
    int fft[MSL_GEN_STEPS];     // 0 none, +1 forward, -1 inverse (both unnormalised)
    int mkind[MSL_GEN_STEPS];   // MUL_NONE / MUL_VEC (mul[i][n]) / MUL_ARRAY (mul[i][r*m1_ls + n]; one array step at most)
    const float2* mul[MSL_GEN_STEPS];
    int m2_kind;
    int out_contiguous;         // store mapping: 1 = neighbouring lines are neighbouring output addresses (transposing store)
    int store_mode;
    int shift_n, shift_r;       // out index: ((n+shift_n)%N, (r+shift_r)%lines_per_image)
    int win_n0, win_nn, win_r0, win_nr;   // win_nn > 0: keep only out indices [win_n0, +win_nn) x [win_r0, +win_nr), rebased to 0
    int n_stages;
    int radix[MSL_MAX_STAGES];
    int M;                      // transform length of the Stockham stages: N, or >= 2N-1 for Bluestein lines
    const float2* chirp;        // Bluestein: w[n] = exp(-i pi n^2 / N), n < N
    const float2* bfilt;        // Bluestein: FFT_M of the wrapped conj chirp, pre-divided by M
    // frame batching: image i is member i % group of frame i / group; its output goes to (i % group) * out_is +
    // (i / group) * out_gs and its MUL_ARRAY factors start (i / group) * m_gs further on (group == 0: off)
    int group;
    long long out_gs, m_gs;
    int npad;                   // LDS line pitch (float2)
    int tw_in_lds;
    float scale;
    float sigma;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i*s  (s=+1 forward rotation, s=-1 inverse)
__device__ __forceinline__ float2 rot_mi(float2 a, float s) { return make_float2(s * a.y, -s * a.x); }

// x / d for 0 <= x < 2^21 with inv = 1.0f / d: exact (the +0.5 keeps the float product away from integer boundaries)
__device__ __forceinline__ int fast_div(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// cos / sin (2 pi m / R) as compile-time constants for the small-prime butterflies
template <int R>
struct PrimeTw {
    float c[R], s[R];
    constexpr PrimeTw() : c{}, s{} {
        for (int m = 0; m < R; ++m) { c[m] = (float)cx_cos2pi(m, R); s[m] = (float)cx_sin2pi(m, R); }
    }
};

template <int R>
__device__ __forceinline__ void butterfly(float2 (&v)[R], float s) {
    if constexpr (R == 2) {
        float2 a = v[0], b = v[1];
        v[0] = cadd(a, b); v[1] = csub(a, b);
    } else if constexpr (R == 4) {
        float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        float2 t2 = cadd(v[1], v[3]), t3 = rot_mi(csub(v[1], v[3]), s);
        v[0] = cadd(t0, t2); v[1] = cadd(t1, t3); v[2] = csub(t0, t2); v[3] = csub(t1, t3);
    } else if constexpr (R == 8) {
        const float h = 0.70710678118654752440f;
        float2 e[4] = {v[0], v[2], v[4], v[6]};
        float2 o[4] = {v[1], v[3], v[5], v[7]};
        butterfly<4>(e, s);
        butterfly<4>(o, s);
        // o[k] *= W8^k (forward: exp(-i pi k/4)); s flips the sign of the imaginary part
        float2 w1 = make_float2(h, -s * h), w3 = make_float2(-h, -s * h);
        o[1] = cmul(o[1], w1);
        o[2] = rot_mi(o[2], s);
        o[3] = cmul(o[3], w3);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = cadd(e[k], o[k]); v[k + 4] = csub(e[k], o[k]); }
    } else {
        // small-prime DFT, O(R^2 / 2): with a_t = v_t + v_{R-t}, b_t = v_t - v_{R-t}
        //   X_q, X_{R-q} = (v_0 + sum_t cos(2 pi q t / R) a_t)  -+  i s sum_t sin(2 pi q t / R) b_t
        constexpr PrimeTw<R> T{};
        constexpr int H = (R - 1) / 2;
        float2 a[H + 1], b[H + 1];
#pragma unroll
        for (int t = 1; t <= H; ++t) { a[t] = cadd(v[t], v[R - t]); b[t] = csub(v[t], v[R - t]); }
        const float2 v0 = v[0];
        float2 x0 = v0;
#pragma unroll
        for (int t = 1; t <= H; ++t) x0 = cadd(x0, a[t]);
        v[0] = x0;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            float2 ev = v0, od = make_float2(0.f, 0.f);
#pragma unroll
            for (int t = 1; t <= H; ++t) {
                const float c = T.c[(q * t) % R], sn = T.s[(q * t) % R];
                ev.x = fmaf(c, a[t].x, ev.x); ev.y = fmaf(c, a[t].y, ev.y);
                od.x = fmaf(sn, b[t].x, od.x); od.y = fmaf(sn, b[t].y, od.y);
            }
            const float2 r = rot_mi(od, s);                      // -i s od
            v[q] = cadd(ev, r); v[R - q] = csub(ev, r);
        }
    }
}

// One Stockham stage over the whole LDS tile, in place through registers.
template <int R, bool CEIL5>
__device__ __forceinline__ void stockham_stage(float2* tile, const float2* tw, int N, int npad, int C, int Ns,
                                               float s, int tid, int nthreads) {
    // butterflies per thread, at most MSL_GEN_E values in registers between the barriers: the host sizes the block so that
    // nthreads * ITERS >= butterflies for every radix of the plan (gen_elems_per_thread).  Rounding down keeps register
    // pressure low (measured 10% faster at 480^2, 360^2, 448^2); CEIL5 rounds radix 5 up instead, for lengths like 500
    // where rounding down would need a ninth wave per workgroup and halve the workgroups per CU.
    constexpr int ITERS = (R == 5 && CEIL5) ? 4 : MSL_GEN_E / R;
    const int nb = N / R;                 // butterflies per line
    const int total = nb * C;
    const int twstep = N / (Ns * R);
    const float inv_nb = 1.0f / (float)nb, inv_ns = 1.0f / (float)Ns;
    float2 regs[ITERS][R];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        int idx = tid + it * nthreads;
        if (idx < total) {
            const int c = fast_div(idx, inv_nb), j = idx - c * nb;
            const float2* line = tile + c * npad;
            float2 v[R];
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = line[j + t * nb];
            if (Ns > 1) {                 // the first stage's twiddles are all 1
                const int k = j - fast_div(j, inv_ns) * Ns;
                const int kstep = k * twstep;             // k * t * twstep < N for k < Ns, t < R: no wrap
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    float2 w = tw[kstep * t];
                    w.y *= s;             // table holds the forward (exp(-i..)) twiddles
                    v[t] = cmul(v[t], w);
                }
            }
            butterfly<R>(v, s);
#pragma unroll
            for (int t = 0; t < R; ++t) regs[it][t] = v[t];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        int idx = tid + it * nthreads;
        if (idx < total) {
            const int c = fast_div(idx, inv_nb), j = idx - c * nb;
            const int k = j - fast_div(j, inv_ns) * Ns;
            float2* line = tile + c * npad;
            const int base = (j - k) * R + k;
#pragma unroll
            for (int t = 0; t < R; ++t) line[base + t * Ns] = regs[it][t];
        }
    }
    __syncthreads();
}

// RSET selects which radices a kernel instantiation carries (register pressure follows the
// largest one): 0 = {2,4,8}, 1 = + {3,5,7}, 2 = + {11,13}.
template <int RSET, bool CEIL5>
__device__ __forceinline__ void tile_stages(float2* tile, const float2* tw, const LineJob& job, int C, int dir, int tid,
                                            int nthreads) {
    const float s = dir > 0 ? 1.0f : -1.0f;
    int Ns = 1;
    for (int st = 0; st < job.n_stages; ++st) {
        const int R = job.radix[st];
        switch (R) {
            case 2: stockham_stage<2, CEIL5>(tile, tw, job.M, job.npad, C, Ns, s, tid, nthreads); break;
            case 4: stockham_stage<4, CEIL5>(tile, tw, job.M, job.npad, C, Ns, s, tid, nthreads); break;
            case 8: stockham_stage<8, CEIL5>(tile, tw, job.M, job.npad, C, Ns, s, tid, nthreads); break;
            default:
                if constexpr (RSET >= 1) {
                    if (R == 3) stockham_stage<3, CEIL5>(tile, tw, job.M, job.npad, C, Ns, s, tid, nthreads);
                    else if (R == 5) stockham_stage<5, CEIL5>(tile, tw, job.M, job.npad, C, Ns, s, tid, nthreads);
                    else if (R == 7) stockham_stage<7, CEIL5>(tile, tw, job.M, job.npad, C, Ns, s, tid, nthreads);
                }
                if constexpr (RSET >= 2) {
                    if (R == 11) stockham_stage<11, CEIL5>(tile, tw, job.M, job.npad, C, Ns, s, tid, nthreads);
                    else if (R == 13) stockham_stage<13, CEIL5>(tile, tw, job.M, job.npad, C, Ns, s, tid, nthreads);
                }
                break;
        }
        Ns *= R;
    }
}

// Line transform of logical length N.  Native when the stages run on N itself; otherwise Bluestein's
// chirp-z: X[k] = w[k] * sum_n (x[n] w[n]) conj(w)[k-n], the convolution done with length-M FFTs in LDS
// (any N up to 4096, e.g. the reference's 501 x 491 test grid, src/unittests/00_probe.py:7-8).
template <int RSET, bool CEIL5>
__device__ __forceinline__ void tile_fft(float2* tile, const float2* tw, const LineJob& job, int C, int dir, int tid,
                                         int nthreads) {
    if (job.M == job.N) { tile_stages<RSET, CEIL5>(tile, tw, job, C, dir, tid, nthreads); return; }
    const int N = job.N, M = job.M, npad = job.npad;
    const float inv_m = 1.0f / (float)M, inv_n = 1.0f / (float)N;
    for (int e = tid; e < C * M; e += nthreads) {
        const int c = fast_div(e, inv_m), n = e - c * M;
        float2 x = make_float2(0.f, 0.f);
        if (n < N) {
            x = tile[(size_t)c * npad + n];
            if (dir < 0) x.y = -x.y;
            x = cmul(x, job.chirp[n]);
        }
        tile[(size_t)c * npad + n] = x;
    }
    __syncthreads();
    tile_stages<RSET, CEIL5>(tile, tw, job, C, +1, tid, nthreads);
    for (int e = tid; e < C * M; e += nthreads) {
        const int c = fast_div(e, inv_m), n = e - c * M;
        float2* p = tile + (size_t)c * npad + n;
        *p = cmul(*p, job.bfilt[n]);
    }
    __syncthreads();
    tile_stages<RSET, CEIL5>(tile, tw, job, C, -1, tid, nthreads);
    for (int e = tid; e < C * N; e += nthreads) {
        const int c = fast_div(e, inv_n), n = e - c * N;
        float2* p = tile + (size_t)c * npad + n;
        float2 y = cmul(*p, job.chirp[n]);
        if (dir < 0) y.y = -y.y;
        *p = y;
    }
    __syncthreads();
}

template <int RSET, bool CEIL5>
__global__ void __launch_bounds__(1024) line_fft_kernel(LineJob job) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // per-line address bases of this tile (at most 16 lines): the 64-bit divisions happen once per line, not per element
    // (first MSL_GEN_HEADER bytes of the dynamic LDS block)
    long long* s_in = reinterpret_cast<long long*>(smem_raw);
    long long* s_out = s_in + 16;
    long long* s_m1 = s_out + 16;
    long long* s_m2 = s_m1 + 16;
    float2* tile = reinterpret_cast<float2*>(smem_raw + MSL_GEN_HEADER);
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int N = job.N, npad = job.npad;
    const long long line0 = (long long)blockIdx.x * job.C;
    const int C = (int)min((long long)job.C, job.n_lines - line0);
    if (tid < C) {
        const long long l = line0 + tid;
        const long long img = l / job.lines_per_image;
        const long long r = l - img * job.lines_per_image;
        long long ro = r + job.shift_r;
        if (ro >= job.lines_per_image) ro -= job.lines_per_image;
        bool keep = true;
        if (job.win_nn > 0) { ro -= job.win_r0; keep = (ro >= 0 && ro < job.win_nr); }
        const long long frame = job.group > 0 ? img / job.group : 0;
        const long long member = job.group > 0 ? img - frame * job.group : img;
        s_in[tid] = img * job.in_is + r * job.in_ls;
        s_out[tid] = keep ? member * job.out_is + frame * job.out_gs + ro * job.out_ls : -1;
        s_m1[tid] = r * job.m1_ls + frame * job.m_gs;
        s_m2[tid] = r * job.m2_ls + frame * job.m_gs;
    }
    const float2* tw = job.tw;
    if (job.tw_in_lds) {
        float2* tws = tile + (size_t)job.C * npad;
        for (int i = tid; i < job.M; i += nthreads) tws[i] = job.tw[i];
        tw = tws;
    }
    __syncthreads();
    const int elems = C * N;
    const float inv_c = 1.0f / (float)C, inv_n = 1.0f / (float)N;
    // element e of the tile -> (line c, position n); neighbouring threads touch neighbouring addresses
    auto split = [&](int e, int lines_fastest, int& c, int& n) {
        if (lines_fastest) { n = fast_div(e, inv_c); c = e - n * C; } else { c = fast_div(e, inv_n); n = e - c * N; }
    };
    // ---- load
    for (int e = tid; e < elems; e += nthreads) {
        int c, n;
        split(e, job.contiguous_lines, c, n);
        float2 x = job.in[s_in[c] + (long long)n * job.in_es];
        if (job.store_mode == STORE_INTENSITY) {
            // TACAW: any constant may be subtracted from a time line (only the DC bin sees it, and that bin is zeroed below); the
            // line's first sample keeps the float32 transform at the size of the thermal part where the mean is orders of magnitude
            // above it (the register kernels do the same: tacaw_time.h)
            const float2 r = job.in[s_in[c]];
            x.x -= r.x; x.y -= r.y;
        }
        tile[c * npad + n] = x;
    }
    __syncthreads();
    for (int st = 0; st < job.n_steps; ++st) {
        if (job.fft[st]) tile_fft<RSET, CEIL5>(tile, tw, job, C, job.fft[st], tid, nthreads);
        if (job.mkind[st] != MUL_NONE) {
            const float2* mp = job.mul[st];
            const bool vec = job.mkind[st] == MUL_VEC;
            for (int e = tid; e < elems; e += nthreads) {
                const int c = fast_div(e, inv_n), n = e - c * N;
                const float2 m = vec ? mp[n] : mp[s_m1[c] + n];
                float2* p = tile + c * npad + n;
                *p = cmul(*p, m);
            }
            __syncthreads();
        }
    }
    // ---- store
    for (int e = tid; e < elems; e += nthreads) {
        int c, n;
        split(e, job.out_contiguous, c, n);
        const long long ob = s_out[c];
        if (ob < 0) continue;
        float2 v = tile[c * npad + n];
        if (job.m2_kind == MUL_VEC) v = cmul(v, job.m2[n]);
        else if (job.m2_kind == MUL_ARRAY) v = cmul(v, job.m2[s_m2[c] + n]);
        v.x *= job.scale; v.y *= job.scale;
        int no = n + job.shift_n; if (no >= N) no -= N;
        if (job.win_nn > 0) {
            no -= job.win_n0;
            if (no < 0 || no >= job.win_nn) continue;
        }
        const long long o = ob + (long long)no * job.out_es;
        if (job.store_mode == STORE_C64) {
            job.out[o] = v;
        } else if (job.store_mode == STORE_POTENTIAL) {
            // V = Re(ifft2(R)) / (dx^2 dy^2) (scale), t = exp(i sigma V)
            if (job.out_real) job.out_real[o] = v.x;
            float sn, cs;
            sincosf(job.sigma * v.x, &sn, &cs);
            job.out[o] = make_float2(cs, sn);
        } else {
            // TACAW: zero the DC bin (== subtracting the time mean before the FFT), |.|^2
            float inten = (n == 0) ? 0.0f : fmaf(v.x, v.x, v.y * v.y);
            job.out_real[o] = inten;
        }
    }
}

}  // namespace msl
