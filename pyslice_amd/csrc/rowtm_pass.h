// The transposing slice-loop pass for lines of a smooth length N = A * B (A, B <= 32, factors 2, 3, 5, 7): a DIRECT mixed-radix
// four-step transform on the register FFTs of fft_regs.h instead of the 2-4 x zero-padded power-of-two convolution of
// rowTB_pass_kernel / rowTB2_pass_kernel (reference Propagate, src/multislice/multislice.py:278-294; the reference's grids are
// int(L / sampling) + 1 points, src/multislice/potentials.py:123-125, so a user who wants a fast grid picks a smooth length).
//
//     out^T = A . t_k . A  in          A = ifft_N . P . fft_N
//
// A line lives in a group of G lanes (G = 32, or 16 when A, B <= 16), in one of two layouts:
//     layout 1 (positions):  lane n1 < A holds x[r A + n1] in register r < B
//     layout 2 (spectrum):   lane k2 < B holds X[k1 B + k2] in register k1 < A
// forward:  B-point register FFT over r -> k2 | x W_N^{n1 k2} | exchange (lane n1, reg k2) -> (lane k2, reg n1) | A-point FFT over n1
// inverse:  A-point inverse over k1 -> n1 | x conj W_N^{n1 k2} | exchange back | B-point inverse over k2 -> r
// Lanes beyond A (layout 1) or B (layout 2) mirror the last active lane (same addresses, same values): no divergence anywhere.
// The exchange goes through the line's own tile row (8-byte accesses; pitches A | 1 and B | 1 are odd: conflict-free both ways).
// Everything around the transform -- next line prefetched in registers, t_k in registers across a chunk of probes, 16-line tile,
// 128-byte transposed segments, ragged last tile -- is rowTB_pass_kernel's.
#pragma once
#include "rowt_pass.h"

// the next line is prefetched in MSL_TM_PFQ parts, behind the first MSL_TM_PFQ of the four exchanges of an iteration
#ifndef MSL_TM_PFQ
#define MSL_TM_PFQ 4
#endif
// ablation bits of tools/rowtm_bench.hip (timing only, results wrong): 1 no exchanges, 2 no table products, 4 no register transforms,
// 8 no tile / barriers / store phase, 16 no prefetch loads
#ifndef MSL_TM_ABL
#define MSL_TM_ABL 0
#endif

namespace msl {

// tile line pitch in float2: room for both exchange images and the line itself; 2 mod 32 (rowT_pass_kernel)
constexpr int rowTM_cs(int A, int B) {
    int m = A * B;
    if (B * (A | 1) > m) m = B * (A | 1);
    if (A * (B | 1) > m) m = A * (B | 1);
    return (m + 31) / 32 * 32 + 2;
}
constexpr size_t rowTM_lds_bytes(int A, int B) { return ((size_t)3 * A * B + (size_t)16 * rowTM_cs(A, B)) * 8; }

// n-point register transform of v[0 .. n), natural order in and out
template <int NP, bool INV, int RM>
__device__ __forceinline__ void tm_fft(float2 (&v)[RM]) {
    static_assert(NP <= RM, "register file of the line");
    if constexpr ((MSL_TM_ABL & 4) != 0) return;
    dif<NP, 1, INV>(v);
    float2 t[NP];
    unscramble<NP, 0>(v, t);
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = t[i];
}

// v[j] *= tab[j * STRIDE] (CONJ: by the conjugate), j < NP, table reads in chunks of 8
template <int NP, int STRIDE, bool CONJ, int RM>
__device__ __forceinline__ void tm_mul(float2 (&v)[RM], const float2* tab) {
    constexpr int TCH = 8;
    if constexpr ((MSL_TM_ABL & 2) != 0) return;
#pragma unroll
    for (int c = 0; c < NP; c += TCH) {
        float2 w[TCH];
#pragma unroll
        for (int j = 0; j < TCH; ++j) if (c + j < NP) w[j] = tab[(c + j) * STRIDE];
#pragma unroll
        for (int j = 0; j < TCH; ++j) if (c + j < NP) v[c + j] = CONJ ? cmulf_conj(v[c + j], w[j]) : cmulf(v[c + j], w[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// (lane a < NA, register b < NB) -> (lane b, register a) through the line's scratch row x (float2, pitch NA | 1)
template <int NA, int NB, int RM>
__device__ __forceinline__ void tm_exchange(float2 (&v)[RM], float2* x, int ln) {
    constexpr int PA = NA | 1;
    const int la = ln < NA ? ln : NA - 1, lb = ln < NB ? ln : NB - 1;
    if constexpr ((MSL_TM_ABL & 1) != 0) return;
    wave_lds_fence();
#pragma unroll
    for (int b = 0; b < NB; ++b) x[b * PA + la] = v[b];
    wave_lds_fence();
#pragma unroll
    for (int a = 0; a < NA; ++a) v[a] = x[lb * PA + a];
    wave_lds_fence();
}

template <int A, int B, int G>
__global__ void __launch_bounds__(16 * G, 2) rowTM_pass_kernel(RowTJob job) {
    constexpr int N = A * B, RM = A > B ? A : B, LINES = 16, NT = LINES * G;
    constexpr int CS = rowTM_cs(A, B);
    constexpr int TPS = LINES / 2, POS_PER_IT = NT / TPS, NIT = (N + POS_PER_IT - 1) / POS_PER_IT;
    static_assert(A <= G && B <= G && 64 % G == 0, "a line's lanes inside one wave");
    static_assert(fft_smooth7(A) && fft_smooth7(B), "radices 2, 3, 4, 5, 7");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw1 = reinterpret_cast<float2*>(smem_raw);        // [k2 A + n1] = W_N^{n1 k2}
    float2* tw2 = tw1 + N;                                    // [n1 B + k2] = W_N^{n1 k2}
    float2* pl = tw2 + N;                                     // Fresnel factor / N, natural order k = k1 B + k2
    float2* tile = pl + N;                                    // LINES * CS
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += NT) { tw1[i] = job.tw[i]; tw2[i] = job.tw[N + i]; pl[i] = job.pl[i]; }
    __syncthreads();
    const int grp = tid / G, ln = tid % G;
    const int lnA = ln < A ? ln : A - 1, lnB = ln < B ? ln : B - 1;
    const int q = tid % TPS, r0 = tid / TPS;
    float2* myrow = tile + grp * CS;
    const int lblocks = (job.n_lines + LINES - 1) / LINES;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const int n_items = lblocks * pchunks;
    const int step_lb = (int)gridDim.x / pchunks, step_pc = (int)gridDim.x % pchunks;
    auto line_ptr = [&](int lbb, int pcc, int kk) {
        const int L = min(lbb * LINES + grp, job.n_lines - 1);
        return job.in + (long long)(pcc * PC + kk) * job.in_image_stride + (long long)L * job.in_pitch + lnA;
    };
    int item = blockIdx.x;
    int lb = item / pchunks, pc = item - lb * pchunks, k = 0;
    float2 vn[B];
    if (item < n_items) {
        const float2* r = line_ptr(lb, pc, 0);
#pragma unroll
        for (int j = 0; j < B; ++j) vn[j] = ld_stream(r + j * A);
    }
    float2 tv[B];
    while (item < n_items) {
        float2 v[RM];
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = vn[j];
        const int p = pc * PC + k;
        const int cur_lb = lb;
        if (k == 0) {
            const float2* trow = job.trans + frame_off(job, pc * PC) + (long long)min(lb * LINES + grp, job.n_lines - 1) * N + lnA;
#pragma unroll
            for (int j = 0; j < B; ++j) tv[j] = ld_stream(trow + j * A);
        }
        int nitem = item, nlb = lb, npc = pc, nk = k + 1;
        if (nk >= min(PC, job.n_images - pc * PC)) {
            nk = 0; nitem = item + (int)gridDim.x; nlb = lb + step_lb; npc = pc + step_pc;
            if (npc >= pchunks) { npc -= pchunks; ++nlb; }
        }
        // prefetch of the next line in four parts, unconditional (past the last item the loads re-read the current line)
        const bool more = nitem < n_items;
        const float2* nptr = line_ptr(more ? nlb : lb, more ? npc : pc, more ? nk : k);
        auto pfx = [&](auto i_c) {
            constexpr int I = decltype(i_c)::value;
            constexpr int LO = I < MSL_TM_PFQ ? B * I / MSL_TM_PFQ : B, HI = I < MSL_TM_PFQ ? B * (I + 1) / MSL_TM_PFQ : B;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = LO; j < HI; ++j) if constexpr ((MSL_TM_ABL & 16) == 0) vn[j] = ld_stream(nptr + j * A);
            __builtin_amdgcn_sched_barrier(0);
        };
        // A = ifft_N . P . fft_N, layout 1 in and out; two prefetch slots
        auto a_op = [&](auto i_c) {
            constexpr int I = decltype(i_c)::value;
            tm_fft<B, false>(v);
            tm_mul<B, A, false>(v, tw1 + lnA);
            tm_exchange<A, B>(v, myrow, ln);
            pfx(MSL_IC(I));
            tm_fft<A, false>(v);
            tm_mul<A, B, false>(v, pl + lnB);
            tm_fft<A, true>(v);
            tm_mul<A, B, true>(v, tw2 + lnB);
            tm_exchange<B, A>(v, myrow, ln);
            pfx(MSL_IC(I + 1));
            tm_fft<B, true>(v);
        };
        if (job.flags & P2_PRE_A) a_op(MSL_IC(0)); else { pfx(MSL_IC(0)); pfx(MSL_IC(1)); }
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = cmulf(v[j], tv[j]);
        if (job.flags & P2_POST_A) a_op(MSL_IC(2)); else { pfx(MSL_IC(2)); pfx(MSL_IC(3)); }
        if constexpr ((MSL_TM_ABL & 8) != 0) {                      // keep the result alive: one store per lane
            float2 acc = v[0];
#pragma unroll
            for (int j = 1; j < B; ++j) { acc.x += v[j].x; acc.y += v[j].y; }
            if (acc.x == 12345.678f) job.out[tid] = acc;
            item = nitem; lb = nlb; pc = npc; k = nk;
            continue;
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < B; ++j) myrow[j * A + lnA] = v[j];
        lds_barrier();
        float2* dst = job.out + (long long)p * job.out_image_stride + cur_lb * LINES;
        int off0 = 2 * q + r0 * job.out_pitch;
        asm volatile("" : "+v"(off0));
        const int ostep = POS_PER_IT * job.out_pitch;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pos = r0 + POS_PER_IT * i;
            if (pos < N) {
                const float2 a = tile[(2 * q) * CS + pos], b = tile[(2 * q + 1) * CS + pos];
                st_stream(dst + (off0 + i * ostep), a.x, a.y, b.x, b.y);
            }
        }
        lds_barrier();
        item = nitem; lb = nlb; pc = npc; k = nk;
    }
}

// ---- N = 2 A B on one wave per line (lengths up to 1728) --------------------------------------------------------------------------
// The same transform with 2 A in the place of A, and the 2 A-point register transform of layout 2 split over a PAIR of lanes (the
// scheme of fft2048_wave in fft_pow2.h): a radix-2 step across the pair through DPP and an A-point transform in each lane.
//     layout 1:  lane n1 < 2 A holds x[r 2A + n1] in register r < B
//     layout 2:  lane L = 2 k2 + h (k2 < B, h < 2) holds X[(2 q + h) B + k2] in register q < A
// forward:  B-point FFT over r -> k2 | x W_N^{n1 k2} | exchange: lane (k2, h) gets n1 = m + A h in register m | pair step
//           (h = 0: a = x[m] + x[m + A]; h = 1: d = (x[m] - x[m + A]) W_2A^m, the partner's value by DPP) | A-point FFT over m -> q
// inverse:  A-point inverse over q -> m | pair step (the odd lane's conj W_2A^m first) | x conj W_N^{n1 k2} | exchange back | B-point inverse
// Tables in LDS: tw1[k2 2A + n1] = W_N^{n1 k2}; tw2[m 2B + L] = W_N^{(m + A h) k2} and the Fresnel factor pl[q 2B + L] in lane
// order (permuted from the natural table while it is copied in); wp[h A + m] = h ? W_2A^m : 1.
// Exchange 1 writes row k2 of pitch P1 (columns n1 < A at [0, A), the others from AH): P1 = 2 mod 4 and AH odd make the pair
// lanes' 8-byte reads conflict-free; exchange 2 writes row n1 of odd pitch B | 1.  Tiles of 8 lines (64-byte transposed segments).
constexpr int rowTM2_ah(int A) { return A | 1; }
constexpr int rowTM2_p1(int A) { int p = rowTM2_ah(A) + A; while (p % 4 != 2) ++p; return p; }
constexpr int rowTM2_cs(int A, int B) {
    int m = 2 * A * B;
    if (B * rowTM2_p1(A) > m) m = B * rowTM2_p1(A);
    if (2 * A * (B | 1) > m) m = 2 * A * (B | 1);
    return (m + 15) / 16 * 16 + 4;                  // 4 mod 16: the store phase's half-wave reads 4 rows (2 q) x 8 consecutive positions
}
constexpr size_t rowTM2_lds_bytes(int A, int B) { return ((size_t)6 * A * B + 2 * A + (size_t)8 * rowTM2_cs(A, B)) * 8; }

__device__ __forceinline__ float tm_dpp_swap_pair(float x) {          // value of the neighbouring lane (L ^ 1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}

template <int A, int B>
__global__ void __launch_bounds__(512, 2) rowTM2_pass_kernel(RowTJob job) {
    constexpr int A2 = 2 * A, B2 = 2 * B, N = A2 * B, RM = A > B ? A : B, LINES = 8, NT = 512;
    constexpr int AH = rowTM2_ah(A), P1 = rowTM2_p1(A), P2 = B | 1;
    constexpr int CS = rowTM2_cs(A, B);
    constexpr int TPS = LINES / 2, POS_PER_IT = NT / TPS, NIT = (N + POS_PER_IT - 1) / POS_PER_IT;
    static_assert(A2 <= 64 && B2 <= 64, "one wave per line");
    static_assert(P1 % 4 == 2 && P1 >= AH + A && (AH & 1), "conflict-free pair reads");
    static_assert(fft_smooth7(A) && fft_smooth7(B), "radices 2, 3, 4, 5, 7");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw1 = reinterpret_cast<float2*>(smem_raw);        // N
    float2* tw2 = tw1 + N;                                    // N, lane order
    float2* pl = tw2 + N;                                     // N, lane order
    float2* wp = pl + N;                                      // 2 A
    float2* tile = wp + A2;                                   // LINES * CS
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += NT) {
        tw1[i] = job.tw[i]; tw2[i] = job.tw[N + i];
        const int qq = i / B2, LL = i - qq * B2;              // lane-order position q 2B + L  <-  natural k = (2 q + h) B + k2
        pl[i] = job.pl[(2 * qq + (LL & 1)) * B + (LL >> 1)];
    }
    for (int i = tid; i < A2; i += NT) wp[i] = job.tw[2 * N + i];
    __syncthreads();
    const int wv = tid >> 6, L = tid & 63;
    const int n1c = L < A2 ? L : A2 - 1;                      // layout 1 lane (idle lanes mirror the last one)
    const int Lc = L < B2 ? L : B2 - 2 + (L & 1);             // layout 2 lane: pairs stay pairs
    const int k2c = Lc >> 1, hh = Lc & 1;
    const float sgn = hh ? -1.f : 1.f;
    const int q = tid % TPS, r0 = tid / TPS;
    float2* myrow = tile + wv * CS;
    float2* x1w = myrow + (n1c < A ? n1c : AH + n1c - A);     // exchange 1: write [k2 P1], read [m]
    const float2* x1r = myrow + k2c * P1 + hh * AH;
    float2* x2w = myrow + hh * A * P2 + k2c;                  // exchange 2: write [m P2], read [k2]
    const float2* x2r = myrow + n1c * P2;
    const int lblocks = (job.n_lines + LINES - 1) / LINES;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const int n_items = lblocks * pchunks;
    const int step_lb = (int)gridDim.x / pchunks, step_pc = (int)gridDim.x % pchunks;
    auto line_ptr = [&](int lbb, int pcc, int kk) {
        const int Ln = min(lbb * LINES + wv, job.n_lines - 1);
        return job.in + (long long)(pcc * PC + kk) * job.in_image_stride + (long long)Ln * job.in_pitch + n1c;
    };
    int item = blockIdx.x;
    int lb = item / pchunks, pc = item - lb * pchunks, k = 0;
    float2 vn[B];
    if (item < n_items) {
        const float2* r = line_ptr(lb, pc, 0);
#pragma unroll
        for (int j = 0; j < B; ++j) vn[j] = ld_stream(r + j * A2);
    }
    float2 tv[B];
    while (item < n_items) {
        float2 v[RM];
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = vn[j];
        const int p = pc * PC + k;
        const int cur_lb = lb;
        if (k == 0) {
            const float2* trow = job.trans + frame_off(job, pc * PC) + (long long)min(lb * LINES + wv, job.n_lines - 1) * N + n1c;
#pragma unroll
            for (int j = 0; j < B; ++j) tv[j] = ld_stream(trow + j * A2);
        }
        int nitem = item, nlb = lb, npc = pc, nk = k + 1;
        if (nk >= min(PC, job.n_images - pc * PC)) {
            nk = 0; nitem = item + (int)gridDim.x; nlb = lb + step_lb; npc = pc + step_pc;
            if (npc >= pchunks) { npc -= pchunks; ++nlb; }
        }
        const bool more = nitem < n_items;
        const float2* nptr = line_ptr(more ? nlb : lb, more ? npc : pc, more ? nk : k);
        auto pfx = [&](auto i_c) {
            constexpr int I = decltype(i_c)::value;
            constexpr int LO = I < MSL_TM_PFQ ? B * I / MSL_TM_PFQ : B, HI = I < MSL_TM_PFQ ? B * (I + 1) / MSL_TM_PFQ : B;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = LO; j < HI; ++j) if constexpr ((MSL_TM_ABL & 16) == 0) vn[j] = ld_stream(nptr + j * A2);
            __builtin_amdgcn_sched_barrier(0);
        };
        // radix-2 step across the lane pair: v[m] <- partner + sgn * own, the odd lane's twiddle behind it (forward) or in front (inverse)
        auto pair_step = [&](auto inv_c) {
            constexpr bool INV = decltype(inv_c)::value;
            constexpr int TCH = 8;
#pragma unroll
            for (int c = 0; c < A; c += TCH) {
                float2 w[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) if (c + j < A) w[j] = wp[hh * A + c + j];
#pragma unroll
                for (int j = 0; j < TCH; ++j) if (c + j < A) {
                    float2 o = v[c + j];
                    if constexpr (INV) o = cmulf_conj(o, w[j]);
                    float2 t = make_float2(fmaf(sgn, o.x, tm_dpp_swap_pair(o.x)), fmaf(sgn, o.y, tm_dpp_swap_pair(o.y)));
                    if constexpr (!INV) t = cmulf(t, w[j]);
                    v[c + j] = t;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto a_op = [&](auto i_c) {
            constexpr int I = decltype(i_c)::value;
            tm_fft<B, false>(v);
            tm_mul<B, A2, false>(v, tw1 + n1c);
            if constexpr ((MSL_TM_ABL & 1) == 0) {
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < B; ++b) x1w[b * P1] = v[b];
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < A; ++m) v[m] = x1r[m];
            wave_lds_fence();
            }
            pfx(MSL_IC(I));
            pair_step(std::false_type{});
            tm_fft<A, false>(v);
            tm_mul<A, B2, false>(v, pl + Lc);
            tm_fft<A, true>(v);
            pair_step(std::true_type{});
            tm_mul<A, B2, true>(v, tw2 + Lc);
            if constexpr ((MSL_TM_ABL & 1) == 0) {
            wave_lds_fence();
#pragma unroll
            for (int m = 0; m < A; ++m) x2w[m * P2] = v[m];
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < B; ++b) v[b] = x2r[b];
            wave_lds_fence();
            }
            pfx(MSL_IC(I + 1));
            tm_fft<B, true>(v);
        };
        if (job.flags & P2_PRE_A) a_op(MSL_IC(0)); else { pfx(MSL_IC(0)); pfx(MSL_IC(1)); }
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = cmulf(v[j], tv[j]);
        if (job.flags & P2_POST_A) a_op(MSL_IC(2)); else { pfx(MSL_IC(2)); pfx(MSL_IC(3)); }
        if constexpr ((MSL_TM_ABL & 8) != 0) {
            float2 acc = v[0];
#pragma unroll
            for (int j = 1; j < B; ++j) { acc.x += v[j].x; acc.y += v[j].y; }
            if (acc.x == 12345.678f) job.out[tid] = acc;
            item = nitem; lb = nlb; pc = npc; k = nk;
            continue;
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < B; ++j) myrow[j * A2 + n1c] = v[j];
        lds_barrier();
        float2* dst = job.out + (long long)p * job.out_image_stride + cur_lb * LINES;
        int off0 = 2 * q + r0 * job.out_pitch;
        asm volatile("" : "+v"(off0));
        const int ostep = POS_PER_IT * job.out_pitch;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pos = r0 + POS_PER_IT * i;
            if (pos < N) {
                const float2 a = tile[(2 * q) * CS + pos], b = tile[(2 * q + 1) * CS + pos];
                st_stream(dst + (off0 + i * ostep), a.x, a.y, b.x, b.y);
            }
        }
        lds_barrier();
        item = nitem; lb = nlb; pc = npc; k = nk;
    }
}

// launch of the instantiation for a line length (slice_mixed_a.hip / _b.hip); false: no kernel for n.  rowTM_factors: its (A, B, G)
bool rowTM_factors(int n, int* A, int* B, int* G);
bool rowTM_launch(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream);

}  // namespace msl
