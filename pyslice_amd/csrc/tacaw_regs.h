// TACAW time -> frequency transform for smooth frame counts on compile-time mixed-radix register networks (fft_regs.h):
// time_direct_kernel<T> (a lane per pixel, 16 <= T <= 128) and time_split_kernel<TP, L, HB> (T = L x TP up to 1024, split over the
// waves of a workgroup).  tacaw_data.py:94-104:  intensity[p, w, k] = | fftshift_t fft_t( Psi[p, :, k] - <Psi[p, :, k]>_t ) |^2.
// Independent of the slice-loop kernels (fft_pow2.h), so that its 70 instantiations compile in translation units of their own
// (tacaw_direct.hip, tacaw_split.hip, tacaw_split2.hip); the chirp-z kernel for the other frame counts is in tacaw_time.h.
#pragma once
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>
#include "fft_regs.h"
#include "kernel_util.h"

namespace msl {

struct TimeJob {
    const float2* in;       // (n_images, T, npix) c64
    float* out;             // (n_images, T, npix) f32, frequency axis fftshifted
    const float2* tw;       // (M) four-step twiddles T[k1 R + n2] = exp(-2 pi i k1 n2 / M)
    const float2* bf;       // (M/2 + 2) chirp filter, first half (even sequence)
    const float2* bw;       // (M/2) chirp w[n], zero for n >= T
    long long image_stride; // T * npix
    int npix, n_images, T;
};

// ---- smooth frame counts up to 128 (2^a 3^b 5^c: 100 = 4.5.5, the reference notebook's run; 64, 96, 120, 128 ...): no convolution ----
// One LANE holds a pixel's whole time line in registers and transforms it with the compile-time mixed-radix network of
// fft_regs.h: ~24 VALU instructions per sample at T = 100 where chirp-z on a 16-lane group needs ~160, no LDS, no cross-lane
// traffic.  A wave's load covers 64 neighbouring pixels of one frame (512-byte runs, its stores 256-byte runs), so nothing is
// staged either; the transform leaves the spectrum in the network's digit-reversed order, which the store addresses absorb.
// The next tile's samples fly in a second register set (the AGPR half of the unified file: one wave per SIMD, 512 registers per
// lane) while the current tile is transformed, PF of them -- as many as fit beside 2 T data registers.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Row loads of the per-lane kernel: raw buffer loads -- a descriptor in 4 SGPRs for the image, the row's byte offset in one more
// (SALU arithmetic) and ONE 32-bit lane offset register for all rows.  Written as plain pointer arithmetic the same loads come
// out with a 64-bit address pair per row, advanced by VALU adds: two registers per load in flight, which this kernel cannot
// spare.  (LLVM intrinsic declared here; aux 2 = "nt", the non-temporal hint of ld_stream.)
// (msl_raw_buffer_load_f2 / make_raw_rsrc: kernel_util.h)

__host__ __device__ constexpr int tdir_prefetch(int T) { return T < 216 - T ? T : 216 - T; }
constexpr int TDIR_MIN = 16, TDIR_MAX = 128;

template <int T>
__global__ void __launch_bounds__(256) time_direct_kernel(TimeJob job) {
    constexpr int PF = tdir_prefetch(T), half = T / 2, KH = (T + 1) / 2;
    const int tid = threadIdx.x;
    const int tiles_per_image = (job.npix + 255) / 256;
    const long long n_tiles = (long long)tiles_per_image * job.n_images;
    const long long step = gridDim.x;
    // lanes beyond the image's last pixel (ragged last tile) work on its last pixel too: same samples, same instructions, the
    // same values stored to the same addresses -- no branch around the stores, which would also let the compiler sink the whole
    // transform into it, past its scheduling fences.  Addresses: a uniform row
    // pointer (SGPR pair, advanced by the SALU) plus the lane's 32-bit byte offset -- no 64-bit VALU address arithmetic.
    auto column = [&](long long t, msl_i4v& rows, msl_i4v& rows_hi, float*& orow, unsigned& c, bool& live) {
        const int p = __builtin_amdgcn_readfirstlane((int)(t / tiles_per_image));
        const int c0 = __builtin_amdgcn_readfirstlane((int)(t % tiles_per_image) * 256);
        live = c0 + tid < job.npix;
        c = live ? (unsigned)(c0 + tid) : (unsigned)(job.npix - 1);
        rows = make_raw_rsrc(job.in + (long long)p * job.image_stride);
        rows_hi = make_raw_rsrc(job.in + (long long)p * job.image_stride + (long long)KH * job.npix);
        orow = job.out + (long long)p * job.image_stride;
    };
    // rows below KH through the first descriptor, the others through the second: the 32-bit row offsets stay below 2^32 for
    // every image the host sends here (KH npix 8 bytes < 4 GB)
    auto load_row = [&](const msl_i4v& rows, const msl_i4v& rows_hi, int k, unsigned c) {
        const msl_f2v t = k < KH ? msl_raw_buffer_load_f2(rows, (int)(8u * c), (int)(8u * (unsigned)k * (unsigned)job.npix), 2)
                                 : msl_raw_buffer_load_f2(rows_hi, (int)(8u * c), (int)(8u * (unsigned)(k - KH) * (unsigned)job.npix), 2);
        return make_float2(t.x, t.y);
    };
    long long tile = blockIdx.x;
    float2 nx[PF];
    msl_i4v rows, rows_hi; float* orow; unsigned c; bool live;
    if (tile < n_tiles) {
        column(tile, rows, rows_hi, orow, c, live);
#pragma unroll
        for (int k = 0; k < PF; ++k) nx[k] = load_row(rows, rows_hi, k, c);
    }
    for (; tile < n_tiles; tile += step) {
        float2 v[T];
#pragma unroll
        for (int k = 0; k < PF; ++k) v[k] = nx[k];
#pragma unroll
        for (int k = PF; k < T; ++k) v[k] = load_row(rows, rows_hi, k, c);
        float* const out_rows = orow;
        const unsigned my_c = c;
        if (tile + step < n_tiles) {
            column(tile + step, rows, rows_hi, orow, c, live);
#pragma unroll
            for (int k = 0; k < PF; ++k) nx[k] = load_row(rows, rows_hi, k, c);
        }
        __builtin_amdgcn_sched_barrier(0);
        // any constant may be subtracted (only bin 0 sees it, and bin 0 is zeroed): the first sample keeps the numbers small
        const float2 ref = v[0];
#pragma unroll
        for (int k = 0; k < T; ++k) v[k] = make_float2(v[k].x - ref.x, v[k].y - ref.y);
        dif<T, 1, false, true>(v);
        static_for<0, T>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int F = dif_out_index(I, T);                 // frequency held by register I
            constexpr int KS = (F + half) % T;                     // np.fft.fftshift
            const float val = (F == 0) ? 0.f : fmaf(v[I].x, v[I].x, v[I].y * v[I].y);
            __builtin_nontemporal_store(val, reinterpret_cast<float*>(reinterpret_cast<char*>(out_rows + (long long)KS * job.npix) + 4u * my_c));
        });
    }
}

// ---- smooth frame counts above 128: the same transform split over the L waves of a workgroup, T = L x TP ----
// Wave q holds the samples q TP + K of 64 pixels (lane = pixel: loads and stores keep their 512- / 256-byte runs) and the first
// decimation-in-frequency level is a radix-L butterfly ACROSS the waves through the LDS:
//     y_q[K] = ( sum_j x[j TP + K] W_L^{j q} ) W_T^{K q},      X[L f + q] = FFT_TP(y_q)[f]
// -- coefficients and twiddles are wave-uniform (q is), read from a T-entry table of W_T^n -- after which every lane runs the
// TP-point register network on its own.  The exchange goes in chunks of CH samples through two alternating buffers (L x CH x 64
// complex each), one barrier per chunk: a wave passes barrier n + 1 only after it has read chunk n, so chunk n + 2 may overwrite
// it.  L = 2, 3, 4 run one wave per SIMD with 512 registers per lane; L = 5, 6 two, with 256 (and TP <= 100).
// HB = 2 (T above 512: L = 6, 8): a wave holds TWO blocks, lanes 0..31 block 2w and lanes 32..63 block 2w + 1 of 32 pixels (256- /
// 128-byte runs) -- eight blocks of 128 samples on four waves with the whole register file each; q, the coefficients and the row
// addresses are then per-lane values, the second block's rows are reached through the lane offset (host: (TP + 65) npix 8 < 4 GB).
__host__ __device__ constexpr int tsplit_chunk(int L, int HB = 1) { return 2 * L * 32 * (64 / HB) * 8 + TDIR_MAX * L * 8 <= 150 * 1024 ? 32 : 16; }
__host__ __device__ constexpr int tsplit_prefetch(int TP, int W, int HB = 1) {          // W: waves of the workgroup
    const int room = ((W <= 4 ? 512 : 256) - (HB == 1 ? 224 : 256) - 2 * TP) / 2;       // (per-lane coefficients and addresses at HB = 2)
    return room < 0 ? 0 : (room > TP ? TP : room);
}
__host__ __device__ constexpr size_t tsplit_lds_bytes(int TP, int L, int HB = 1) {
    return ((size_t)TP * L + (size_t)2 * L * tsplit_chunk(L, HB) * (64 / HB)) * 8;
}

template <int TP, int L, int HB = 1>
__global__ void __launch_bounds__(64 * L / HB) time_split_kernel(TimeJob job) {
    static_assert(HB == 1 || (HB == 2 && L % 2 == 0), "one block per wave, or two on its halves");
    constexpr int T = TP * L, half = T / 2, KH = (TP + 1) / 2, CH = tsplit_chunk(L, HB), W = L / HB, PW = 64 / HB;
    // Prefetch of the next tile: PF rows before the transform starts -- what fits beside the 2 TP data registers -- and the others
    // block by block: the register network's first level leaves R1 independent blocks of M1 samples; as soon as a block is
    // transformed and stored its registers take the next LATE rows.  All TP rows of the next tile are in flight or landed when the
    // tile ends, and the loads are spread over the whole of it.
    // (Rows that still do not fit -- two waves per SIMD leave few spare registers -- are fetched at the top of the tile.)
    constexpr int R1 = fft_radix(TP), M1 = TP / R1;
    constexpr int PF = tsplit_prefetch(TP, W, HB);
    constexpr int LATE_ALL = (R1 - 1) * M1 < TP - PF ? (R1 - 1) * M1 : TP - PF;      // rows fetched behind finished blocks
    constexpr int NPRE = PF + LATE_ALL;                                                // rows of the next tile in flight at its start
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* twl = reinterpret_cast<float2*>(smem_raw);             // W_T^n, n < T
    float2* xbuf = twl + T;                                        // two exchange buffers [j][k][lane]
    const int tid = threadIdx.x, lane = tid & 63, px = lane % PW, hb = lane / PW;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = HB == 1 ? wave : wave * HB + hb;
    for (int i = tid; i < T; i += 64 * W) twl[i] = job.tw[i];
    __syncthreads();
    float cr[L], ci[L];                                            // W_L^{j q} = W_T^{(j q mod L) TP}
#pragma unroll
    for (int j = 1; j < L; ++j) {
        const float2 w = twl[((j * q) % L) * TP];
        if constexpr (HB == 1) {
            cr[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.x)));
            ci[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.y)));
        } else {
            cr[j] = w.x; ci[j] = w.y;
        }
    }
    const int tiles_per_image = (job.npix + PW - 1) / PW;
    const long long n_tiles = (long long)tiles_per_image * job.n_images;
    const long long step = gridDim.x;
    // (ragged last tile: the surplus lanes repeat the image's last pixel, as in time_direct_kernel)
    auto column = [&](long long t, msl_i4v& rows0, msl_i4v& rows, msl_i4v& rows_hi, float*& orow, unsigned& c) {
        const int p = __builtin_amdgcn_readfirstlane((int)(t / tiles_per_image));
        const int c0 = __builtin_amdgcn_readfirstlane((int)(t % tiles_per_image) * PW);
        c = c0 + px < job.npix ? (unsigned)(c0 + px) : (unsigned)(job.npix - 1);
        const float2* img = job.in + (long long)p * job.image_stride;
        rows0 = make_raw_rsrc(img);
        rows = make_raw_rsrc(img + (long long)(wave * HB * TP) * job.npix);
        rows_hi = make_raw_rsrc(img + (long long)(wave * HB * TP + KH) * job.npix);
        orow = job.out + (long long)p * job.image_stride;
    };
    // (row offsets, LDS addresses beyond the 64 KB an instruction's offset field reaches: all the same for every tile, and the
    // compiler would keep hundreds of them in registers across the loop -- and spill them; `hide` makes a value look new)
    auto hide_s = [](int x) { asm volatile("" : "+s"(x)); return x; };
    auto hide_v = [](int x) { asm volatile("" : "+v"(x)); return x; };
    auto load_row = [&](const msl_i4v& rows, const msl_i4v& rows_hi, int k, unsigned c, int npix_now) {
        // (HB = 2: the upper half-wave's block lies TP rows further on)
        const unsigned vo = HB == 1 ? 8u * c : 8u * c + (unsigned)hb * (8u * (unsigned)TP * (unsigned)npix_now);
        const msl_f2v t = k < KH ? msl_raw_buffer_load_f2(rows, (int)vo, (int)(8u * (unsigned)k * (unsigned)npix_now), 2)
                                 : msl_raw_buffer_load_f2(rows_hi, (int)vo, (int)(8u * (unsigned)(k - KH) * (unsigned)npix_now), 2);
        return make_float2(t.x, t.y);
    };
    long long tile = blockIdx.x;
    float2 nx[NPRE > 0 ? NPRE : 1];
    float2 nref = make_float2(0.f, 0.f);
    msl_i4v rows0, rows, rows_hi; float* orow; unsigned c;
    if (tile < n_tiles) {
        column(tile, rows0, rows, rows_hi, orow, c);
        { const msl_f2v t = msl_raw_buffer_load_f2(rows0, (int)(8u * c), 0, 0); nref = make_float2(t.x, t.y); }
#pragma unroll
        for (int k = 0; k < NPRE; ++k) nx[k] = load_row(rows, rows_hi, k, c, job.npix);
    }
    int par = 0;
    for (; tile < n_tiles; tile += step) {
        float2 v[TP];
        // the line's first sample, subtracted from all of it (time_cz_kernel's note): every wave fetches it, wave 0's fetch pays
        const float2 ref = nref;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) v[k] = nx[k];
        int npix_now = hide_s(job.npix);
#pragma unroll
        for (int k = NPRE; k < TP; ++k) v[k] = load_row(rows, rows_hi, k, c, npix_now);
        float* const out_rows = orow;
        const unsigned my_c = c;
        // (a workgroup's last tile prefetches itself again: unconditional loads -- a branch would keep the old values alive as
        // the other arm of the merge, a second copy of the line)
        column(tile + step < n_tiles ? tile + step : tile, rows0, rows, rows_hi, orow, c);
        { const msl_f2v t = msl_raw_buffer_load_f2(rows0, (int)(8u * c), 0, 0); nref = make_float2(t.x, t.y); }
#pragma unroll
        for (int k = 0; k < PF; ++k) nx[k] = load_row(rows, rows_hi, k, c, npix_now);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, (TP + CH - 1) / CH>([&](auto cc) {
            constexpr int cb = decltype(cc)::value * CH;
            float2* buf = xbuf + par * (L * CH * PW) + hide_v(px);
            const int qx = HB == 1 ? q : hide_v(q);            // (per-lane q: its multiples are loop-invariant vector values, see hide_v)
            par ^= 1;
#pragma unroll
            for (int k = cb; k < cb + CH; ++k)
                if (k < TP) buf[(qx * CH + (k - cb)) * PW] = make_float2(v[k].x - ref.x, v[k].y - ref.y);
            lds_barrier();
            // the butterfly of my wave, four samples at a time, the next four already on their way from the LDS (one wave per
            // SIMD: nobody else would cover the round trip).  Even L use the structure of the coefficients:
            // sum_m W_L^{m q} (a_m + s a_{m + L/2}) with s = W_L^{(L/2) q} = +-1.
            constexpr int G = L <= 4 ? 4 : 2, KEND = cb + CH < TP ? cb + CH : TP;
            float2 A[G][L], W[G];
            auto fetch = [&](int k0, float2 (&a)[G][L], float2 (&w)[G]) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (k0 + g < KEND) {
#pragma unroll
                        for (int j = 0; j < L; ++j) a[g][j] = buf[(j * CH + (k0 + g - cb)) * PW];
                        w[g] = twl[(k0 + g) * qx];
                    }
            };
            fetch(cb, A, W);
#pragma unroll
            for (int k0 = cb; k0 < KEND; k0 += G) {
                float2 B[G][L], Wn[G];
                if (k0 + G < KEND) fetch(k0 + G, B, Wn);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (k0 + g < KEND) {
                        float2 acc;
                        if constexpr (L % 2 == 0) {
                            // W_L^{(m + L/2) q} = (-1)^q W_L^{m q}: fold the upper half onto the lower one first (real sign),
                            // then L/2 - 1 complex coefficients instead of L - 1
                            constexpr int HL = L / 2;
                            float2 b[HL];
#pragma unroll
                            for (int m = 0; m < HL; ++m)
                                b[m] = make_float2(fmaf(A[g][m + HL].x, cr[HL], A[g][m].x), fmaf(A[g][m + HL].y, cr[HL], A[g][m].y));
                            acc = b[0];
#pragma unroll
                            for (int m = 1; m < HL; ++m) {
                                acc.x = fmaf(b[m].x, cr[m], fmaf(-b[m].y, ci[m], acc.x));
                                acc.y = fmaf(b[m].x, ci[m], fmaf(b[m].y, cr[m], acc.y));
                            }
                        } else {
                            acc = A[g][0];
#pragma unroll
                            for (int j = 1; j < L; ++j) {
                                acc.x = fmaf(A[g][j].x, cr[j], fmaf(-A[g][j].y, ci[j], acc.x));
                                acc.y = fmaf(A[g][j].x, ci[j], fmaf(A[g][j].y, cr[j], acc.y));
                            }
                        }
                        v[k0 + g] = cmulf(acc, W[g]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    W[g] = Wn[g];
#pragma unroll
                    for (int j = 0; j < L; ++j) A[g][j] = B[g][j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (R1 == 4) dif4_level<TP, 1, false, 0, true>(v);
        else if constexpr (R1 == 2) dif2_level<TP, 1, false, 0, true>(v);
        else if constexpr (R1 == 5) dif5_level<TP, 1, false, 0, true>(v);
        else dif3_level<TP, 1, false, 0, true>(v);
        // (the row offsets below are the same for every tile: hidden from the compiler, which would otherwise keep all TP of them,
        // 64 bits each, in scalar registers across the loop and spill those through the vector file)
        static_for<0, R1>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            dif<M1, 1, false, true>(v + B * M1);
            npix_now = hide_s(npix_now);
            const int qs = HB == 1 ? q : hide_v(q);
            static_for<B * M1, (B + 1) * M1>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                constexpr int F = dif_out_index(I, TP);            // sub-frequency held by register I: bin L F + q
                int row = L * F + qs + half;                        // np.fft.fftshift (uniform when a wave holds one block)
                if (row >= T) row -= T;
                float val = fmaf(v[I].x, v[I].x, v[I].y * v[I].y);
                if (F == 0 && qs == 0) val = 0.f;
                if constexpr (HB == 1)
                    __builtin_nontemporal_store(val, reinterpret_cast<float*>(reinterpret_cast<char*>(out_rows + (long long)row * npix_now) + 4u * my_c));
                else
                    __builtin_nontemporal_store(val, out_rows + ((long long)row * npix_now + my_c));
            });
            if constexpr (B < R1 - 1 && PF + B * M1 < NPRE) {
                constexpr int K0 = PF + B * M1, K1 = K0 + M1 < NPRE ? K0 + M1 : NPRE;
#pragma unroll
                for (int k = K0; k < K1; ++k) nx[k] = load_row(rows, rows_hi, k, c, npix_now);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

}  // namespace msl
