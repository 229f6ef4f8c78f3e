// Register-resident four-step FFT passes for square power-of-two line lengths N = R*R
// (R = 32 -> 1024, R = 16 -> 256) on gfx950.  These are the roofline-judged kernels of the slice
// loop (reference Propagate, src/multislice/multislice.py:278-294):
//
//   row pass     per row (p,x):    [ifft_y] -> x t_z[x,:] -> [fft_y] -> [x Py]      one HBM read + write
//   column pass  per column (p,ky): fft_x -> x Px -> ifft_x                          one HBM read + write
//
// A line of N = R*R points lives in a group of R lanes, R registers per lane, element index
// n = reg*R + lane -- both before and after a transform (four-step: register FFT over `reg`,
// twiddle W_N^{lane*k1}, lane<->register transpose through LDS, register FFT).  So a whole
// ifft -> multiply -> fft chain never leaves registers except for the two transposes, and HBM is
// touched exactly once per pass with 8R-byte (row pass) or 128-byte (column pass) contiguous
// segments.  Index algebra is unit-tested on the host (tools/fft_regs_host.cpp).
#pragma once
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>
#include "fft_regs.h"
#include "kernel_util.h"
#include "rowt_pass.h"

namespace msl {

// Multiply v[j] (j = J0..R-1) by tab[j*R + ln] (conjugated when CONJ) in chunks of CH with a scheduling
// barrier between chunks, so the compiler cannot hoist all R table loads at once (register pressure).  Every chunk
// is one exposed LDS round trip: the transposing kernel, which has the registers, uses chunks of 16 (291 -> 288 us).
template <int R, int J0, bool CONJ, int STRIDE = R, int CH = 8, typename TabPtr>
__device__ __forceinline__ void mul_table(float2 (&v)[R], TabPtr tab, int ln) {
#pragma unroll
    for (int c = 0; c < R; c += CH) {
        float2 w[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) if (c + j >= J0) w[j] = tab[(c + j) * STRIDE + ln];
#pragma unroll
        for (int j = 0; j < CH; ++j) if (c + j >= J0) v[c + j] = CONJ ? cmulf_conj(v[c + j], w[j]) : cmulf(v[c + j], w[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- four-step transform of one line held by an R-lane group ------------------------------------
// tw: LDS table T[k1*R + lane] = exp(-2 pi i lane k1 / N) (forward); conjugated for INV.
// Transpose through a float scratch of R*(R+1) words, real and imaginary parts one after the other
// (row pass: halves the LDS footprint so more waves fit a CU).
template <int R, bool INV, int CH = 8>
__device__ __forceinline__ void fourstep_split(float2 (&v)[R], float* scratch, const float2* tw, int ln) {
    static_assert(64 % R == 0, "an R-lane group must lie inside one wave: the scratch is ordered per wave only");
    // decimation in time with the twiddles folded into the second transform's leaves (see fourstep_split_addtid) for 16 registers;
    // with 32 the kernels that use this form (row_pass_pf_kernel, col_pass_kernel: work buffers next to the line) spill on it and
    // keep the decimation-in-frequency network with its separate twiddle pass
    constexpr bool DIT = R <= 16;
    constexpr int LCH = (CH / 2 < dit_leaf_count(R)) ? (CH / 2 > 0 ? CH / 2 : 1) : dit_leaf_count(R);
    if constexpr (DIT) { fft_regs_dit<R, INV, 0, 1>(v, v); pin_all(v); }
    else { fft_regs<R, INV>(v); mul_table<R, 1, INV, R, CH>(v, tw, ln); }
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) scratch[k1 * (R + 1) + ln] = v[k1].x;
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < R; ++n2) v[n2].x = scratch[ln * (R + 1) + n2];
    wave_lds_fence();
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) scratch[k1 * (R + 1) + ln] = v[k1].y;
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < R; ++n2) v[n2].y = scratch[ln * (R + 1) + n2];
    wave_lds_fence();
    if constexpr (DIT) {
        dit_leaf_chunks<R, INV, INV ? 2 : 1, R, LCH, 0>(v, v, tw, ln);
        dit_upper<R, INV, 1>(v);
        pin_all(v);
    } else {
        fft_regs<R, INV>(v);
    }
}

// Exchange with ds_write_addtid_b32 stores and 16-byte reads: the store address is M0 + offset + 4 * lane, so it needs no address
// register and runs at twice the rate of ds_write_b32 (128 B/clk: MI355X_MICROARCH.md, LDS).  The 64 / R line groups of a wave
// share one scratch of R rows x 68 floats: row k1 holds the k1-th register of all 64 lanes (group g at columns [g R, (g+1) R)),
// and lane (g, l) reads back row l, columns g R + n2, as R/4 ds_read_b128 (row pitch 68: 16-byte aligned, conflict-free).
// wave_scratch: LDS address (bytes) of the wave's scratch, wave-uniform; M0 is not used by anything else in these kernels, which
// tests/test_abi_and_host.py checks on the ISA (every write of M0 in the library is one of these s_mov).  Declaring an "m0" clobber
// draws clang's reserved-register diagnostic ("may lead to undefined behaviour"); one self-contained asm block per 16 stores, with
// its own s_mov, cost 0.7 % at 1024^2 and 1.8 % at 512^2 -- so the statements stay as they are.
template <int R, bool INV, int CH = 8>
__device__ __forceinline__ void fourstep_split_addtid(float2 (&v)[R], const float* scratch_base, unsigned wave_scratch, const float2* tw, int ln, int lane64) {
    // decimation-in-time form (round 4, see rowt_pass.h): register FFT, exchange, and the twiddles as weights of the second
    // register FFT's leaf level -- the table is symmetric, the weight of element n2 in lane k1 is T[n2 R + k1] -- 388 + 484
    // instructions per 32 registers where the decimation-in-frequency network with a separate twiddle pass took 2 x 430 + 128
    constexpr int LCH = (CH / 2 < dit_leaf_count(R)) ? (CH / 2 > 0 ? CH / 2 : 1) : dit_leaf_count(R);
    fft_regs_dit<R, INV, 0, 1>(v, v);
    pin_all(v);
    exchange_addtid<R>(v, scratch_base, wave_scratch, ln, lane64);
    dit_leaf_chunks<R, INV, INV ? 2 : 1, R, LCH, 0>(v, v, tw, ln);
    dit_upper<R, INV, 1>(v);
    pin_all(v);
}

// same with a complex scratch of R*(R+1) float2 (column pass: the tile is in LDS anyway)
template <int R, bool INV, int CH = 8>
__device__ __forceinline__ void fourstep_c64(float2 (&v)[R], float2* scratch, const float2* tw, int ln) {
    static_assert(64 % R == 0, "an R-lane group must lie inside one wave: the scratch is ordered per wave only");
    constexpr bool DIT = R <= 16;                       // see fourstep_split
    constexpr int LCH = (CH / 2 < dit_leaf_count(R)) ? (CH / 2 > 0 ? CH / 2 : 1) : dit_leaf_count(R);
    if constexpr (DIT) { fft_regs_dit<R, INV, 0, 1>(v, v); pin_all(v); }
    else { fft_regs<R, INV>(v); mul_table<R, 1, INV, R, CH>(v, tw, ln); }
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) scratch[k1 * (R + 1) + ln] = v[k1];
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < R; ++n2) v[n2] = scratch[ln * (R + 1) + n2];
    wave_lds_fence();
    if constexpr (DIT) {
        dit_leaf_chunks<R, INV, INV ? 2 : 1, R, LCH, 0>(v, v, tw, ln);
        dit_upper<R, INV, 1>(v);
        pin_all(v);
    } else {
        fft_regs<R, INV>(v);
    }
}

struct RowJob {
    float2* psi;            // (P, nx, pitch) working waves, rows contiguous
    const float2* trans;    // t_z (nx, ny) of this slice, or null
    const float2* py;       // (ny) Fresnel factor along y with 1/ny folded in, or null
    const float2* tw;       // (N) four-step twiddles T[k1*R + n2]
    long long image_stride; // elements between probes
    int pitch;              // elements between rows
    int nx;                 // rows per image
    int n_images;           // P
    int do_ifft, do_fft;
    int pchunk;             // probes per work item of the pipelined row kernel (t_z row reuse in registers)
    // frame batching: images [g*t_group, (g+1)*t_group) belong to frame g of the batch, whose transmission stack starts
    // t_stride elements after the previous frame's (t_group == 0: one frame, every image uses `trans`)
    int t_group;
    unsigned t_magic;       // floor(2^32 / t_group) + 1
    long long t_stride;
};

// Row pass (two-pass loop, stand-alone FFTs), software-pipelined.  Workgroup = 256 threads = 256/R lines per iteration.
// A work item is (x-group, chunk of `pchunk` probes): the t_z rows
// of the x-group are loaded once into registers and reused for every probe of the chunk, so t_z costs
// nx*ny*8*P/pchunk bytes per launch instead of depending on L2 luck.  The next line's wave data (HBM) is
// in flight while the current one is transformed; Py comes from LDS.  ~230 VGPRs -> 2 waves per SIMD,
// every wave keeps 8R^2 bytes of HBM loads outstanding all the time.
template <int R>
__global__ void __launch_bounds__(256, 2) row_pass_pf_kernel(RowJob job) {
    constexpr int N = R * R;
    constexpr int G = 256 / R;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);                 // N
    float2* pyl = tw + N;                                             // N
    float* scratch_all = reinterpret_cast<float*>(pyl + N);           // G * R*(R+1) floats
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += 256) { tw[i] = job.tw[i]; pyl[i] = job.py ? job.py[i] : make_float2(1.f, 0.f); }
    __syncthreads();
    const int grp = tid / R, ln = tid % R;
    float* scratch = scratch_all + grp * (R * (R + 1));
    const int xgroups = job.nx / G;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const long long n_items = (long long)xgroups * pchunks;
    // cursor over (item, k): item = xg*pchunks + pc, probe p = pc*PC + k
    long long item = blockIdx.x;
    int k = 0;
    auto chunk_len = [&](long long it) { const int pc = (int)(it % pchunks); return min(PC, job.n_images - pc * PC); };
    auto row_of = [&](long long it, int kk) {
        const int xg = (int)(it / pchunks), pc = (int)(it % pchunks);
        const int x = xg * G + grp;
        return job.psi + (long long)(pc * PC + kk) * job.image_stride + (long long)x * job.pitch;
    };
    float2 vn[R];
    if (item < n_items) {
        const float2* r = row_of(item, 0);
#pragma unroll
        for (int j = 0; j < R; ++j) vn[j] = r[j * R + ln];
    }
    float2 tv[R];
    while (item < n_items) {
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = vn[j];
        float2* cur_row = row_of(item, k);
        if (k == 0 && job.trans) {          // new x-group: its transmission rows first (L2), before the next HBM loads
            const int x = (int)(item / pchunks) * G + grp;
            const float2* trow = job.trans + frame_off(job, (int)(item % pchunks) * PC) + (long long)x * N;
#pragma unroll
            for (int j = 0; j < R; ++j) tv[j] = trow[j * R + ln];
        }
        // advance the cursor and put the next line's loads in flight
        long long nitem = item;
        int nk = k + 1;
        if (nk >= chunk_len(item)) { nitem = item + gridDim.x; nk = 0; }
        if (nitem < n_items) {
            const float2* r = row_of(nitem, nk);
#pragma unroll
            for (int j = 0; j < R; ++j) vn[j] = r[j * R + ln];
        }
        if (job.do_ifft) fourstep_split<R, true>(v, scratch, tw, ln);
        if (job.trans) {
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = cmulf(v[j], tv[j]);
        }
        if (job.do_fft) {
            fourstep_split<R, false>(v, scratch, tw, ln);
            if (job.py) mul_table<R, 0, false>(v, pyl, ln);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) cur_row[j * R + ln] = v[j];
        item = nitem; k = nk;
    }
}

struct ColJob {
    const float2* in;       // (P, nx, pitch)
    float2* out;            // same buffer for the slice loop; the (P,T,nx,ny) result for the epilogue
    const float2* px;       // (nx) Fresnel factor along x with 1/nx folded in (mode 0)
    const float2* tw;       // (N) four-step twiddles
    long long in_image_stride, out_image_stride;
    int in_pitch, out_pitch;
    int ny;                 // columns per image
    int n_images;
    int flags;              // COL_FWD | COL_MULPX | COL_INV | COL_SHIFT (fftshifted scatter of both axes into `out`)
    float scale;            // applied at the store (1/(nx ny) of a stand-alone inverse transform)
    float sigma;            // COL_POTENTIAL: t = exp(i sigma V)
    float* out_real;        // COL_POTENTIAL: optional V (same pitch / image stride as `out`)
    int slice_mod;          // COL_TPOT: slice number of image p = p % slice_mod (several frames' stacks in one launch); 0: p itself
    int tparity;            // COL_TPOT: images whose slice-number parity differs from this are stored transposed ...
    float2* out_t;          // ... into this separate (n_images, ny, nx) buffer (in-place transposition would race)
    // COL_SHIFT only: k-window in fftshifted coordinates.  Columns [win_c0, win_c0 + win_nc) are transformed (win_nc == 0:
    // all ny), rows [win_x0, win_x0 + win_nx) are stored, both rebased to 0.  win_c0, win_nc and ny/2 are multiples of 16.
    int win_c0, win_nc, win_x0, win_nx;
    // frame batching (COL_SHIFT epilogue): image p is probe p % out_group of frame p / out_group and goes to
    // out + (p % out_group) * out_image_stride + (p / out_group) * out_group_stride   (out_group == 0: p * out_image_stride)
    int out_group;
    long long out_group_stride;
};
// COL_POTENTIAL: epilogue of the potential build, V = Re(x)*scale, out = exp(i sigma V)  (potentials.py:336-342, multislice.py:282)
// COL_TPOT (with COL_POTENTIAL): every second slice's t is stored transposed, (ny, nx), for the one-pass slice loop
// COL_INTENSITY: TACAW epilogue -- DC bin zeroed (== time-mean subtraction), |x|^2 as float into out_real with the
//                line index fftshifted (tacaw_data.py:94-104); the "columns" are pixels, the lines run along time
enum { COL_FWD = 1, COL_MULPX = 2, COL_INV = 4, COL_SHIFT = 8, COL_POTENTIAL = 16, COL_TPOT = 32, COL_INTENSITY = 64 };

// Column pass.  Workgroup = COLS*R threads owns a tile of COLS = 16 neighbouring columns (128-byte row
// segments in HBM) x N rows: staged into LDS column-major, one R-lane group per column, results
// staged back and stored as 128-byte segments.  Persistent over tiles with the next tile's loads in
// flight in registers while the current one is transformed.
// COLS = 32 (TACAW time transform only: no window, no shift): 256-byte row segments in, 128-byte segments of the float output.
// HERM (potential build): the input is Hermitian along the column, in[N - x] = conj(in[x]) -- the spectrum of a real image after
// its row pass -- and only rows 0 .. N/2 exist: a tile reads half the rows and mirrors them while staging into the LDS.
template <int R, int COLS = 16, bool HERM = false>
__global__ void __launch_bounds__(COLS * R) col_pass_kernel(ColJob job) {
    constexpr int N = R * R;
    constexpr int NT = COLS * R;                      // threads
    constexpr int CS = R * (R + 1) + 1;               // LDS column stride in float2 (odd*8 B: conflict-free staging)
    constexpr int QN = COLS / 2;                      // threads (16 B each) per row segment
    constexpr int ROWS_PER_IT = NT / QN;
    constexpr int NIT = N / ROWS_PER_IT;              // float4 per thread per tile
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);                 // N
    float2* px = tw + N;                                              // N
    float2* cols = px + N;                                            // COLS * CS
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += NT) { tw[i] = job.tw[i]; px[i] = job.px ? job.px[i] : make_float2(1.f, 0.f); }
    const int grp = tid / R, ln = tid % R;            // column handled in the transform phase
    const int q = tid % QN, r0 = tid / QN;            // staging role: column pair q, row r0 + ROWS_PER_IT*i
    float2* mycol = cols + grp * CS;
    const bool windowed = (job.flags & COL_SHIFT) && job.win_nc > 0;
    const int tiles_per_image = (windowed ? job.win_nc : job.ny) / COLS;
    const long long n_tiles = (long long)tiles_per_image * job.n_images;
    // first (unshifted) column of tile t of an image
    auto tile_col = [&](long long t) {
        int c = (int)t * COLS;
        if (windowed) { c += job.win_c0 + job.ny / 2; if (c >= job.ny) c -= job.ny; }
        return c;
    };
    constexpr int NLD = HERM ? NIT / 2 : NIT;         // rows r0 + ROWS_PER_IT * i < N/2; row N/2 is the extra load of the r0 == 0 threads
    float4 stage[NIT];
    float4 stage_ny = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_tile = [&](long long t) {
        const long long p = t / tiles_per_image, c0 = tile_col(t % tiles_per_image);
        const float2* src = job.in + p * job.in_image_stride + c0 + 2 * q;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            stage[i] = *reinterpret_cast<const float4*>(src + (long long)(r0 + ROWS_PER_IT * i) * job.in_pitch);
        if constexpr (HERM) { if (r0 == 0) stage_ny = *reinterpret_cast<const float4*>(src + (long long)(N / 2) * job.in_pitch); }
    };
    long long tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    __syncthreads();
    for (; tile < n_tiles; tile += gridDim.x) {
        // ---- registers -> LDS, column-major
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int x = r0 + ROWS_PER_IT * i;
            cols[(2 * q) * CS + x] = make_float2(stage[i].x, stage[i].y);
            cols[(2 * q + 1) * CS + x] = make_float2(stage[i].z, stage[i].w);
            if constexpr (HERM) {
                if (x > 0) {                                  // rows N/2 + 1 .. N - 1 from their mirror images
                    cols[(2 * q) * CS + N - x] = make_float2(stage[i].x, -stage[i].y);
                    cols[(2 * q + 1) * CS + N - x] = make_float2(stage[i].z, -stage[i].w);
                }
            }
        }
        if constexpr (HERM) {
            if (r0 == 0) {
                cols[(2 * q) * CS + N / 2] = make_float2(stage_ny.x, stage_ny.y);
                cols[(2 * q + 1) * CS + N / 2] = make_float2(stage_ny.z, stage_ny.w);
            }
        }
        lds_barrier();
        // ---- next tile's loads go out now and fly during the transform
        const long long nxt = tile + gridDim.x;
        if (nxt < n_tiles) load_tile(nxt);
        // ---- transform my column
        bool tstore = false;
        {
            float2 v[R];
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = mycol[j * R + ln];
            if (job.flags & COL_INTENSITY) {
                // TACAW: the line's first sample instead of its time mean -- any constant only changes the DC bin, which is zeroed
                // below -- so that a pixel whose mean dwarfs its thermal part is transformed at the size of the result
                const float2 ref = mycol[0];
#pragma unroll
                for (int j = 0; j < R; ++j) v[j] = make_float2(v[j].x - ref.x, v[j].y - ref.y);
            }
            wave_lds_fence();
            if (job.flags & COL_FWD) fourstep_c64<R, false>(v, mycol, tw, ln);
            if (job.flags & COL_MULPX) {
                mul_table<R, 0, false>(v, px, ln);
            }
            if (job.flags & COL_INV) fourstep_c64<R, true>(v, mycol, tw, ln);
            wave_lds_fence();
            const long long pimg = tile / tiles_per_image;
            tstore = (job.flags & COL_TPOT) && ((((int)(job.slice_mod > 0 ? pimg % job.slice_mod : pimg)) & 1) != job.tparity);
            if (tstore) {
                // transposed transmission slice: column (fixed y) is a contiguous line of the (ny, nx) image
                const int y = tile_col(tile % tiles_per_image) + grp;
                float2* trow = job.out_t + pimg * job.out_image_stride + (long long)y * N;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    float sn, cs;
                    sincosf(job.sigma * (v[j].x * job.scale), &sn, &cs);
                    trow[j * R + ln] = make_float2(cs, sn);
                }
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) mycol[j * R + ln] = v[j];
            }
        }
        lds_barrier();
        // ---- LDS -> registers -> HBM (128-byte segments)
        if (!tstore) {
            const long long p = tile / tiles_per_image;
            const int c0 = tile_col(tile % tiles_per_image);
            int cshift = c0, xshift = 0;
            if (job.flags & COL_SHIFT) { cshift = (c0 + job.ny / 2) % job.ny; xshift = N / 2; }
            if (windowed) cshift -= job.win_c0;
            const long long obase = job.out_group > 0 ? (p % job.out_group) * job.out_image_stride + (p / job.out_group) * job.out_group_stride
                                                      : p * job.out_image_stride;
            float2* dst = job.out + obase + cshift + 2 * q;
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int x = r0 + ROWS_PER_IT * i;
                float2 a = cols[(2 * q) * CS + x], b = cols[(2 * q + 1) * CS + x];
                int xo = x + xshift;
                if (xo >= N) xo -= N;
                if (windowed) {
                    xo -= job.win_x0;
                    if (xo < 0 || xo >= job.win_nx) continue;
                }
                if (job.flags & COL_INTENSITY) {
                    const float ia = (x == 0) ? 0.f : fmaf(a.x, a.x, a.y * a.y), ib = (x == 0) ? 0.f : fmaf(b.x, b.x, b.y * b.y);
                    int xs = x + N / 2;
                    if (xs >= N) xs -= N;
                    *reinterpret_cast<float2*>(job.out_real + p * job.out_image_stride + c0 + 2 * q +
                                               (long long)xs * job.out_pitch) = make_float2(ia, ib);
                } else if (job.flags & COL_POTENTIAL) {
                    const float va = a.x * job.scale, vb = b.x * job.scale;
                    if (job.out_real)
                        *reinterpret_cast<float2*>(job.out_real + p * job.out_image_stride + cshift + 2 * q +
                                                   (long long)xo * job.out_pitch) = make_float2(va, vb);
                    float sa, ca, sb, cb;
                    sincosf(job.sigma * va, &sa, &ca);
                    sincosf(job.sigma * vb, &sb, &cb);
                    *reinterpret_cast<float4*>(dst + (long long)xo * job.out_pitch) = make_float4(ca, sa, cb, sb);
                } else {
                    *reinterpret_cast<float4*>(dst + (long long)xo * job.out_pitch) =
                        make_float4(a.x * job.scale, a.y * job.scale, b.x * job.scale, b.y * job.scale);
                }
            }
        }
        lds_barrier();                          // LDS is free for the next tile's staging from here on
    }
}

// =================================================================================================
// One-pass-per-slice kernels.  The Fresnel propagator is separable, P = Px(kx) Py(ky)
// (multislice.py:273-275), so propagation factors into two commuting 1-D operators
//     A_y = ifft_y Py fft_y   (row-local)        A_x = ifft_x Px fft_x   (column-local)
// and the slice recursion psi <- A_x A_y (t_z psi) can be regrouped into passes that alternate direction,
//     row:  A_y . t_k . A_y        column:  A_x . t_{k+1} . A_x        row:  A_y . t_{k+2} . A_y   ...
// each pass finishing the propagation of the previous slice along its axis, applying the next
// transmission function (pointwise, so local to any tile) and starting the next propagation.  One HBM
// read + write of psi per slice instead of two: 16 B/pixel/slice-step.  Same arithmetic as the
// reference up to fp32 rounding order.
// =================================================================================================

struct Row2Job {
    float2* psi;
    const float2* trans;    // t_k (nx, ny)
    const float2* py;       // (ny), 1/ny folded in
    const float2* tw;
    long long image_stride;
    int pitch, nx, n_images, flags, pchunk;
    int t_group;            // frame batching, see RowJob
    unsigned t_magic;
    long long t_stride;
};

template <int R>
__global__ void __launch_bounds__(256, 2) row_pass2_kernel(Row2Job job) {
    constexpr int N = R * R;
    constexpr int G = 256 / R;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);
    float2* pyl = tw + N;
    float* scratch_all = reinterpret_cast<float*>(pyl + N);
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += 256) { tw[i] = job.tw[i]; pyl[i] = job.py[i]; }
    __syncthreads();
    const int grp = tid / R, ln = tid % R;
    float* scratch = scratch_all + grp * (R * (R + 1));
    const int xgroups = job.nx / G;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const long long n_items = (long long)xgroups * pchunks;
    long long item = blockIdx.x;
    int k = 0;
    auto chunk_len = [&](long long it) { const int pc = (int)(it % pchunks); return min(PC, job.n_images - pc * PC); };
    auto row_of = [&](long long it, int kk) {
        const int xg = (int)(it / pchunks), pc = (int)(it % pchunks);
        const int x = xg * G + grp;
        return job.psi + (long long)(pc * PC + kk) * job.image_stride + (long long)x * job.pitch;
    };
    float2 vn[R];
    if (item < n_items) {
        const float2* r = row_of(item, 0);
#pragma unroll
        for (int j = 0; j < R; ++j) vn[j] = r[j * R + ln];
    }
    float2 tv[R];
    while (item < n_items) {
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = vn[j];
        float2* cur_row = row_of(item, k);
        if (k == 0) {
            const int x = (int)(item / pchunks) * G + grp;
            const float2* trow = job.trans + frame_off(job, (int)(item % pchunks) * PC) + (long long)x * N;
#pragma unroll
            for (int j = 0; j < R; ++j) tv[j] = trow[j * R + ln];
        }
        long long nitem = item;
        int nk = k + 1;
        if (nk >= chunk_len(item)) { nitem = item + gridDim.x; nk = 0; }
        if (nitem < n_items) {
            const float2* r = row_of(nitem, nk);
#pragma unroll
            for (int j = 0; j < R; ++j) vn[j] = r[j * R + ln];
        }
        if (job.flags & P2_PRE_A) {
            fourstep_split<R, false>(v, scratch, tw, ln);
            mul_table<R, 0, false>(v, pyl, ln);
            fourstep_split<R, true>(v, scratch, tw, ln);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = cmulf(v[j], tv[j]);
        if (job.flags & P2_POST_A) {
            fourstep_split<R, false>(v, scratch, tw, ln);
            mul_table<R, 0, false>(v, pyl, ln);
            fourstep_split<R, true>(v, scratch, tw, ln);
        }
        if (job.flags & P2_POST_F) fourstep_split<R, false>(v, scratch, tw, ln);
#pragma unroll
        for (int j = 0; j < R; ++j) cur_row[j * R + ln] = v[j];
        item = nitem; k = nk;
    }
}

// ---- lines of ANY length N <= R^2/2: propagation as a zero-padded cyclic convolution on the register FFTs -------------------------
// The reference's grids are n = int(L/sampling) + 1 points (potentials.py:123-125; its own probe test uses 501 x 491,
// src/unittests/00_probe.py:7-8), almost never a power of two.  The generic LDS Stockham kernel (fft_generic.h) serves them at
// 0.1 of the roofline: every radix stage is a round trip of the whole tile through the LDS between two barriers.  Here a line of
// N points is embedded, zero-padded, in the M = R^2 point register layout of the power-of-two kernels (element n = reg R + lane).
// Inside the slice loop the N-point spectrum is never needed: A = ifft_N . P . fft_N is a circular convolution of length N with
// the fixed kernel a = ifft_N(P), and on the zero-padded line that is ONE cyclic convolution of length M >= 2N - 1 with
// q[j] = a[j], q[M - j] = a[N - j] (0 < j < N):  A x = IFFT_M(FFT_M(pad x) . Q)[0:N],  Q = FFT_M(q) / M (built in float64 on the
// host; symmetric, Q[M - k] = Q[k], so its first half is stored).  One pass A . t_k . A is four register FFTs of length M and
// three table products per line, for ANY length: a 501 x 491 grid runs at the speed of a 1024 x 1024 one instead of 3.5 x slower.
// (Round 2's first form evaluated every N-point DFT by Bluestein's chirp-z -- eight FFTs and eleven products per pass; the
// chirp-z tables live on in the potential's inverse transform, ifftTB_kernel, which does need the spectrum.)
// Everything else -- prefetch of the next line, t_k in registers across a chunk of probes, 16-line tile, 128-byte transposed
// segments -- is rowT_pass_kernel's.  The number of lines need not be a multiple of 16: the last tile re-reads the last line and
// its surplus columns land in the row padding of the output (pitch >= n_lines rounded up to 16), which no pass reads.
template <int R>
__global__ void __launch_bounds__(16 * R, (R == 32) ? 2 : 4) rowTB_pass_kernel(RowTJob job) {
    constexpr int M = R * R, H = R / 2, NH = M / 2, LINES = 16, NT = LINES * R, TCH = 8;
    constexpr int CS = R * (R + 1) + 2;               // even (16-byte aligned rows for the wide exchange), conflict-free staging
    constexpr int TPS = LINES / 2, POS_PER_IT = NT / TPS, NIT = NH / POS_PER_IT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // M: four-step twiddles
    float2* bf = tw + M;                                      // NH + 2: filter Q, first half
    float2* tile = bf + NH + 2;                               // LINES * CS
    const int tid = threadIdx.x;
    const int N = job.n_line;
    for (int i = tid; i < M; i += NT) tw[i] = job.tw[i];
    for (int i = tid; i <= NH; i += NT) bf[i] = job.bf[i];
    __syncthreads();
    const int grp = tid / R, ln = tid % R;
    const int q = tid % TPS, r0 = tid / TPS;
    float2* myrow = tile + grp * CS;
    // exchange scratch of the wave (ds_write_addtid_b32 stores): starts at the first tile row of the wave's 64 / R lines
    const float* wscr = reinterpret_cast<const float*>(tile + (grp - grp % (64 / R)) * CS);
    const unsigned wscr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(tile + (grp - grp % (64 / R)) * CS));
    const float2* fa = bf + ln;                               // Bf[j R + ln],                      j <  R/2
    const float2* fb = bf - ln;                               // Bf[M - (j R + ln)] = bf[(R - j) R - ln],  j >= R/2
    auto mul_filter = [&](float2 (&vv)[R]) {
#pragma unroll
        for (int c = 0; c < R; c += TCH) {
            float2 w[TCH];
#pragma unroll
            for (int j = 0; j < TCH; ++j) w[j] = (c + j < H) ? fa[(c + j) * R] : fb[(R - (c + j)) * R];
#pragma unroll
            for (int j = 0; j < TCH; ++j) vv[c + j] = cmulf(vv[c + j], w[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int lblocks = (job.n_lines + LINES - 1) / LINES;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const int n_items = lblocks * pchunks;
    const int step_lb = (int)gridDim.x / pchunks, step_pc = (int)gridDim.x % pchunks;
    auto line_ptr = [&](int lbb, int pcc, int kk) {
        const int L = min(lbb * LINES + grp, job.n_lines - 1);
        return job.in + (long long)(pcc * PC + kk) * job.in_image_stride + (long long)L * job.in_pitch;
    };
    int item = blockIdx.x;
    int lb = item / pchunks, pc = item - lb * pchunks, k = 0;
    float2 vn[H];
    if (item < n_items) {
        const float2* r = line_ptr(lb, pc, 0);
#pragma unroll
        for (int j = 0; j < H; ++j) vn[j] = (j * R + ln < N) ? r[j * R + ln] : make_float2(0.f, 0.f);
    }
    float2 tv[H];
    while (item < n_items) {
        float2 v[R];
#pragma unroll
        for (int j = 0; j < H; ++j) v[j] = vn[j];
#pragma unroll
        for (int j = H; j < R; ++j) v[j] = make_float2(0.f, 0.f);
        const int p = pc * PC + k;
        const int cur_lb = lb;
        if (k == 0) {
            const float2* trow = job.trans + frame_off(job, pc * PC) + (long long)min(lb * LINES + grp, job.n_lines - 1) * N;
#pragma unroll
            for (int j = 0; j < H; ++j) tv[j] = (j * R + ln < N) ? trow[j * R + ln] : make_float2(0.f, 0.f);
        }
        int nitem = item, nlb = lb, npc = pc, nk = k + 1;
        if (nk >= min(PC, job.n_images - pc * PC)) {
            nk = 0; nitem = item + (int)gridDim.x; nlb = lb + step_lb; npc = pc + step_pc;
            if (npc >= pchunks) { npc -= pchunks; ++nlb; }
        }
        auto prefetch_part = [&](auto lo_c, auto hi_c) {
            constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
            __builtin_amdgcn_sched_barrier(0);
            if (nitem < n_items) {
                const float2* r = line_ptr(nlb, npc, nk);
#pragma unroll
                for (int j = LO; j < HI; ++j) vn[j] = (j * R + ln < N) ? r[j * R + ln] : make_float2(0.f, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // A = ifft_N . P . fft_N as one cyclic convolution of length M: two M-point FFTs and the filter product
        auto a_conv = [&]() {
            fourstep_split_addtid<R, false, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
            mul_filter(v);
            fourstep_split_addtid<R, true, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
#pragma unroll
            for (int j = 0; j < H; ++j) if (j * R + ln >= N) v[j] = make_float2(0.f, 0.f);    // keep outputs 0 .. N-1: the padding stays zero
#pragma unroll
            for (int j = H; j < R; ++j) v[j] = make_float2(0.f, 0.f);
        };
        if (job.flags & P2_PRE_A) a_conv();
        prefetch_part(MSL_IC(0), MSL_IC(H / 2));
#pragma unroll
        for (int j = 0; j < H; ++j) v[j] = cmulf(v[j], tv[j]);
        if (job.flags & P2_POST_A) a_conv();
        prefetch_part(MSL_IC(H / 2), MSL_IC(H));
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < H; ++j) myrow[j * R + ln] = v[j];
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
                *reinterpret_cast<float4*>(dst + (off0 + i * ostep)) = make_float4(a.x, a.y, b.x, b.y);       // (streaming stores measured 1.8 % slower here)
            }
        }
        lds_barrier();
        item = nitem; lb = nlb; pc = npc; k = nk;
    }
}

// ---- 2048-point register FFT, one wave per line; lines of any length 513..1024 by the cyclic convolution on it ------------------------------
// A line of M = 2048 points in ONE wave: 64 lanes x 32 registers, element index = reg * 64 + lam(lane), where
//     lam(L) = 32 (L & 1) + (L >> 1)
// spreads the logical positions over the physical lanes so that the two lanes that share a 64-point sub-transform are
// neighbours.  Forward transform (k = k1 + 32 k2):
//     32-point register FFT over reg (n1 -> k1)  ->  x W_2048^(k1 n2)  ->  LDS transpose: lane (k1, h) gets A[k1; n2 = m + 32 h]
//     in register m  ->  the 64-point DFT over n2 as ONE radix-2 step across the lane pair (h = 0: a = x_own + x_partner,
//     h = 1: d = (x_partner - x_own) W_64^m; the partner's value comes through DPP quad_perm, no LDS)  ->  32-point register FFT
//     (h = 0 holds k2 = 2q, h = 1 holds k2 = 2q + 1: element k = 64 q + (32 h + k1) = reg * 64 + lam(lane): the input layout).
// The inverse runs the mirror image with conjugated twiddles.  Against the 2 R^2 layout of rowT2_pass_kernel<32> (64 complex
// per lane) a lane holds 32 complex, which leaves registers for the prefetch of the next line and for t_k.
__device__ __forceinline__ int lam64(int L) { return 32 * (L & 1) + (L >> 1); }
// LDS images of the tables and of the tile rows keep every block of 64 elements in PHYSICAL lane order (element 64 b + lam(L) at
// 64 b + L): a table or tile access of a wave is then 64 consecutive 8-byte words.  In logical order the lanes L and L ^ 1 sit
// 256 bytes apart -- the same banks, a two-way conflict on every such read and write (r02 profile: 33 % of the LDS cycles).
__device__ __forceinline__ int lam64_inv(int m) { return 2 * (m & 31) + (m >> 5); }
__device__ __forceinline__ int lds_pos64(int e) { return (e & ~63) + lam64_inv(e & 63); }
// lane whose element mirrors this lane's one inside a block of 64, as an offset from the block that holds element 64 r - lam(L):
// lane 0 -> element 64 r itself (next block, position 0); else block r - 1, element 64 - lam(L) = position 1 (L == 1) or 65 - L
__device__ __forceinline__ int lds_mirror64(int L) { return L == 0 ? 0 : (L == 1 ? 1 : 65 - L) - 64; }
__device__ __forceinline__ float dpp_swap_pair(float x) {          // value of the neighbouring lane (L ^ 1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}
constexpr int W2K_PITCH = 97;               // floats per k1 row of the transpose scratch: n2 < 32 at [0,32), n2 >= 32 at [48,80); 97 = 1 mod 32

// tw: LDS table T[k1 * 64 + n2] = exp(-2 pi i k1 n2 / 2048) stored at lds_pos64 (lane L reads tw[k1 * 64 + L]); w64: LDS table [h * 32 + m] = (h ? exp(-2 pi i m / 64) : 1);
// scr: this wave's scratch of 32 * W2K_PITCH floats; L = lane, la = lam64(L), sgn = (L & 1) ? -1 : +1
struct NoMid { __device__ __forceinline__ void operator()() const {} };
// mid(): called between the two halves of the transform (behind the exchange going forward, in front of it going back) -- the
// transposing pass puts a prefetch slot there
template <bool INV, int CH = 8, typename Mid = NoMid, bool DIT = true>
__device__ __forceinline__ void fft2048_wave(float2 (&v)[32], float* scr, const float2* tw, const float2* w64, int L, int la, float sgn, Mid mid = Mid()) {
    constexpr int R = 32;
    const int col = (la & 31) + 48 * (la >> 5);                     // this lane's column n2 = la in a k1 row
    const int rowbase = (L >> 1) * W2K_PITCH + 48 * (L & 1);        // lane (k1, h): row k1, columns m + 32 h
    // radix-2 step across the lane pair (L, L ^ 1) through DPP; going back the odd lane's W_64^m comes first, going forward it is
    // left to the leaf level of the register FFT that follows (weights w64: ones on the even lane)
    auto pair_step = [&](bool twiddle_first) {
#pragma unroll
        for (int c = 0; c < R; c += CH) {
            float2 w[CH];
            if (twiddle_first) {
#pragma unroll
                for (int j = 0; j < CH; ++j) w[j] = w64[(L & 1) * R + c + j];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float2 x = v[c + j];
                if (twiddle_first) x = cmulf_conj(x, w[j]);                    // inverse: d' = d~ conj(W_64^m) on the odd lane
                v[c + j] = make_float2(fmaf(sgn, x.x, dpp_swap_pair(x.x)), fmaf(sgn, x.y, dpp_swap_pair(x.y)));
                if constexpr (!DIT) if (!twiddle_first) v[c + j] = cmulf(v[c + j], w64[(L & 1) * R + c + j]);     // (.. ) W_64^m, forward
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr int LCH = CH / 2 > 0 ? CH / 2 : 1;                   // leaf butterflies per chunk of table reads
    if constexpr (!INV) {
        if constexpr (DIT) { fft_regs_dit<R, false, 0, 1>(v, v); pin_all(v); }
        else fft_regs<R, false>(v);
        mul_table<R, 1, false, 64, CH>(v, tw, L);
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) scr[k1 * W2K_PITCH + col] = v[k1].x;
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < R; ++m) v[m].x = scr[rowbase + m];
        wave_lds_fence();
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) scr[k1 * W2K_PITCH + col] = v[k1].y;
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < R; ++m) v[m].y = scr[rowbase + m];
        wave_lds_fence();
        mid();
        pair_step(false);
        if constexpr (DIT) {
            dit_leaf_chunks<R, false, 1, 1, LCH, 0>(v, v, w64, (L & 1) * R);    // (..) W_64^m on the odd lane, folded into the leaves
            dit_upper<R, false, 1>(v);
            pin_all(v);
        } else {
            fft_regs<R, false>(v);
        }
    } else {
        if constexpr (DIT) { fft_regs_dit<R, true, 0, 1>(v, v); pin_all(v); }
        else fft_regs<R, true>(v);
        pair_step(true);
        mid();
#pragma unroll
        for (int m = 0; m < R; ++m) scr[rowbase + m] = v[m].x;
        wave_lds_fence();
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) v[k1].x = scr[k1 * W2K_PITCH + col];
        wave_lds_fence();
#pragma unroll
        for (int m = 0; m < R; ++m) scr[rowbase + m] = v[m].y;
        wave_lds_fence();
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) v[k1].y = scr[k1 * W2K_PITCH + col];
        wave_lds_fence();
        if constexpr (DIT) {
            dit_leaf_chunks<R, true, 2, 64, LCH, 0>(v, v, tw, L);               // conj W_2048^{k1 n2}, folded into the leaves
            dit_upper<R, true, 1>(v);
            pin_all(v);
        } else {
            mul_table<R, 1, true, 64, CH>(v, tw, L);
            fft_regs<R, true>(v);
        }
    }
}

// The same transform with the cheap exchange of the R^2 kernels (round 4; the transposing pass of 2048-point lines).  The line comes in
// the IDENTITY layout -- element 64 j + L in register j of lane L -- and leaves in the lambda layout above (frequency 64 q + lam64(L)
// in register q); the inverse takes the lambda layout back to the identity.  With the identity on the writing side a register of
// all 64 lanes is one row of 64 consecutive floats (ds_write_addtid_b32, no address register, twice the rate of ds_write_b32), and
// lane (k1, h) reads its 32 values m + 32 h of row k1 as eight ds_read_b128; going back it writes them as eight ds_write_b128 and
// lane n2 gathers column n2 (32 ds_read_b32).  160 LDS instructions per forward + inverse pair instead of 256; rows of 68 floats
// (16-byte aligned, conflict-free both ways).  tw: T[k1 * 64 + n2] in natural order; scr_lds: LDS byte address of scr (wave-uniform).
template <bool INV, int CH = 8>
__device__ __forceinline__ void fft2048_wave_io(float2 (&v)[32], float* scr, unsigned scr_lds, const float2* tw, const float2* w64, int L, float sgn) {
    constexpr int R = 32, PW = 68;
    constexpr int LCH = CH / 2 > 0 ? CH / 2 : 1;
    float* mine = scr + (L >> 1) * PW + 32 * (L & 1);                // lane (k1, h): row k1, floats [32 h, 32 h + 32)
    auto pair_step = [&](bool twiddle_first) {
#pragma unroll
        for (int c = 0; c < R; c += CH) {
            float2 w[CH];
            if (twiddle_first) {
#pragma unroll
                for (int j = 0; j < CH; ++j) w[j] = w64[(L & 1) * R + c + j];
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                float2 x = v[c + j];
                if (twiddle_first) x = cmulf_conj(x, w[j]);
                v[c + j] = make_float2(fmaf(sgn, x.x, dpp_swap_pair(x.x)), fmaf(sgn, x.y, dpp_swap_pair(x.y)));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (!INV) {
        fft_regs_dit<R, false, 0, 1>(v, v);
        pin_all(v);
        mul_table<R, 1, false, 64, CH>(v, tw, L);
        if constexpr (!(MSL_ABL2 & 4)) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" :: "s"(scr_lds) : "memory");
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(v[k1].x), "n"(k1 * PW * 4) : "memory");
        wave_lds_fence();
#pragma unroll
        for (int g = 0; g < R / 4; ++g) {
            const float4 q = *reinterpret_cast<const float4*>(mine + 4 * g);
            v[4 * g].x = q.x; v[4 * g + 1].x = q.y; v[4 * g + 2].x = q.z; v[4 * g + 3].x = q.w;
        }
        wave_lds_fence();
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" :: "s"(scr_lds) : "memory");
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(v[k1].y), "n"(k1 * PW * 4) : "memory");
        wave_lds_fence();
#pragma unroll
        for (int g = 0; g < R / 4; ++g) {
            const float4 q = *reinterpret_cast<const float4*>(mine + 4 * g);
            v[4 * g].y = q.x; v[4 * g + 1].y = q.y; v[4 * g + 2].y = q.z; v[4 * g + 3].y = q.w;
        }
        wave_lds_fence();
        }
        pair_step(false);
        dit_leaf_chunks<R, false, 1, 1, LCH, 0>(v, v, w64, (L & 1) * R);        // (..) W_64^m on the odd lane, folded into the leaves
        dit_upper<R, false, 1>(v);
        pin_all(v);
    } else {
        fft_regs_dit<R, true, 0, 1>(v, v);
        pin_all(v);
        pair_step(true);
        if constexpr (!(MSL_ABL2 & 4)) {
#pragma unroll
        for (int g = 0; g < R / 4; ++g) *reinterpret_cast<float4*>(mine + 4 * g) = make_float4(v[4 * g].x, v[4 * g + 1].x, v[4 * g + 2].x, v[4 * g + 3].x);
        wave_lds_fence();
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) v[k1].x = scr[k1 * PW + L];
        wave_lds_fence();
#pragma unroll
        for (int g = 0; g < R / 4; ++g) *reinterpret_cast<float4*>(mine + 4 * g) = make_float4(v[4 * g].y, v[4 * g + 1].y, v[4 * g + 2].y, v[4 * g + 3].y);
        wave_lds_fence();
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) v[k1].y = scr[k1 * PW + L];
        wave_lds_fence();
        }
        dit_leaf_chunks<R, true, 2, 64, LCH, 0>(v, v, tw, L);                   // conj W_2048^{k1 n2}, folded into the leaves
        dit_upper<R, true, 1>(v);
        pin_all(v);
    }
}

// Transposing pass A . t_k . A for lines of any length 513 <= N <= 1024: rowTB_pass_kernel's convolution scheme on fft2048_wave (M = 2048).
// One wave per line, 8 lines per workgroup.  IN_P / OUT_P: the PAIRED-LINES layout of a work buffer between two such passes --
// element e of line L at (L/2) * (2 * pitch) + 2 * e + (L & 1) [float2 units], the two lines of a pair interleaved element by
// element -- in which a transposed 128-byte segment is two neighbouring positions of the tile's eight lines (in the natural
// layout eight lines give 64-byte runs).  Reading stays coalesced (a wave's load covers 512 contiguous bytes of its pair).
template <bool IN_P, bool OUT_P>
__global__ void __launch_bounds__(512, 2) rowTB2_pass_kernel(RowTJob job) {
    constexpr int R = 32, M = 2048, H = 16, NH = M / 2, LINES = 8, NT = 512, TCH = 8;
    constexpr int RS = (R * W2K_PITCH) / 2 + 1;        // tile row in float2 (1553: the transpose scratch; >= NH positions; = 17 mod 32)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // 2048
    float2* w64 = tw + M;                                     // 64
    float2* bf = w64 + 64;                                    // NH + 2
    float2* tile = bf + NH + 2;                               // LINES * RS
    const int tid = threadIdx.x;
    const int N = job.n_line;
    for (int i = tid; i < M; i += NT) tw[lds_pos64(i)] = job.tw[i];
    if (tid < 64) w64[tid] = job.tw2[tid];
    for (int i = tid; i <= NH; i += NT) bf[i] = job.bf[i];
    __syncthreads();
    const int wv = tid >> 6, L = tid & 63, la = lam64(L);
    const float sgn = (L & 1) ? -1.f : 1.f;
    float2* myrow = tile + wv * RS;
    float* scr = reinterpret_cast<float*>(myrow);
    const float2* fa = bf + la;                               // Bf[64 j + la],              j < 16
    const float2* fb = bf - la;                               // Bf[M - (64 j + la)] = bf[64 (32 - j) - la],  j >= 16
    auto mul_filter = [&](float2 (&vv)[R]) {
#pragma unroll
        for (int c = 0; c < R; c += TCH) {
            float2 w[TCH];
#pragma unroll
            for (int j = 0; j < TCH; ++j) w[j] = (c + j < H) ? fa[(c + j) * 64] : fb[(R - (c + j)) * 64];
#pragma unroll
            for (int j = 0; j < TCH; ++j) vv[c + j] = cmulf(vv[c + j], w[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int lblocks = (job.n_lines + LINES - 1) / LINES;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const int n_items = lblocks * pchunks;
    const int step_lb = (int)gridDim.x / pchunks, step_pc = (int)gridDim.x % pchunks;
    constexpr int ES = IN_P ? 2 : 1;
    auto line_ptr = [&](int lbb, int pcc, int kk) {
        const int Lc = min(lbb * LINES + wv, job.n_lines - 1);
        const long long off = IN_P ? (long long)(Lc >> 1) * (2 * job.in_pitch) + (Lc & 1) : (long long)Lc * job.in_pitch;
        return job.in + (long long)(pcc * PC + kk) * job.in_image_stride + off;
    };
    int item = blockIdx.x;
    int lb = item / pchunks, pc = item - lb * pchunks, k = 0;
    float2 vn[H];
    if (item < n_items) {
        const float2* r = line_ptr(lb, pc, 0);
#pragma unroll
        for (int j = 0; j < H; ++j) vn[j] = (j * 64 + la < N) ? ld_stream(r + ((j * 64 + la) * ES)) : make_float2(0.f, 0.f);
    }
    float2 tv[H];
    while (item < n_items) {
        float2 v[R];
#pragma unroll
        for (int j = 0; j < H; ++j) v[j] = vn[j];
#pragma unroll
        for (int j = H; j < R; ++j) v[j] = make_float2(0.f, 0.f);
        const int p = pc * PC + k;
        const int cur_lb = lb;
        if (k == 0) {
            const float2* trow = job.trans + frame_off(job, pc * PC) + (long long)min(lb * LINES + wv, job.n_lines - 1) * N;
#pragma unroll
            for (int j = 0; j < H; ++j) tv[j] = (j * 64 + la < N) ? trow[j * 64 + la] : make_float2(0.f, 0.f);
        }
        int nitem = item, nlb = lb, npc = pc, nk = k + 1;
        if (nk >= min(PC, job.n_images - pc * PC)) {
            nk = 0; nitem = item + (int)gridDim.x; nlb = lb + step_lb; npc = pc + step_pc;
            if (npc >= pchunks) { npc -= pchunks; ++nlb; }
        }
        auto prefetch_part = [&](auto lo_c, auto hi_c) {
            constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
            __builtin_amdgcn_sched_barrier(0);
            if (nitem < n_items) {
                const float2* r = line_ptr(nlb, npc, nk);
#pragma unroll
                for (int j = LO; j < HI; ++j) vn[j] = (j * 64 + la < N) ? ld_stream(r + ((j * 64 + la) * ES)) : make_float2(0.f, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto a_conv = [&]() {                                   // see rowTB_pass_kernel
            fft2048_wave<false, TCH>(v, scr, tw, w64, L, la, sgn);
            mul_filter(v);
            fft2048_wave<true, TCH>(v, scr, tw, w64, L, la, sgn);
#pragma unroll
            for (int j = 0; j < H; ++j) if (j * 64 + la >= N) v[j] = make_float2(0.f, 0.f);
#pragma unroll
            for (int j = H; j < R; ++j) v[j] = make_float2(0.f, 0.f);
        };
        if (job.flags & P2_PRE_A) a_conv();
        prefetch_part(MSL_IC(0), MSL_IC(8));
#pragma unroll
        for (int j = 0; j < H; ++j) v[j] = cmulf(v[j], tv[j]);
        if (job.flags & P2_POST_A) a_conv();
        prefetch_part(MSL_IC(8), MSL_IC(16));
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < H; ++j) myrow[j * 64 + la] = v[j];
        lds_barrier();
        if constexpr (OUT_P) {
            // segment = positions (2 mm, 2 mm + 1) of the tile's 8 lines; thread = (line i, pair mm); the four octets of a
            // half-wave take pairs 4 apart so that, with the tile row pitch = 17 mod 32, their LDS reads fall into different banks
            const int i = tid & 7, oct = tid >> 3, q = oct & 3, hh = oct >> 2;
            const int mm0 = (hh & 3) + 4 * q + 16 * (hh >> 2);             // 0 .. 63
            const float2* src = tile + i * RS + 2 * mm0;
            float2* dst = job.out + (long long)p * job.out_image_stride + 2 * (cur_lb * LINES + i);
            int off0 = mm0 * 2 * job.out_pitch;
            asm volatile("" : "+v"(off0));
            const int ostep = 64 * 2 * job.out_pitch;
#pragma unroll
            for (int it = 0; it < NH / 2 / 64; ++it) {
                if (2 * (mm0 + 64 * it) < N) {
                    const float2 a = src[it * 128], b = src[it * 128 + 1];
                    st_stream(dst + (off0 + it * ostep), a.x, a.y, b.x, b.y);
                }
            }
        } else {
            const int q4 = tid & 3, e0 = tid >> 2;                         // natural output: 8 lines = 64-byte segments
            float2* dst = job.out + (long long)p * job.out_image_stride + cur_lb * LINES + 2 * q4;
            int off0 = e0 * job.out_pitch;
            asm volatile("" : "+v"(off0));
            const int ostep = (NT / 4) * job.out_pitch;
#pragma unroll
            for (int it = 0; it < NH / (NT / 4); ++it) {
                const int e = e0 + (NT / 4) * it;
                if (e < N) {
                    const float2 a = tile[(2 * q4) * RS + e], b = tile[(2 * q4 + 1) * RS + e];
                    st_stream(dst + (off0 + it * ostep), a.x, a.y, b.x, b.y);
                }
            }
        }
        lds_barrier();
        item = nitem; lb = nlb; pc = npc; k = nk;
    }
}

// Transposing pass A . t_k . A for lines of 1025 .. 2047 points: the convolution form of rowTB2_pass_kernel with the cyclic length
// M = 4096, the 4096-point transforms built from fft2048_wave by one radix-2 step that the zero padding makes half trivial:
//   forward  (x[n + 2048] = 0):  X[2k] = FFT_2048(x)[k],  X[2k+1] = FFT_2048(x W_4096^n)[k]
//   filter in that split order:   X[2k] *= Q[2k], X[2k+1] *= Q[2k+1]   (both halves symmetric: Q[2k] about k = 1024, Q[2k+1] about 1023.5)
//   inverse, outputs n < 2048 only:  y[n] = IFFT_2048(X[2k])[n] + conj(W_4096^n) IFFT_2048(X[2k+1])[n]
// i.e. four 2048-point transforms per propagation, eight per pass.  The two branches of the radix-2 step run on TWO waves: wave 2l
// takes X[2k] (E = IFFT(FFT(x) Qe)), wave 2l + 1 takes X[2k+1] (O = conj(W) IFFT(FFT(x W) Qo)), the halves meet in the LDS
// (y = E + O).  A wave then holds ONE set of 32 complex registers -- both branches in one wave spill 916 B per lane -- at the
// price of four lines per workgroup (32-byte runs in the transposed store; the pass is bound by its eight 2048-point transforms
// per line, not by the stores).  The two transpose-scratch rows of a pair (2 x 1553 float2) double as the pair's exchange buffer
// (2048 float2) whenever neither wave is inside a transform.
// job.bf: Q[2k] (k = 0..1024, padded to 1026) followed by Q[2k+1] (k = 0..1023); job.bw: W_4096^n, n < 2048; job.n_line = N.
__global__ void __launch_bounds__(512, 2) rowTC2_pass_kernel(RowTJob job) {
    constexpr int R = 32, M2 = 2048, LINES = 4, NT = 512, TCH = 8;
    constexpr int RS = (R * W2K_PITCH) / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // 2048, lane order
    float2* w64 = tw + M2;                                    // 64
    float2* wq = w64 + 64;                                    // 2048: W_4096^n, lane order
    float2* qe = wq + M2;                                     // 1026
    float2* qo = qe + 1026;                                   // 1024 (+2)
    float2* tile = qo + 1026;                                 // 8 rows of RS: rows 2l, 2l + 1 = pair l
    const int tid = threadIdx.x;
    const int N = job.n_line;
    for (int i = tid; i < M2; i += NT) { tw[lds_pos64(i)] = job.tw[i]; wq[lds_pos64(i)] = job.bw[i]; }
    if (tid < 64) w64[tid] = job.tw2[tid];
    for (int i = tid; i < 2052; i += NT) qe[i] = job.bf[i];
    __syncthreads();
    const int wv = tid >> 6, L = tid & 63, la = lam64(L);
    const int line = wv >> 1, role = wv & 1;
    const float sgn = (L & 1) ? -1.f : 1.f;
    const int li = tid % LINES, r0 = tid / LINES;
    float* scr = reinterpret_cast<float*>(tile + wv * RS);
    float2* pair = tile + 2 * line * RS;                      // the pair's exchange buffer: 2048 float2, natural order
    const float2* fa = (role ? qo : qe) + la;                 // first half of this branch's filter
    const float2* fb = role ? qo + 63 - la : qe - la;         // mirrored half: Q[2 (2048 - m)] = qe[.. - la], Q[2 (2047 - m) + 1] = qo[.. + 63 - la]
    const int lblocks = (job.n_lines + LINES - 1) / LINES;
    const int n_items = lblocks * job.n_images;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int p = item / lblocks, lb = item - p * lblocks;
        const int Lc = min(lb * LINES + line, job.n_lines - 1);
        float2 v[R];
        {
            const float2* r = job.in + (long long)p * job.in_image_stride + (long long)Lc * job.in_pitch;
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = (j * 64 + la < N) ? ld_stream(r + (j * 64 + la)) : make_float2(0.f, 0.f);
        }
        auto mul_wq = [&](auto conj_c) {
            constexpr bool CONJ = decltype(conj_c)::value;
#pragma unroll
            for (int c = 0; c < R; c += TCH) {
                float2 w[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) w[j] = wq[(c + j) * 64 + L];
#pragma unroll
                for (int j = 0; j < TCH; ++j) v[c + j] = CONJ ? cmulf_conj(v[c + j], w[j]) : cmulf(v[c + j], w[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // this wave's branch of the cyclic convolution; afterwards role 0 holds y (outputs 0 .. N-1, the rest zero)
        auto conv = [&]() __attribute__((always_inline)) {
            if (role) mul_wq(std::false_type{});
            fft2048_wave<false, TCH, NoMid, false>(v, scr, tw, w64, L, la, sgn);       // (the fused-multiply-add network spills in this kernel)
#pragma unroll
            for (int c = 0; c < R; c += TCH) {
                float2 w[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) w[j] = (c + j < 16) ? fa[(c + j) * 64] : fb[(role ? R - 1 - (c + j) : R - (c + j)) * 64];
#pragma unroll
                for (int j = 0; j < TCH; ++j) v[c + j] = cmulf(v[c + j], w[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            fft2048_wave<true, TCH, NoMid, false>(v, scr, tw, w64, L, la, sgn);
            if (role) mul_wq(std::true_type{});
            lds_barrier();                                  // every wave has left its transform: the rows are free
            if (role) {
#pragma unroll
                for (int j = 0; j < R; ++j) pair[j * 64 + la] = v[j];
            }
            lds_barrier();
            if (!role) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const float2 o = pair[j * 64 + la];
                    v[j] = (j * 64 + la < N) ? make_float2(v[j].x + o.x, v[j].y + o.y) : make_float2(0.f, 0.f);
                }
            }
        };
        if (job.flags & P2_PRE_A) conv();
        if (!role) {
            const float2* trow = job.trans + frame_off(job, p) + (long long)Lc * N;
#pragma unroll
            for (int c = 0; c < R; c += TCH) {
                float2 t[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) t[j] = ((c + j) * 64 + la < N) ? trow[(c + j) * 64 + la] : make_float2(0.f, 0.f);
#pragma unroll
                for (int j = 0; j < TCH; ++j) v[c + j] = cmulf(v[c + j], t[j]);
            }
        }
        if (job.flags & P2_POST_A) {
            lds_barrier();                                  // role 0 has read the first convolution's O
            if (!role) {
#pragma unroll
                for (int j = 0; j < R; ++j) pair[j * 64 + la] = v[j];
            }
            lds_barrier();
            if (role) {
#pragma unroll
                for (int j = 0; j < R; ++j) v[j] = pair[j * 64 + la];
            }
            lds_barrier();                                  // ... before anybody's transform writes its scratch row again
            conv();
        }
        lds_barrier();                                      // role 0 has read O: the pair buffers become the tile rows
        if (!role) {
#pragma unroll
            for (int j = 0; j < R; ++j) pair[j * 64 + la] = v[j];
        }
        lds_barrier();
        {
            float2* dst = job.out + (long long)p * job.out_image_stride + lb * LINES + li;
            const float2* srcrow = tile + 2 * li * RS;
#pragma unroll
            for (int i = 0; i < M2 / (NT / LINES); ++i) {
                const int pos = r0 + (NT / LINES) * i;
                if (pos < N) dst[(long long)pos * job.out_pitch] = srcrow[pos];
            }
        }
        lds_barrier();
    }
}

// Transposing pass A . t_k . A for 2048-point lines on fft2048_wave: one wave per line, 8 lines per workgroup, the next line
// prefetched in registers, t_k in registers across a chunk of probes (rowT_pass_kernel's scheme; the 2 R^2 layout of
// rowT2_pass_kernel<32> has room for neither).  IN_P / OUT_P: paired-lines layout of the work buffers between two passes.
template <bool IN_P, bool OUT_P>
__global__ void __launch_bounds__(512, 2) rowTW_pass_kernel(RowTJob job) {
    constexpr int R = 32, N = 2048, H = 16, LINES = 8, NT = 512, TCH = 8;
    constexpr int RS = N + 1;                                 // tile row in float2 (>= the 32 x 97 floats of transpose scratch; = 1 mod 32)
    constexpr int NHT = N / 2 + 64;                           // N/2 + 2 entries; the last block is stored in lane order too
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // 2048
    float2* w64 = tw + N;                                     // 64
    float2* plh = w64 + 64;                                   // N/2 + 2: symmetric Fresnel table, first half
    float2* tile = plh + NHT;                                 // LINES * RS
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += NT) tw[i] = job.tw[i];          // natural order: fft2048_wave_io
    if (tid < 64) w64[tid] = job.tw2[tid];
    for (int i = tid; i <= N / 2; i += NT) plh[lds_pos64(i)] = job.pl[i];
    __syncthreads();
    const int wv = tid >> 6, L = tid & 63, la = lam64(L);
    const float sgn = (L & 1) ? -1.f : 1.f;
    float2* myrow = tile + wv * RS;
    float* scr = reinterpret_cast<float*>(myrow + (wv & 1));      // the exchange scratch (32 rows x 68 floats) on a 16-byte boundary: RS is odd
    const unsigned scr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)scr);
    const float2* pa = plh + L;                               // P[64 j + lam(L)]
    const float2* pb = plh + lds_mirror64(L);                 // pb[64 r] = P[64 r - lam(L)]
    auto mul_p = [&](float2 (&vv)[R]) {
#pragma unroll
        for (int c = 0; c < R; c += TCH) {
            float2 w[TCH];
#pragma unroll
            for (int j = 0; j < TCH; ++j) w[j] = (c + j < H) ? pa[(c + j) * 64] : pb[(R - (c + j)) * 64];
#pragma unroll
            for (int j = 0; j < TCH; ++j) vv[c + j] = cmulf(vv[c + j], w[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int lblocks = job.n_lines / LINES;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const int n_items = lblocks * pchunks;
    const int step_lb = (int)gridDim.x / pchunks, step_pc = (int)gridDim.x % pchunks;
    // The work buffer between two of these passes (IN_P / OUT_P): lines in pairs -- the pair L/2 shares a row of 2 x pitch entries --
    // and inside a pair the elements in chunks of 128: entry 256 (e >> 7) + ((2 (e & 63) + (L & 1)) * 2 + ((e >> 6) & 1)) holds element
    // e of line L.  A lane's registers 2 jp and 2 jp + 1 (elements 128 jp + la and 128 jp + 64 + la) are adjacent: 16-byte loads, a
    // wave's load covers 1 KB of which it uses half and its pair's wave the other half (round 2's layout 2 e + (L & 1) made every
    // load an 8-byte access at a 16-byte stride).  The writer's tile is 4 lines l0 .. l0 + 3 of block 2 jp and the same 4 of block
    // 2 jp + 1 (tile row r = line 128 jp + 64 (r & 1) + l0 + (r >> 1)): with the two positions of a position pair they fill 16
    // consecutive entries -- the 128-byte segments of before.
    auto line_of = [&](int lbb) {
        if constexpr (OUT_P) return 128 * (lbb >> 4) + 64 * (wv & 1) + 4 * (lbb & 15) + (wv >> 1);
        else return lbb * LINES + wv;
    };
    auto line_ptr = [&](int lbb, int pcc, int kk) {
        const int Ln = line_of(lbb);
        const long long off = IN_P ? (long long)(Ln >> 1) * (2 * job.in_pitch) + (2 * L + (Ln & 1)) * 2 : (long long)Ln * job.in_pitch + L;     // identity layout: element 64 j + L
        return job.in + (long long)(pcc * PC + kk) * job.in_image_stride + off;
    };
    auto load_regs = [&](float2 (&dst)[R], const float2* r, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        if constexpr (IN_P) {
#pragma unroll
            for (int jp = LO / 2; jp < HI / 2; ++jp) {
                const msl_f4v t = __builtin_nontemporal_load(reinterpret_cast<const msl_f4v*>(r + 256 * jp));
                dst[2 * jp] = make_float2(t.x, t.y); dst[2 * jp + 1] = make_float2(t.z, t.w);
            }
        } else {
#pragma unroll
            for (int j = LO; j < HI; ++j) dst[j] = ld_stream(r + j * 64);
        }
    };
    int item = blockIdx.x;
    int lb = item / pchunks, pc = item - lb * pchunks, k = 0;
    float2 vn[R];
    if (item < n_items) load_regs(vn, line_ptr(lb, pc, 0), MSL_IC(0), MSL_IC(R));
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): nothing in flight at the loop entry, see rowT_pass_kernel
    float2 tv[R];
#ifdef MSL_CLOCK
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    while (item < n_items) {
        // (No stagger of the waves here.  This pass is not at the power limit -- 1 210-1 280 W, 2.29 GHz -- but its eight waves run in
        // lockstep between the two barriers of an iteration and meet at every exchange: without the tile, the barriers and the store
        // phase an iteration takes 28.9 k cycles, with the two barriers alone 40.0 k, complete 44.5 k (tools/rowtw_bench.hip,
        // profiles/r04_k2048_cycles.txt).  Delaying waves by SIMD (0 / 256 / 512 / 768 cycles) costs 2 %, delaying waves 4-7 by 1-6 k
        // cycles 3 %: MSL_STAGGER_W / MSL_STAGGER_HALF rebuild those experiments.)
#ifdef MSL_STAGGER_W
        for (int i = __builtin_amdgcn_readfirstlane(tid >> 6) & 3; i > 0; --i) __builtin_amdgcn_s_sleep(MSL_STAGGER_W);
#endif
#ifdef MSL_STAGGER_HALF
        if (__builtin_amdgcn_readfirstlane(tid >> 8)) __builtin_amdgcn_s_sleep(MSL_STAGGER_HALF);
#endif
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = vn[j];
        const int p = pc * PC + k;
        const int cur_lb = lb;
        if (k == 0) {
            const float2* trow = job.trans + frame_off(job, pc * PC) + (long long)line_of(lb) * N + L;
#pragma unroll
            for (int j = 0; j < R; ++j) tv[j] = trow[j * 64];
        }
        int nitem = item, nlb = lb, npc = pc, nk = k + 1;
        if (nk >= min(PC, job.n_images - pc * PC)) {
            nk = 0; nitem = item + (int)gridDim.x; nlb = lb + step_lb; npc = pc + step_pc;
            if (npc >= pchunks) { npc -= pchunks; ++nlb; }
        }
        // unconditional prefetch (past the last item: the current line again), one address per iteration
        const bool more = nitem < n_items;
        const float2* nptr = line_ptr(more ? nlb : lb, more ? npc : pc, more ? nk : k);
        auto prefetch_part = [&](auto lo_c, auto hi_c) {
            constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(MSL_ABL2 & 1)) load_regs(vn, nptr, MSL_IC(LO), MSL_IC(HI));
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (MSL_ABL2 & 8) job.flags = 0;
        if (job.flags & P2_PRE_A) fft2048_wave_io<false, TCH>(v, scr, scr_lds, tw, w64, L, sgn);
        prefetch_part(MSL_IC(0), MSL_IC(8));
        if (job.flags & P2_PRE_A) {
            mul_p(v);
            fft2048_wave_io<true, TCH>(v, scr, scr_lds, tw, w64, L, sgn);
        }
        prefetch_part(MSL_IC(8), MSL_IC(16));
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = cmulf(v[j], tv[j]);
        if (job.flags & P2_POST_A) fft2048_wave_io<false, TCH>(v, scr, scr_lds, tw, w64, L, sgn);
        prefetch_part(MSL_IC(16), MSL_IC(24));
        if (job.flags & P2_POST_A) {
            mul_p(v);
            fft2048_wave_io<true, TCH>(v, scr, scr_lds, tw, w64, L, sgn);
        }
        prefetch_part(MSL_IC(24), MSL_IC(32));
        if constexpr (MSL_ABL2 & 32) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < R; ++j) acc += v[j].x + v[j].y;
            if (acc == 1.2345e-30f) job.out[tid] = make_float2(acc, acc);
            item = nitem; lb = nlb; pc = npc; k = nk;
            continue;
        }
        wave_lds_fence();
        if constexpr (!(MSL_ABL2 & 128)) {
#pragma unroll
        for (int j = 0; j < R; ++j) myrow[j * 64 + lam64_inv(L)] = v[j];          // element 64 j + L at lds_pos64
        }
        lds_barrier();
        if constexpr (MSL_ABL2 & 64) {
        } else if constexpr (OUT_P) {
            // one 16-byte store = position 2 mm + c of the tile rows 2 i and 2 i + 1 (blocks 2 jp / 2 jp + 1 of the output line pair
            // mm); eight lanes (i, c) make a 128-byte run; the four octets of a half-wave take pairs 8 positions apart (LDS banks)
            const int i8 = tid & 7, i = i8 >> 1, c = i8 & 1, oct = tid >> 3, q = oct & 3, hh = oct >> 2;
            const int mm0 = 2 * q + (hh & 1) + 8 * ((hh >> 1) & 1) + 16 * ((hh >> 2) & 1) + 32 * (hh >> 3);
            const float2* src = tile + (2 * i) * RS + lds_pos64(2 * mm0 + c);
            float2* dst = job.out + (long long)p * job.out_image_stride + 256 * (cur_lb >> 4) + (2 * (4 * (cur_lb & 15) + i) + c) * 2;
            int off0 = mm0 * 2 * job.out_pitch;
            asm volatile("" : "+v"(off0));
            const int ostep = 64 * 2 * job.out_pitch;
#pragma unroll
            for (int it = 0; it < N / 2 / 64; ++it) {
                const float2 a = src[it * 128], b = src[RS + it * 128];
#if MSL_ABL2 & 2
                if (a.x == 1.2345e-30f)
#endif
                st_stream(dst + (off0 + it * ostep), a.x, a.y, b.x, b.y);
            }
        } else {
            const int q4 = tid & 3, e0 = tid >> 2;
            float2* dst = job.out + (long long)p * job.out_image_stride + cur_lb * LINES + 2 * q4;
            int off0 = e0 * job.out_pitch;
            asm volatile("" : "+v"(off0));
            const int ostep = (NT / 4) * job.out_pitch;
#pragma unroll
            for (int it = 0; it < N / (NT / 4); ++it) {
                const int e = e0 + (NT / 4) * it;
                const float2 a = tile[(2 * q4) * RS + lds_pos64(e)], b = tile[(2 * q4 + 1) * RS + lds_pos64(e)];
                st_stream(dst + (off0 + it * ostep), a.x, a.y, b.x, b.y);
            }
        }
        lds_barrier();
        item = nitem; lb = nlb; pc = npc; k = nk;
    }
#ifdef MSL_CLOCK
    if (tid == 0 && job.clk) {
        job.clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        job.clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}

// ---- lines of N = 2 R^2 points (512 = 2*16^2) ---------------------------------------------------------------
// One radix-2 step wrapped around two four-step transforms.  A group of R lanes holds two register sets; in the
// natural domain set b, register j, lane l is element b*R^2 + j*R + l.  The forward transform is decimation in
// frequency (a = x0 + x1, d = (x0 - x1) W_N^m, then N/2-point transforms of a and d give X[2k] and X[2k+1]); the
// inverse is the mirror decimation in time.  The frequency domain therefore lives in "split" order -- set b, slot k
// holds X[2k+b] -- which never leaves the kernel: the propagator table is stored in the same order.
// XM: exchange of the two R^2-point transforms -- 0 narrow, 1 wide reads, 2 wide reads + add-tid stores (then wscr / wscr_lds /
// lane64 describe the wave's shared scratch, see fourstep_split_addtid)
template <int R, bool INV, int XM = 0>
__device__ __forceinline__ void line2_transform(float2 (&v)[2 * R], float* scratch, const float2* tw, const float2* tw2, int ln,
                                                const float* wscr = nullptr, unsigned wscr_lds = 0, int lane64 = 0) {
    static_assert(64 % R == 0, "an R-lane group must lie inside one wave: the scratch is ordered per wave only");
    float2 (&lo)[R] = reinterpret_cast<float2 (&)[R]>(v[0]);
    float2 (&hi)[R] = reinterpret_cast<float2 (&)[R]>(v[R]);
    constexpr int CH = 8;                              // table chunks with scheduling barriers: see mul_table
    if constexpr (!INV) {
#pragma unroll
        for (int c = 0; c < R; c += CH) {
            float2 w[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) w[j] = tw2[(c + j) * R + ln];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float2 a = v[c + j], b = v[R + c + j];
                v[c + j] = make_float2(a.x + b.x, a.y + b.y);
                v[R + c + j] = cmulf(make_float2(a.x - b.x, a.y - b.y), w[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        { if constexpr (XM == 2) fourstep_split_addtid<R, false>(lo, wscr, wscr_lds, tw, ln, lane64); else fourstep_split<R, false>(lo, scratch, tw, ln); }
        __builtin_amdgcn_sched_barrier(0);
        { if constexpr (XM == 2) fourstep_split_addtid<R, false>(hi, wscr, wscr_lds, tw, ln, lane64); else fourstep_split<R, false>(hi, scratch, tw, ln); }
    } else {
        { if constexpr (XM == 2) fourstep_split_addtid<R, true>(lo, wscr, wscr_lds, tw, ln, lane64); else fourstep_split<R, true>(lo, scratch, tw, ln); }
        __builtin_amdgcn_sched_barrier(0);
        { if constexpr (XM == 2) fourstep_split_addtid<R, true>(hi, wscr, wscr_lds, tw, ln, lane64); else fourstep_split<R, true>(hi, scratch, tw, ln); }
#pragma unroll
        for (int c = 0; c < R; c += CH) {
            float2 w[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) w[j] = tw2[(c + j) * R + ln];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float2 e = v[c + j], o = cmulf_conj(v[R + c + j], w[j]);
                v[c + j] = make_float2(e.x + o.x, e.y + o.y);
                v[R + c + j] = make_float2(e.x - o.x, e.y - o.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Transposing pass A . t_k . A for N = 2 R^2 with R = 16 (N = 512): next line prefetched and t_k kept in registers like
// rowT_pass_kernel, full-line tile.  (2048 = 2 * 32^2 holds 64 complex per lane in this layout, with room for neither: 2048-point
// lines run on the wave-per-line transform, rowTW_pass_kernel.)
// IN_P / OUT_P: interleaved line order of the work buffer on the input / output side (see rowT_pass_kernel): 16-byte loads.
template <int R, bool IN_P = false, bool OUT_P = false>
__global__ void __launch_bounds__(16 * R, 2) rowT2_pass_kernel(RowTJob job) {      // two waves per SIMD: at most 256 VGPRs + AGPRs
    static_assert(R == 16, "the 2 R^2 layout is used for 512-point lines only");
    constexpr int N2 = R * R, N = 2 * N2, NT = 16 * R;
    constexpr int XM = 2;                              // exchange: 16-byte reads and add-tid stores (rows 16-byte aligned: even pitch)
    constexpr int CS = N + 2;                          // tile line pitch: 2 mod 32 -- conflict-free staging
    constexpr int POS_PER_IT = NT / 8;
    constexpr int NIT = N / POS_PER_IT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);            // N2: four-step twiddles
    float2* tw2 = tw + N2;                                       // N2: W_N^m of the outer radix-2 step
    float2* pl = tw2 + N2;                                       // N: propagator, split order
    float2* tile = pl + N;                                       // 16 * CS, also the groups' transpose scratch
    const int tid = threadIdx.x;
    for (int i = tid; i < N2; i += NT) { tw[i] = job.tw[i]; tw2[i] = job.tw2[i]; }
    for (int i = tid; i < N; i += NT) pl[i] = job.pl[i];
    __syncthreads();
    const int grp = tid / R, ln = tid % R;
    const int q = tid & 7, r0 = tid >> 3;
    float* scratch = reinterpret_cast<float*>(tile + grp * CS);
    const float* wscr = reinterpret_cast<const float*>(tile + (grp - grp % (64 / R)) * CS);     // the wave's shared exchange scratch
    const unsigned wscr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(tile + (grp - grp % (64 / R)) * CS));
    const int lblocks = job.n_lines / 16;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const int n_items = lblocks * pchunks;
    const int step_lb = (int)gridDim.x / pchunks, step_pc = (int)gridDim.x % pchunks;
    int item = blockIdx.x;
    int lb = item / pchunks, pc = item % pchunks, k = 0;
    // input line of tile row g in line block lbb (OUT_P: 8 lines of block 2 jp and the same 8 of block 2 jp + 1 of the reader's
    // 2 R' chunks; the thread index is re-derived at every use: the kernel sits at its register limit)
    auto line_of = [&](int lbb, int g) {
        if constexpr (OUT_P) {
            const int sh = job.perm_shift;
            return (((lbb >> sh) * 2 + (g & 1)) << (sh + 3)) + 8 * (lbb & ((1 << sh) - 1)) + (g >> 1);
        } else {
            return lbb * 16 + g;
        }
    };
    auto image_base = [&](int pcc, int kk) { return job.in + (long long)(pcc * PC + kk) * job.in_image_stride; };
    auto load_line = [&](float2 (&dst)[2 * R], const float2* img, int lbb) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const float2* r = img + (long long)line_of(lbb, t / R) * job.in_pitch;
        const int l = t % R;
        if constexpr (IN_P) {
#pragma unroll
            for (int jp = 0; jp < R; ++jp) {
                const msl_f4v q4 = __builtin_nontemporal_load(reinterpret_cast<const msl_f4v*>(r + (2 * R * jp + 2 * l)));
                dst[2 * jp] = make_float2(q4.x, q4.y); dst[2 * jp + 1] = make_float2(q4.z, q4.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) dst[j] = ld_stream(r + (l + j * R));
        }
    };
    float2 vn[2 * R];
    float2 tv[2 * R];
    if (item < n_items) load_line(vn, image_base(pc, 0), lb);
    while (item < n_items) {
        float2 v[2 * R];
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) v[j] = vn[j];
        const int p = pc * PC + k;
        const int cur_lb = lb;
        if (k == 0) {
            int toff = tid;                               // re-derived here (one chunk of probes in PC): not worth a register across the loop
            asm volatile("" : "+v"(toff));
            const float2* trow = job.trans + frame_off(job, pc * PC) + (long long)line_of(lb, toff / R) * N + (toff % R);
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) tv[j] = trow[j * R];
        }
        int nitem = item, nlb = lb, npc = pc, nk = k + 1;
        if (nk >= min(PC, job.n_images - pc * PC)) {
            nk = 0; nitem = item + (int)gridDim.x; nlb = lb + step_lb; npc = pc + step_pc;
            if (npc >= pchunks) { npc -= pchunks; ++nlb; }
        }
        if (nitem < n_items) load_line(vn, image_base(npc, nk), nlb);      // (a mid-iteration prefetch as in rowT_pass_kernel measured 3% slower here)
        if (job.flags & P2_PRE_A) {
            line2_transform<R, false, XM>(v, scratch, tw, tw2, ln, wscr, wscr_lds, tid & 63);
            mul_table<2 * R, 0, false, R>(v, pl, ln);
            line2_transform<R, true, XM>(v, scratch, tw, tw2, ln, wscr, wscr_lds, tid & 63);
        }
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) v[j] = cmulf(v[j], tv[j]);
        if (job.flags & P2_POST_A) {
            line2_transform<R, false, XM>(v, scratch, tw, tw2, ln, wscr, wscr_lds, tid & 63);
            mul_table<2 * R, 0, false, R>(v, pl, ln);
            line2_transform<R, true, XM>(v, scratch, tw, tw2, ln, wscr, wscr_lds, tid & 63);
        }
        float2* dst = job.out + ((long long)p * job.out_image_stride + cur_lb * 16);
        int off0 = 2 * q + r0 * job.out_pitch;
        asm volatile("" : "+v"(off0));
        const int ostep = POS_PER_IT * job.out_pitch;
        wave_lds_fence();
        float2* myrow = tile + grp * CS;
#pragma unroll
        for (int j = 0; j < N / R; ++j) myrow[j * R + ln] = v[j];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pos = r0 + POS_PER_IT * i;
            const float2 a = tile[(2 * q) * CS + pos], b = tile[(2 * q + 1) * CS + pos];
            st_stream(dst + (off0 + i * ostep), a.x, a.y, b.x, b.y);
        }
        lds_barrier();
        item = nitem; lb = nlb; pc = npc; k = nk;
    }
}

// ---- stand-alone inverse FFT passes for lines of 2 R^2 = 512 points: the potential build on 512 x 512 grids ---------------------
// V_s = Re ifft2(R_s) / (dx^2 dy^2), t_s = exp(i sigma V_s) (potentials.py:336-342, multislice.py:282) went through the generic LDS
// kernel on these grids (two passes at 2.6 TB/s plus a transposition of every second slice; 0.35 of 0.73 ms per frame at 512^2 x
// 100, which a single probe cannot amortise).  Two passes of this kernel instead: the line is loaded in the split order the
// register transform wants -- slot k of set b is X[2k + b], i.e. one 16-byte load per lane and register -- inverse-transformed
// (line2_transform) and stored transposed through the 16-line tile; the second pass applies the potential epilogue and writes a
// slice either transposed back (natural orientation) or as rows (the orientation the passes along x read).
struct IfftT2Job {
    const float2* in;           // (n_images, n_lines, in_pitch): lines in natural frequency order
    float2* out_t;              // transposed store: out_t[img][pos][line]
    float2* out_rows;           // row store (potential epilogue only): out_rows[img][line][pos]
    const float2* tw;
    const float2* tw2;
    long long in_is, out_t_is, out_rows_is;
    int in_pitch, out_t_pitch, out_rows_pitch, n_lines, n_images;
    int potential;              // 1: V = Re(.) * scale, out = exp(i sigma V)
    int rows_parity;            // potential: images whose slice number has this parity are stored as rows, the others transposed; -1: none
    int slice_mod;              // slice number of image img = img % slice_mod (several frames' stacks in one launch); 0: img itself
    int herm;                   // 1: only elements 0 .. N/2 of an input line exist, the others are conj(line[N - e]) (spectrum of a real image)
    float scale, sigma_over_pi;
};

// slice number of an image of a potential-build launch (the stacks of several frames follow each other)
template <typename Job>
__device__ __forceinline__ int slice_of(const Job& job, int img) { return job.slice_mod > 0 ? img % job.slice_mod : img; }

// PAIR (second pass: job.herm and job.potential set, line count a multiple of 32): two real lines per complex transform,
// Z = X_a + i X_b -> V_a + i V_b; a group takes the lines g and g + 16 of a 32-line block (ifftTB_kernel's note).  Line a of the
// next item is prefetched in registers as before, line b arrives at the top of the item (two prefetched lines do not fit).
template <int R, bool PAIR = false>
__global__ void __launch_bounds__(16 * R, 2) ifftT2_kernel(IfftT2Job job) {
    static_assert(R == 16, "512-point lines only (64 complex values per lane at R = 32 spill: 2048-point axes use ifftTW_kernel)");
    constexpr int N2 = R * R, N = 2 * N2, NT = 16 * R;
    constexpr int BL = PAIR ? 32 : 16;                             // lines per work item
    constexpr int CPOS = N;
    constexpr int CS = CPOS + 1;
    constexpr int POS_PER_IT = NT / 8;
    constexpr int NIT = CPOS / POS_PER_IT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);
    float2* tw2 = tw + N2;
    float2* tile = tw2 + N2;                                     // 16 * CS, also the transpose scratch
    const int tid = threadIdx.x;
    for (int i = tid; i < N2; i += NT) { tw[i] = job.tw[i]; tw2[i] = job.tw2[i]; }
    __syncthreads();
    const int grp = tid / R, ln = tid % R;
    const int q = tid & 7, r0 = tid >> 3;
    float* scratch = reinterpret_cast<float*>(tile + grp * CS);
    const int lblocks = job.n_lines / BL;
    const int n_items = lblocks * job.n_images;
    // the next item's line is loaded into registers while the current one is transformed (two 256-thread workgroups per CU = two
    // waves per SIMD: without it the loads of an item were exposed -- 193 us of a 0.33 ms potential per frame at 512^2 x 100)
    float2 vn[2 * R];
    // slot j of set b <- element 2 (j R + ln) + b of line `line`; with herm the mirrored element (conjugated unless RAW)
    auto load_line = [&](int it, int line_in_block, float2 (&dst)[2 * R], auto raw_c) {
        constexpr bool RAW = decltype(raw_c)::value;
        const int img = it / lblocks, lb = it - img * lblocks;
        const float2* src = job.in + (long long)img * job.in_is + (long long)(lb * BL + line_in_block) * job.in_pitch;
        int lnl = ln;
        asm volatile("" : "+v"(lnl));
        if (job.herm) {                                     // workgroup-uniform: mirrored elements as two 8-byte loads
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int e0 = 2 * (j * R + lnl), e1 = e0 + 1;
                float2 a = src[e0 <= N / 2 ? e0 : N - e0], b = src[e1 <= N / 2 ? e1 : N - e1];
                if (!RAW && e0 > N / 2) a.y = -a.y;
                if (!RAW && e1 > N / 2) b.y = -b.y;
                dst[j] = a; dst[R + j] = b;
            }
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const float4 x = *reinterpret_cast<const float4*>(src + 2 * (j * R + lnl));
                dst[j] = make_float2(x.x, x.y); dst[R + j] = make_float2(x.z, x.w);
            }
        }
    };
    using RawC = std::integral_constant<bool, PAIR>;
    if ((int)blockIdx.x < n_items) load_line(blockIdx.x, grp, vn, RawC{});
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / lblocks, lb = item - img * lblocks;
        int lnv = ln;                                       // laundered: keeps dozens of per-lane LDS / global addresses from being
        asm volatile("" : "+v"(lnv));                       // hoisted out of the loop into registers the transform needs
        float2 v[2 * R];
        if constexpr (PAIR) {
            float2 vb[2 * R];
            load_line(item, 16 + grp, vb, RawC{});
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) {               // element 2 ((j % R) R + ln) + j / R: a + i b, mirrored half conj(a) + i conj(b)
                const int e = 2 * ((j % R) * R + lnv) + j / R;
                float2 a = vn[j], b = vb[j];
                if (e == N / 2) a.y = b.y = 0.f;            // (the Nyquist element is its own mirror image: ifftTB_kernel's note)
                v[j] = (e <= N / 2) ? make_float2(a.x - b.y, a.y + b.x) : make_float2(a.x + b.y, b.x - a.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) v[j] = vn[j];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (item + (int)gridDim.x < n_items) load_line(item + (int)gridDim.x, grp, vn, RawC{});
        __builtin_amdgcn_sched_barrier(0);
        float* scr = scratch;
        asm volatile("" : "+v"(scr));
        line2_transform<R, true>(v, scr, tw, tw2, lnv);
        // exp(i sigma V) through sincospi: its range reduction is exact, where sincosf carries a slow path with a private
        // array (scratch); sigma / pi is formed in double on the host.
        auto trans_of = [&](float x) {
            float sn, cs;
            sincospif(job.sigma_over_pi * (x * job.scale), &sn, &cs);
            return make_float2(cs, sn);
        };
        float vy[PAIR ? 2 * R : 1];
        if constexpr (PAIR) {
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) vy[j] = v[j].y;
        }
        const bool as_rows = job.potential && (slice_of(job, img) & 1) == job.rows_parity;       // workgroup-uniform
#pragma unroll
        for (int half = 0; half < (PAIR ? 2 : 1); ++half) {
            if (job.potential) {
#pragma unroll
                for (int j = 0; j < 2 * R; ++j) {
                    if (j % 4 == 0) __builtin_amdgcn_sched_barrier(0);          // four evaluations in flight, not all of them
                    v[j] = trans_of(half == 0 ? v[j].x : vy[j]);
                }
            }
            if (as_rows) {
                float2* dst = job.out_rows + (long long)img * job.out_rows_is + (long long)(lb * BL + half * 16 + grp) * job.out_rows_pitch;
#pragma unroll
                for (int j = 0; j < 2 * R; ++j) dst[(j / R) * N2 + (j % R) * R + lnv] = v[j];
                continue;
            }
            float2* dst = job.out_t + (long long)img * job.out_t_is + lb * BL + half * 16;
            int off0 = 2 * q + r0 * job.out_t_pitch;
            asm volatile("" : "+v"(off0));
            const int ostep = POS_PER_IT * job.out_t_pitch;
            lds_barrier();                             // the tile (and every group's scratch use) is free
            float2* myrow = reinterpret_cast<float2*>(scr);
#pragma unroll
            for (int j = 0; j < CPOS / R; ++j) myrow[j * R + lnv] = v[j];
            lds_barrier();
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int pos = r0 + POS_PER_IT * i;
                const float2 a = tile[(2 * q) * CS + pos], b = tile[(2 * q + 1) * CS + pos];
                st_stream(dst + (off0 + i * ostep), a.x, a.y, b.x, b.y);
            }
        }
        if (!as_rows) lds_barrier();
    }
}

// Inverse transform of the potential build for grid lengths that are not powers of two (N <= R^2 / 2): every line an inverse
// N-point DFT by chirp-z on the register FFTs, x = conj(w) . IFFT_M(FFT_M(pad(X conj(w))) conj(Bf)) (the inverse of
// rowTB_pass_kernel's chirp-z form: same tables), stored TRANSPOSED through a 16-line tile -- two such passes give ifft2 with
// the data back in its natural layout, the second one with the potential epilogue like ifftT2_kernel.  Replaces the generic
// LDS-resident Bluestein kernel in that role (501^2 x 100 slices: 2 x 500 us -> 2 x ~90 us per frame: for the reference's
// default single-probe runs the inverse transform of the potential WAS the frame).  The buffers are dense (pitch = line count,
// any parity): stores are 8 bytes per line and guarded, a tile row still lands as one 128-byte run.
struct IfftTBJob {
    const float2* in;           // (n_images, n_lines, in_pitch): lines of n_line points in natural frequency order
    float2* out_t;              // transposed store: out_t[img][pos][line]
    float2* out_rows;           // row store (potential epilogue only): out_rows[img][line][pos]
    const float2* tw;           // four-step twiddles of length M = R^2 (ifftTB2_kernel: the 2048-point wave FFT's T table)
    const float2* tw2;          // ifftTB2_kernel: W_64 table
    const float2* bf;           // chirp filter, first half + 1
    const float2* bw;           // chirp, zero beyond n_line
    long long in_is, out_t_is, out_rows_is;
    int in_pitch, out_t_pitch, out_rows_pitch, n_lines, n_line, n_images;
    int potential;              // 1: V = Re(.) * scale, out = exp(i sigma V)
    int rows_parity;            // potential: images whose slice number has this parity are stored as rows, the others transposed; -1: none
    int slice_mod;              // slice number of image img = img % slice_mod (several frames' stacks in one launch); 0: img itself
    int herm;                   // 1: only elements 0 .. n_line/2 of an input line exist, the others are conj(line[n_line - e]) (spectrum of a real image)
    float scale, sigma_over_pi;
};

// PAIR (second pass of the potential build: job.herm and job.potential set): the lines are spectra of REAL lines, so two of them
// ride one complex transform, Z = X_a + i X_b -> V_a + i V_b -- half the transforms of this pass.  A group takes the lines
// g and g + 16 of a 32-line block; both leave through one 32-row tile round (256-byte runs).
template <int R, bool PAIR = false>
__global__ void __launch_bounds__(16 * R, (R == 32) ? 2 : 4) ifftTB_kernel(IfftTBJob job) {
    constexpr int M = R * R, H = R / 2, NH = M / 2, LINES = 16, NT = LINES * R, TCH = 8;
    constexpr int BL = PAIR ? 2 * LINES : LINES;              // lines per work item
    constexpr int CS = R * (R + 1) + 2;
    constexpr int POS_PER_IT = NT / LINES, NIT = NH / POS_PER_IT;       // one thread per (position, line) of a tile row
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // M
    float2* bf = tw + M;                                      // NH + 2
    float2* bw = bf + NH + 2;                                 // NH
    float2* tile = bw + NH;                                   // LINES * CS
    const int tid = threadIdx.x;
    const int N = job.n_line;
    for (int i = tid; i < M; i += NT) tw[i] = job.tw[i];
    for (int i = tid; i <= NH; i += NT) bf[i] = job.bf[i];
    for (int i = tid; i < NH; i += NT) bw[i] = job.bw[i];
    __syncthreads();
    const int grp = tid / R, ln = tid % R;
    const int li = tid % LINES, r0 = tid / LINES;             // store role: line li of the tile, positions r0 + POS_PER_IT * i
    // PAIR: a 32-row tile of the N <= 512 positions (pitch CSP), row 2 g + h = line g + 16 h -- both lines leave in ONE round of
    // 256-byte runs, and a wave's four contiguous rows hold its exchange scratch (ifftTB_two_kernel's layout)
    constexpr int CSP = NH + 2;
    static_assert(!PAIR || (R == 32 && 4 * CSP >= R * 68 / 2), "paired lines: 1024-point transforms only");
    float2* myrow = PAIR ? tile + (2 * grp) * CSP : tile + grp * CS;
    float2* const wtile = PAIR ? tile + (4 * (grp / 2)) * CSP : tile + (grp - grp % (64 / R)) * CS;
    const float* wscr = reinterpret_cast<const float*>(wtile);
    const unsigned wscr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)wtile);
    const float2* fa = bf + ln;
    const float2* fb = bf - ln;
    const int lblocks = (job.n_lines + BL - 1) / BL;
    const int n_items = lblocks * job.n_images;
    // the next item's line is loaded into registers while the current one is transformed
    float2 vn[PAIR ? R : H];
    auto load_line = [&](int it) {
        const int img = it / lblocks, lb = it - img * lblocks;
        const int L = min(lb * BL + grp, job.n_lines - 1);                    // surplus lines of the last block repeat the last one
        const float2* src = job.in + (long long)img * job.in_is + (long long)L * job.in_pitch;
        const int hx = job.herm ? N / 2 : N;
        if constexpr (PAIR) {
            const int Lb = min(lb * BL + LINES + grp, job.n_lines - 1);
            const float2* srcb = job.in + (long long)img * job.in_is + (long long)Lb * job.in_pitch;
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int e = j * R + ln, m = e <= hx ? e : N - e;
                vn[j] = (e < N) ? src[m] : make_float2(0.f, 0.f);
                vn[H + j] = (e < N) ? srcb[m] : make_float2(0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int e = j * R + ln;
                float2 x = (e < N) ? src[e <= hx ? e : N - e] : make_float2(0.f, 0.f);
                if (e > hx) x.y = -x.y;
                vn[j] = x;
            }
        }
    };
    if ((int)blockIdx.x < n_items) load_line(blockIdx.x);
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / lblocks, lb = item - img * lblocks;
        const int L = min(lb * BL + grp, job.n_lines - 1);
        float2 v[R];
        if constexpr (PAIR) {
            const int hx = N / 2;
#pragma unroll
            for (int j = 0; j < H; ++j) {                      // a + i b below the mirror point, conj(a) + i conj(b) above it
                float2 a = vn[j], b = vn[H + j];
                // the Nyquist element of an even length is its own mirror image: only its real part belongs to a real line (the
                // unpaired kernel drops the rest with the imaginary part of its output; here it would land in the partner line)
                if (2 * (j * R + ln) == N) a.y = b.y = 0.f;
                v[j] = (j * R + ln <= hx) ? make_float2(a.x - b.y, a.y + b.x) : make_float2(a.x + b.y, b.x - a.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < H; ++j) v[j] = vn[j];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (item + (int)gridDim.x < n_items) load_line(item + (int)gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        // x conj(w); the chirp table is zero beyond N, the upper half of the registers is padding
        auto mul_chirp = [&]() {
#pragma unroll
            for (int c = 0; c < H; c += TCH) {
                float2 w[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) w[j] = bw[(c + j) * R + ln];
#pragma unroll
                for (int j = 0; j < TCH; ++j) v[c + j] = cmulf_conj(v[c + j], w[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = H; j < R; ++j) v[j] = make_float2(0.f, 0.f);
        };
        mul_chirp();
        fourstep_split_addtid<R, false, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
#pragma unroll
        for (int c = 0; c < R; c += TCH) {
            float2 w[TCH];
#pragma unroll
            for (int j = 0; j < TCH; ++j) w[j] = (c + j < H) ? fa[(c + j) * R] : fb[(R - (c + j)) * R];
#pragma unroll
            for (int j = 0; j < TCH; ++j) v[c + j] = cmulf_conj(v[c + j], w[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        fourstep_split_addtid<R, true, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
        mul_chirp();
        if constexpr (PAIR) {                                  // v[j] = V_a + i V_b: t_a into v[j], t_b into v[H + j]
#pragma unroll
            for (int j = 0; j < H; ++j) {
                if (j % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                float sn, cs, sn2, cs2;
                sincospif(job.sigma_over_pi * (v[j].x * job.scale), &sn, &cs);
                sincospif(job.sigma_over_pi * (v[j].y * job.scale), &sn2, &cs2);
                v[j] = make_float2(cs, sn);
                v[H + j] = make_float2(cs2, sn2);
            }
        } else if (job.potential) {
#pragma unroll
            for (int j = 0; j < H; ++j) {
                if (j % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                float sn, cs;
                sincospif(job.sigma_over_pi * (v[j].x * job.scale), &sn, &cs);
                v[j] = make_float2(cs, sn);
            }
        }
        if (job.potential && (slice_of(job, img) & 1) == job.rows_parity) {       // workgroup-uniform: this slice is kept as rows
            if (lb * BL + grp < job.n_lines) {
                float2* dst = job.out_rows + (long long)img * job.out_rows_is + (long long)L * job.out_rows_pitch;
#pragma unroll
                for (int j = 0; j < H; ++j) if (j * R + ln < N) dst[j * R + ln] = v[j];
            }
            if constexpr (PAIR) {
                if (lb * BL + LINES + grp < job.n_lines) {
                    float2* dst = job.out_rows + (long long)img * job.out_rows_is + (long long)(lb * BL + LINES + grp) * job.out_rows_pitch;
#pragma unroll
                    for (int j = 0; j < H; ++j) if (j * R + ln < N) dst[j * R + ln] = v[H + j];
                }
            }
            continue;
        }
        if constexpr (PAIR) {
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < H; ++j) { myrow[j * R + ln] = v[j]; myrow[CSP + j * R + ln] = v[H + j]; }
            lds_barrier();
            const int l32 = tid % BL;
            int p0 = tid / BL;                                  // (laundered: 32 row offsets of 64 bits would be kept across the loop)
            asm volatile("" : "+v"(p0));
            const int col = lb * BL + l32;
            if (col < job.n_lines) {
                float2* dst = job.out_t + (long long)img * job.out_t_is + col;
                const float2* srow = tile + (2 * (l32 & 15) + (l32 >> 4)) * CSP;
#pragma unroll
                for (int i = 0; i < NH / (NT / BL); ++i) {
                    const int pos = p0 + (NT / BL) * i;
                    if (pos < N) dst[(long long)pos * job.out_t_pitch] = srow[pos];
                }
            }
            lds_barrier();
        } else {
            wave_lds_fence();
#pragma unroll
            for (int j = 0; j < H; ++j) myrow[j * R + ln] = v[j];
            lds_barrier();
            const int col = lb * BL + li;
            if (col < job.n_lines) {
                float2* dst = job.out_t + (long long)img * job.out_t_is + col;
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int pos = r0 + POS_PER_IT * i;
                    if (pos < N) dst[(long long)pos * job.out_t_pitch] = tile[li * CS + pos];
                }
            }
            lds_barrier();
        }
    }
}

// First pass of the potential build (no epilogue) with TWO lines per group and item: ifftTB_kernel spends one tile round -- write,
// barrier, 16 x (LDS read + store), barrier -- per transform, twice the share it has in the slice-loop kernels with their four
// transforms per line, and its vector pipe idles half the time (50 % busy against rowTB_pass_kernel's 82 %, profiles/r03_*).  Here
// a group transforms line g, parks the 16 result registers, transforms line g + 16, and ONE round stores the 32-line tile: 256-byte
// runs, half the barriers per line.  Tile rows hold the N <= 512 positions only (pitch 514); row 2 g + h is line g + 16 h, so that a
// wave's four rows are contiguous and hold its exchange scratch without reaching into another wave's rows.  Line a of the next
// item is prefetched before the first transform, line b once the results are in the tile (it is needed a whole transform later).
__global__ void __launch_bounds__(512, 2) ifftTB_two_kernel(IfftTBJob job) {
    constexpr int R = 32, M = R * R, H = R / 2, NH = M / 2, GROUPS = 16, BL = 32, NT = GROUPS * R, TCH = 8;
    constexpr int CSN = NH + 2;                               // tile row pitch (float2): 2 mod 32, as the 16-line tiles' 1058
    constexpr int POS_PER_IT = NT / BL, NIT = NH / POS_PER_IT;          // one thread per (position, line): 16 positions per sweep
    static_assert(4 * CSN >= R * 68 / 2, "a wave's four tile rows hold its exchange scratch (R rows x 68 floats)");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // M
    float2* bf = tw + M;                                      // NH + 2
    float2* bw = bf + NH + 2;                                 // NH
    float2* tile = bw + NH;                                   // BL * CSN
    const int tid = threadIdx.x;
    const int N = job.n_line;
    for (int i = tid; i < M; i += NT) tw[i] = job.tw[i];
    for (int i = tid; i <= NH; i += NT) bf[i] = job.bf[i];
    for (int i = tid; i < NH; i += NT) bw[i] = job.bw[i];
    __syncthreads();
    const int grp = tid / R, ln = tid % R;
    const int li = tid % BL, r0 = tid / BL;                   // store role: line li of the block, positions r0 + 16 i
    float2* rowa = tile + (2 * grp) * CSN;                    // my two rows: lines grp and grp + 16
    const float* wscr = reinterpret_cast<const float*>(tile + (4 * (grp / 2)) * CSN);
    const unsigned wscr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(tile + (4 * (grp / 2)) * CSN));
    const float2* fa = bf + ln;
    const float2* fb = bf - ln;
    const int lblocks = (job.n_lines + BL - 1) / BL;
    const int n_items = lblocks * job.n_images;
    float2 vna[H], vnb[H];
    auto load_line = [&](int it, int h, float2 (&dst)[H]) {
        const int img = it / lblocks, lb = it - img * lblocks;
        const int L = min(lb * BL + h * GROUPS + grp, job.n_lines - 1);      // surplus lines of the last block repeat the last one
        const float2* src = job.in + (long long)img * job.in_is + (long long)L * job.in_pitch;
        const int hx = job.herm ? N / 2 : N;
        int lnl = ln;                                       // laundered: the 16 element indices are the same for every item, and
        asm volatile("" : "+v"(lnl));                       // the compiler would keep them all in registers across the loop (and spill)
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int e = j * R + lnl;
            float2 x = (e < N) ? src[e <= hx ? e : N - e] : make_float2(0.f, 0.f);
            if (e > hx) x.y = -x.y;
            dst[j] = x;
        }
    };
    if ((int)blockIdx.x < n_items) { load_line(blockIdx.x, 0, vna); load_line(blockIdx.x, 1, vnb); }
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / lblocks, lb = item - img * lblocks;
        // (the last item of a workgroup prefetches itself again: a conditional load would keep the old registers alive as the other
        // arm of the merge)
        const int nitem = item + (int)gridDim.x < n_items ? item + (int)gridDim.x : item;
        float2 v[R];
        auto mul_chirp = [&]() {
#pragma unroll
            for (int c = 0; c < H; c += TCH) {
                float2 w[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) w[j] = bw[(c + j) * R + ln];
#pragma unroll
                for (int j = 0; j < TCH; ++j) v[c + j] = cmulf_conj(v[c + j], w[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = H; j < R; ++j) v[j] = make_float2(0.f, 0.f);
        };
        auto transform = [&]() __attribute__((always_inline)) {
            mul_chirp();
            fourstep_split_addtid<R, false, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
#pragma unroll
            for (int c = 0; c < R; c += TCH) {
                float2 w[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) w[j] = (c + j < H) ? fa[(c + j) * R] : fb[(R - (c + j)) * R];
#pragma unroll
                for (int j = 0; j < TCH; ++j) v[c + j] = cmulf_conj(v[c + j], w[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            fourstep_split_addtid<R, true, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
            mul_chirp();
        };
#pragma unroll
        for (int j = 0; j < H; ++j) v[j] = vna[j];
        __builtin_amdgcn_sched_barrier(0);
        load_line(nitem, 0, vna);
        __builtin_amdgcn_sched_barrier(0);
        transform();
        float2 pa[H];
#pragma unroll
        for (int j = 0; j < H; ++j) { pa[j] = v[j]; v[j] = vnb[j]; }
        __builtin_amdgcn_sched_barrier(0);
        transform();
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < H; ++j) { rowa[j * R + ln] = pa[j]; rowa[CSN + j * R + ln] = v[j]; }
        __builtin_amdgcn_sched_barrier(0);
        load_line(nitem, 1, vnb);                           // (needed a whole transform from now)
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        const int col = lb * BL + li;
        if (col < job.n_lines) {
            float2* dst = job.out_t + (long long)img * job.out_t_is + col;
            const float2* srow = tile + (2 * (li & 15) + (li >> 4)) * CSN;
            int r0v = r0;                                   // (laundered like the lane index above: 32 row offsets of 64 bits)
            asm volatile("" : "+v"(r0v));
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int pos = r0v + POS_PER_IT * i;
                if (pos < N) dst[(long long)pos * job.out_t_pitch] = srow[pos];
            }
        }
        lds_barrier();
    }
}

// The same for lines of 513..1024 points on the 2048-point wave-per-line FFT (tables of rowTB2_pass_kernel): one wave per line,
// 8 lines per workgroup, 64-byte runs in the transposed store.
__global__ void __launch_bounds__(512, 2) ifftTB2_kernel(IfftTBJob job) {
    constexpr int R = 32, M = 2048, H = 16, NH = M / 2, LINES = 8, NT = 512, TCH = 8;
    constexpr int RS = (R * W2K_PITCH) / 2 + 1;
    constexpr int POS_PER_IT = NT / LINES, NIT = NH / POS_PER_IT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // 2048, lane order (lds_pos64)
    float2* w64 = tw + M;                                     // 64
    float2* bf = w64 + 64;                                    // NH + 2
    float2* bw = bf + NH + 2;                                 // NH
    float2* tile = bw + NH;                                   // LINES * RS
    const int tid = threadIdx.x;
    const int N = job.n_line;
    for (int i = tid; i < M; i += NT) tw[lds_pos64(i)] = job.tw[i];
    if (tid < 64) w64[tid] = job.tw2[tid];
    for (int i = tid; i <= NH; i += NT) bf[i] = job.bf[i];
    for (int i = tid; i < NH; i += NT) bw[i] = job.bw[i];
    __syncthreads();
    const int wv = tid >> 6, L = tid & 63, la = lam64(L);
    const float sgn = (L & 1) ? -1.f : 1.f;
    const int li = tid % LINES, r0 = tid / LINES;
    float2* myrow = tile + wv * RS;
    float* scr = reinterpret_cast<float*>(myrow);
    const float2* fa = bf + la;
    const float2* fb = bf - la;
    const int lblocks = (job.n_lines + LINES - 1) / LINES;
    const int n_items = lblocks * job.n_images;
    float2 vn[H];
    auto load_line = [&](int it) {
        const int img = it / lblocks, lb = it - img * lblocks;
        const int Lc = min(lb * LINES + wv, job.n_lines - 1);
        const float2* src = job.in + (long long)img * job.in_is + (long long)Lc * job.in_pitch;
        const int hx = job.herm ? N / 2 : N;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int e = j * 64 + la;
            float2 x = (e < N) ? src[e <= hx ? e : N - e] : make_float2(0.f, 0.f);
            if (e > hx) x.y = -x.y;
            vn[j] = x;
        }
    };
    if ((int)blockIdx.x < n_items) load_line(blockIdx.x);
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int img = item / lblocks, lb = item - img * lblocks;
        const int Lc = min(lb * LINES + wv, job.n_lines - 1);
        float2 v[R];
#pragma unroll
        for (int j = 0; j < H; ++j) v[j] = vn[j];
        __builtin_amdgcn_sched_barrier(0);
        if (item + (int)gridDim.x < n_items) load_line(item + (int)gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        auto mul_chirp = [&]() {
#pragma unroll
            for (int c = 0; c < H; c += TCH) {
                float2 w[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) w[j] = bw[(c + j) * 64 + la];
#pragma unroll
                for (int j = 0; j < TCH; ++j) v[c + j] = cmulf_conj(v[c + j], w[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = H; j < R; ++j) v[j] = make_float2(0.f, 0.f);
        };
        mul_chirp();
        fft2048_wave<false, TCH>(v, scr, tw, w64, L, la, sgn);
#pragma unroll
        for (int c = 0; c < R; c += TCH) {
            float2 w[TCH];
#pragma unroll
            for (int j = 0; j < TCH; ++j) w[j] = (c + j < H) ? fa[(c + j) * 64] : fb[(R - (c + j)) * 64];
#pragma unroll
            for (int j = 0; j < TCH; ++j) v[c + j] = cmulf_conj(v[c + j], w[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        fft2048_wave<true, TCH>(v, scr, tw, w64, L, la, sgn);
        mul_chirp();
        if (job.potential) {
#pragma unroll
            for (int j = 0; j < H; ++j) {
                if (j % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                float sn, cs;
                sincospif(job.sigma_over_pi * (v[j].x * job.scale), &sn, &cs);
                v[j] = make_float2(cs, sn);
            }
        }
        if (job.potential && (slice_of(job, img) & 1) == job.rows_parity) {
            if (lb * LINES + wv < job.n_lines) {
                float2* dst = job.out_rows + (long long)img * job.out_rows_is + (long long)Lc * job.out_rows_pitch;
#pragma unroll
                for (int j = 0; j < H; ++j) if (j * 64 + la < N) dst[j * 64 + la] = v[j];
            }
            continue;
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < H; ++j) myrow[j * 64 + la] = v[j];
        lds_barrier();
        const int col = lb * LINES + li;
        if (col < job.n_lines) {
            float2* dst = job.out_t + (long long)img * job.out_t_is + col;
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int pos = r0 + POS_PER_IT * i;
                if (pos < N) dst[(long long)pos * job.out_t_pitch] = tile[li * RS + pos];
            }
        }
        lds_barrier();
    }
}

// Inverse transform of the potential build for 2048-point axes: one wave per line on fft2048_wave, stored transposed through an
// 8-line tile (64-byte runs); two passes as ifftT2_kernel, the second with the potential epilogue.  (ifftT2_kernel<32>, the 2 R^2
// layout with 64 complex per lane, spills and lost to the generic LDS kernel, which this one replaces.)  job.herm: as ifftTB_kernel.
__global__ void __launch_bounds__(512, 2) ifftTW_kernel(IfftTBJob job) {
    constexpr int R = 32, N = 2048, LINES = 8, NT = 512;
    constexpr int RS = N + 1;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // 2048, lane order
    float2* w64 = tw + N;                                     // 64
    float2* tile = w64 + 64;                                  // LINES * RS
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += NT) tw[lds_pos64(i)] = job.tw[i];
    if (tid < 64) w64[tid] = job.tw2[tid];
    __syncthreads();
    const int wv = tid >> 6, L = tid & 63, la = lam64(L);
    const float sgn = (L & 1) ? -1.f : 1.f;
    const int li = tid % LINES, r0 = tid / LINES;
    float2* myrow = tile + wv * RS;
    float* scr = reinterpret_cast<float*>(myrow);
    const int lblocks = job.n_lines / LINES;
    const int n_items = lblocks * job.n_images;
    float2 vn[R];
    auto load_line = [&](int it) {
        const int img = it / lblocks, lb = it - img * lblocks;
        const float2* src = job.in + (long long)img * job.in_is + (long long)(lb * LINES + wv) * job.in_pitch;
        const int hx = job.herm ? N / 2 : N;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int e = j * 64 + la;
            float2 x = src[e <= hx ? e : N - e];
            if (e > hx) x.y = -x.y;
            vn[j] = x;
        }
    };
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {      // (no register prefetch: 64 + 64 registers plus the transform spill)
        const int img = item / lblocks, lb = item - img * lblocks;
        load_line(item);
        float2 (&v)[R] = vn;
        fft2048_wave<true, 8>(v, scr, tw, w64, L, la, sgn);
        if (job.potential) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (j % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                float sn, cs;
                sincospif(job.sigma_over_pi * (v[j].x * job.scale), &sn, &cs);
                v[j] = make_float2(cs, sn);
            }
        }
        if (job.potential && (slice_of(job, img) & 1) == job.rows_parity) {
            float2* dst = job.out_rows + (long long)img * job.out_rows_is + (long long)(lb * LINES + wv) * job.out_rows_pitch;
#pragma unroll
            for (int j = 0; j < R; ++j) dst[j * 64 + la] = v[j];
            continue;
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < R; ++j) myrow[j * 64 + la] = v[j];
        lds_barrier();
        {
            float2* dst = job.out_t + (long long)img * job.out_t_is + lb * LINES + li;
#pragma unroll
            for (int i = 0; i < N / (NT / LINES); ++i) {
                const int pos = r0 + (NT / LINES) * i;
                dst[(long long)pos * job.out_t_pitch] = tile[li * RS + pos];
            }
        }
        lds_barrier();
    }
}

// out[img][c][r] = in[img][r][c]  (rows x cols -> cols x rows), 32x32 tiles through LDS
__global__ void __launch_bounds__(256) transpose_kernel(const float2* __restrict__ in, float2* __restrict__ out, int rows,
                                                        int cols, int in_pitch, int out_pitch, long long in_is,
                                                        long long out_is) {
    __shared__ float2 t[32][33];
    const float2* src = in + (long long)blockIdx.z * in_is;
    float2* dst = out + (long long)blockIdx.z * out_is;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) t[i][tx] = src[(long long)(r0 + i) * in_pitch + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) dst[(long long)(c0 + i) * out_pitch + r0 + tx] = t[tx][i];
}

}  // namespace msl
