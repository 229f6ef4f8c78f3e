// TACAW time -> frequency transform for ANY number of frames T <= 512 on the register FFTs (tacaw_data.py:94-104):
//     intensity[p, w, k] = | fftshift_t fft_t( Psi[p, :, k] - <Psi[p, :, k]>_t ) |^2
// The reference takes whatever frame count the trajectory holds (05_tacaw.py:24-29; its notebook run has 100 frames,
// example.ipynb:578) -- hardly ever a power of two -- and only T = 256 / 1024 had a register kernel (col_pass_kernel,
// COL_INTENSITY); every other T ran the LDS-resident Stockham / Bluestein kernel at 1.1 TB/s.
//
// A time line has stride npix, so pixels play the role of columns: a workgroup owns a tile of COLS neighbouring pixels x T
// frames (COLS x 8-byte row segments in, COLS x 4-byte segments out), staged column-major into the LDS exactly like
// col_pass_kernel.  One R-lane group per pixel then evaluates the T-point DFT by Bluestein's chirp-z convolution on the
// zero-padded M = R^2 point register layout of the slice-loop kernels (element n = reg R + lane; M >= 2T - 1):
//     X[k] = w[k] . IFFT_M( FFT_M(pad(x w)) . Bf )[k],     w[n] = exp(-i pi n^2 / T),   Bf = FFT_M(conj w, wrapped) / M
// (the tables of rowTB_pass_kernel's chirp-z form: make_cz_tables).  R = 16 (M = 256) serves T <= 128 on 32-pixel tiles, two
// workgroups per CU; R = 32 (M = 1024) serves 129 .. 512 on 16-pixel tiles.
//
// Mean subtraction: subtracting the time mean only changes the u = 0 bin, which it zeroes -- so the kernel zeroes that bin and
// subtracts the line's FIRST sample instead (any constant does): at Bragg pixels, where the mean is orders of magnitude above
// the thermal part, the float32 transform then works on numbers of the size of the result instead of cancelling T large terms.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include "fft_pow2.h"
#include "tacaw_regs.h"

namespace msl {


// column stride (float2) of the LDS tile: the T samples of a line or its share of the wave's exchange scratch (R x 68 floats per
// 64 / R columns = 17 R^2 / 32 float2 per column), whichever is larger, made odd
__host__ __device__ constexpr int tcz_min_stride(int R) { return 17 * R * R / 32; }
__host__ __device__ inline int tcz_stride(int R, int T) { const int s = T > tcz_min_stride(R) ? T : tcz_min_stride(R); return s | 1; }

// VEC: npix is even -- a thread's two pixels are one 16-byte load and one 8-byte store; otherwise (odd grids: the reference's own
// 501 x 491 test grid has 245 991 pixels) they are separate 8- / 4-byte accesses.  Either way npix need not be a multiple of COLS:
// the last tile of an image is ragged and its surplus pixels are neither loaded nor stored (their columns transform garbage).
template <int R, int COLS, bool VEC>
__global__ void __launch_bounds__(COLS * R) time_cz_kernel(TimeJob job) {
    constexpr int M = R * R, H = R / 2, NH = M / 2, NT = COLS * R, TCH = 8;
    // LDS column stride in float2: odd, so that the column-major staging (lanes = pixel pairs: stride 2 CS) and the row reads of
    // the output fall into distinct banks (an even stride put pixel pairs q and q + 8 into the same bank: 31 % of the LDS
    // cycles were conflicts, profiles/r03_tacaw_t100_before_summary.json).  A wave's exchange scratch starts at its first column:
    // 64/R columns = a multiple of 16 bytes for any stride.  The stride follows T (tcz_stride): 137 float2 for T <= 136 on the
    // 256-point kernel, 545 on the 1024-point one.
    const int CS = tcz_stride(R, job.T);
    constexpr int QN = COLS / 2;                      // threads (two pixels = 16 bytes each) per row segment
    constexpr int ROWS_PER_IT = NT / QN;              // 2 R
    constexpr int NIT = NH / ROWS_PER_IT;             // R / 4 staging loads per thread and tile (rows beyond T skipped)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // M
    float2* bf = tw + M;                                      // NH + 2
    float2* bw = bf + NH + 2;                                 // NH
    float2* bufs = bw + NH;                                   // two tiles of COLS * CS
    const int tid = threadIdx.x;
    const int T = job.T;
    for (int i = tid; i < M; i += NT) tw[i] = job.tw[i];
    for (int i = tid; i <= NH; i += NT) bf[i] = job.bf[i];
    for (int i = tid; i < NH; i += NT) bw[i] = job.bw[i];
    const int grp = tid / R, ln = tid % R;            // pixel handled in the transform phase
    const int q = tid % QN, r0 = tid / QN;            // staging role: pixel pair q, frames r0 + ROWS_PER_IT * i
    const float2* fa = bf + ln;                       // Bf[j R + ln],                             j <  R/2
    const float2* fb = bf - ln;                       // Bf[M - (j R + ln)] = bf[(R - j) R - ln],  j >= R/2
    const int tiles_per_image = (job.npix + COLS - 1) / COLS;
    const long long n_tiles = (long long)tiles_per_image * job.n_images;
    const int half = T / 2;                           // np.fft.fftshift: bin u lands at (u + T/2) mod T
    float4 stage[NIT];
    auto load_tile = [&](long long t) {
        const long long p = t / tiles_per_image, c0 = (t % tiles_per_image) * COLS;
        const float2* src = job.in + p * job.image_stride + c0 + 2 * q;
        const int left = job.npix - (int)c0 - 2 * q;        // pixels of the image from this thread's first one on
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int f = r0 + ROWS_PER_IT * i;
            if (f < T) {
                const float2* s = src + (long long)f * job.npix;
                if constexpr (VEC) {
                    if (left > 0) stage[i] = *reinterpret_cast<const float4*>(s);       // npix even: pixels come in pairs
                } else {
                    const float2 a = left > 0 ? s[0] : make_float2(0.f, 0.f), b = left > 1 ? s[1] : make_float2(0.f, 0.f);
                    stage[i] = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        }
    };
    // registers -> LDS, column-major: thread (q, r0) owns the slots (2q, f), (2q + 1, f) of its frames f in BOTH directions -- it
    // is also the thread that reads the intensities of those slots back (store_tile)
    auto stage_tile = [&](float2* cols) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int f = r0 + ROWS_PER_IT * i;
            if (f < T) {
                cols[(2 * q) * CS + f] = make_float2(stage[i].x, stage[i].y);
                cols[(2 * q + 1) * CS + f] = make_float2(stage[i].z, stage[i].w);
            }
        }
    };
    // LDS -> HBM: COLS x 4-byte row segments of the float output (the intensity of bin f sits in the .x of slot f)
    auto store_tile = [&](const float2* cols, long long t) {
        const long long p = t / tiles_per_image, c0 = (t % tiles_per_image) * COLS;
        float* dst = job.out + p * job.image_stride + c0 + 2 * q;
        const int left = job.npix - (int)c0 - 2 * q;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int f = r0 + ROWS_PER_IT * i;
            if (f < T) {
                float* d = dst + (long long)f * job.npix;
                const float a = cols[(2 * q) * CS + f].x, b = cols[(2 * q + 1) * CS + f].x;
                if constexpr (VEC) {
                    if (left > 0) *reinterpret_cast<float2*>(d) = make_float2(a, b);
                } else {
                    if (left > 0) d[0] = a;
                    if (left > 1) d[1] = b;
                }
            }
        }
    };
    // my pixel's time line: subtract the first sample, chirp, FFT_M, filter, IFFT_M; |.|^2 (fftshifted) back into the column
    auto transform_tile = [&](float2* cols) {
        float2* mycol = cols + grp * CS;
        // exchange scratch of the wave (ds_write_addtid_b32 stores): starts at the first column of the wave's 64 / R pixels
        const float* wscr = reinterpret_cast<const float*>(cols + (grp - grp % (64 / R)) * CS);
        const unsigned wscr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(cols + (grp - grp % (64 / R)) * CS));
        float2 v[R];
        const float2 ref = mycol[0];
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int n = j * R + ln;
            const float2 x = (n < T) ? mycol[n] : ref;
            v[j] = make_float2(x.x - ref.x, x.y - ref.y);
        }
        wave_lds_fence();
        {
#pragma unroll
            for (int c = 0; c < H; c += TCH) {
                float2 w[TCH];
#pragma unroll
                for (int j = 0; j < TCH; ++j) if (c + j < H) w[j] = bw[(c + j) * R + ln];
#pragma unroll
                for (int j = 0; j < TCH; ++j) if (c + j < H) v[c + j] = cmulf(v[c + j], w[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = H; j < R; ++j) v[j] = make_float2(0.f, 0.f);
            fourstep_split_addtid<R, false, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
            {
#pragma unroll
                for (int c = 0; c < R; c += TCH) {
                    float2 w[TCH];
#pragma unroll
                    for (int j = 0; j < TCH; ++j) w[j] = (c + j < H) ? fa[(c + j) * R] : fb[(R - (c + j)) * R];
#pragma unroll
                    for (int j = 0; j < TCH; ++j) v[c + j] = cmulf(v[c + j], w[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                fourstep_split_addtid<R, true, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
            }
        }
        // |X[k]|^2 with the chirp's unit modulus dropped (|w[k]| = 1): the last chirp product is not needed for an intensity
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int k = j * R + ln;
            if (k < T) {
                int ks = k + half;
                if (ks >= T) ks -= T;
                mycol[ks].x = (k == 0) ? 0.f : fmaf(v[j].x, v[j].x, v[j].y * v[j].y);
            }
        }
    };
    // Software pipeline over the workgroup's tiles with ONE barrier per tile: while tile i is transformed in one LDS buffer, the
    // intensities of tile i - 1 leave the other one and tile i + 1 is staged into it -- by the same thread slot for slot, so no
    // barrier is needed between the two -- and the loads of tile i + 2 fly in registers.  A wave that has finished its share of the
    // memory work starts its transform while the others are still storing: the phases of a workgroup overlap instead of
    // alternating.  (Measured equal to the three-barrier form, 64 probes x 100 frames x 1024^2: 31.4 ms either way, 16.1 ms
    // with the transform disabled -- the kernel is bound by the rate of its register FFTs, two M-point transforms per line
    // whatever T is: ~1 G 1024-point FFTs per second on the chip, as in the slice-loop kernels.)
    const long long step = gridDim.x;
    long long tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    __syncthreads();
    if (tile < n_tiles) stage_tile(bufs);
    if (tile + step < n_tiles) load_tile(tile + step);
    lds_barrier();
    int it = 0;
    for (; tile < n_tiles; tile += step, ++it) {
        float2* cur = bufs + (it & 1) * (COLS * CS);
        float2* oth = bufs + ((it & 1) ^ 1) * (COLS * CS);
        if (it > 0) store_tile(oth, tile - step);
        if (tile + step < n_tiles) stage_tile(oth);
        if (tile + 2 * step < n_tiles) load_tile(tile + 2 * step);
        transform_tile(cur);
        lds_barrier();
    }
    if (it > 0) store_tile(bufs + ((it - 1) & 1) * (COLS * CS), tile - step);
}


}  // namespace msl
