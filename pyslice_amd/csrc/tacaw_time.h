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
#include "fft_pow2.h"

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

template <int R, int COLS>
__global__ void __launch_bounds__(COLS * R) time_cz_kernel(TimeJob job) {
    constexpr int M = R * R, H = R / 2, NH = M / 2, NT = COLS * R, TCH = 8;
    constexpr int CS = R * (R + 1) + 2;               // LDS column stride in float2: even (16-byte aligned exchange scratch), conflict-free staging
    constexpr int QN = COLS / 2;                      // threads (two pixels = 16 bytes each) per row segment
    constexpr int ROWS_PER_IT = NT / QN;              // 2 R
    constexpr int NIT = NH / ROWS_PER_IT;             // R / 4 staging loads per thread and tile (rows beyond T skipped)
    static_assert(R * 68 * 4 <= (64 / R) * CS * 8, "the wave's add-tid exchange scratch must fit its own columns");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);         // M
    float2* bf = tw + M;                                      // NH + 2
    float2* bw = bf + NH + 2;                                 // NH
    float2* cols = bw + NH;                                   // COLS * CS
    const int tid = threadIdx.x;
    const int T = job.T;
    for (int i = tid; i < M; i += NT) tw[i] = job.tw[i];
    for (int i = tid; i <= NH; i += NT) bf[i] = job.bf[i];
    for (int i = tid; i < NH; i += NT) bw[i] = job.bw[i];
    const int grp = tid / R, ln = tid % R;            // pixel handled in the transform phase
    const int q = tid % QN, r0 = tid / QN;            // staging role: pixel pair q, frames r0 + ROWS_PER_IT * i
    float2* mycol = cols + grp * CS;
    // exchange scratch of the wave (ds_write_addtid_b32 stores): starts at the first column of the wave's 64 / R pixels
    const float* wscr = reinterpret_cast<const float*>(cols + (grp - grp % (64 / R)) * CS);
    const unsigned wscr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(cols + (grp - grp % (64 / R)) * CS));
    const float2* fa = bf + ln;                       // Bf[j R + ln],                             j <  R/2
    const float2* fb = bf - ln;                       // Bf[M - (j R + ln)] = bf[(R - j) R - ln],  j >= R/2
    const int tiles_per_image = job.npix / COLS;
    const long long n_tiles = (long long)tiles_per_image * job.n_images;
    float4 stage[NIT];
    auto load_tile = [&](long long t) {
        const long long p = t / tiles_per_image, c0 = (t % tiles_per_image) * COLS;
        const float2* src = job.in + p * job.image_stride + c0 + 2 * q;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int f = r0 + ROWS_PER_IT * i;
            if (f < T) stage[i] = *reinterpret_cast<const float4*>(src + (long long)f * job.npix);
        }
    };
    long long tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    __syncthreads();
    const int half = T / 2;                           // np.fft.fftshift: bin u lands at (u + T/2) mod T
    for (; tile < n_tiles; tile += gridDim.x) {
        // ---- registers -> LDS, column-major
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int f = r0 + ROWS_PER_IT * i;
            if (f < T) {
                cols[(2 * q) * CS + f] = make_float2(stage[i].x, stage[i].y);
                cols[(2 * q + 1) * CS + f] = make_float2(stage[i].z, stage[i].w);
            }
        }
        lds_barrier();
        // ---- next tile's loads go out now and fly during the transform
        const long long nxt = tile + gridDim.x;
        if (nxt < n_tiles) load_tile(nxt);
        // ---- my pixel's time line: subtract the first sample, chirp, FFT_M, filter, IFFT_M, chirp
        {
            float2 v[R];
            const float2 ref = mycol[0];
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int n = j * R + ln;
                const float2 x = (n < T) ? mycol[n] : ref;
                v[j] = make_float2(x.x - ref.x, x.y - ref.y);
            }
            wave_lds_fence();
            auto mul_chirp = [&]() {
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
            };
            mul_chirp();
            fourstep_split_addtid<R, false, TCH>(v, wscr, wscr_lds, tw, ln, tid & 63);
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
            // |X[k]|^2 with the chirp's unit modulus dropped (|w[k]| = 1): the last chirp product is not needed for an intensity
            wave_lds_fence();
            float* fcol = reinterpret_cast<float*>(mycol);
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const int k = j * R + ln;
                if (k < T) {
                    int ks = k + half;
                    if (ks >= T) ks -= T;
                    fcol[ks] = (k == 0) ? 0.f : fmaf(v[j].x, v[j].x, v[j].y * v[j].y);
                }
            }
        }
        lds_barrier();
        // ---- LDS -> HBM: COLS x 4-byte row segments of the float output
        {
            const long long p = tile / tiles_per_image, c0 = (tile % tiles_per_image) * COLS;
            float* dst = job.out + p * job.image_stride + c0 + 2 * q;
            const float* fa0 = reinterpret_cast<const float*>(cols + (2 * q) * CS);
            const float* fa1 = reinterpret_cast<const float*>(cols + (2 * q + 1) * CS);
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int f = r0 + ROWS_PER_IT * i;
                if (f < T) *reinterpret_cast<float2*>(dst + (long long)f * job.npix) = make_float2(fa0[f], fa1[f]);
            }
        }
        lds_barrier();                          // LDS is free for the next tile's staging from here on
    }
}

}  // namespace msl
