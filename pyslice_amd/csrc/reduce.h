// Reductions over the resident results: TACAW intensity (B, F, K) float32 and exit waves (B, T, K) complex64.
// They replace the consumers that follow the hot path in the reference:
//   tacaw_data.py:109-143 spectrum, :145-179 spectrum_image, :256-300 masked_spectrum   -> sum over K (optional mask)
//   tacaw_data.py:183-217 diffraction, :219-254 spectral_diffraction                      -> sum over a (b, f) range
//   tacaw_data.py:302-353 dispersion                                                      -> gather of K indices
//   haadf_data.py:72-94 calculateADF                                                      -> masked sum of |Psi| over K
// All of them stream the array once (HBM bound); sums are accumulated in float64 like the reference's.
// A row of K pixels starts every `ld` elements (ld >= K): the library's own result buffers keep their images at a line-aligned
// pitch (msl_result_pitch), a caller's dense array has ld = K.  Nothing is read beyond the K pixels of a row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msl {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block of 256 threads; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* lds4) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds4[w] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// partial[row * n_chunks + chunk] = sum over the chunk's share of k of weight[k] * src[row*ld + k]   (float64 weights: a
// non-boolean mask of TACAWData.masked_spectrum multiplies the intensity, tacaw_data.py:286-296)
__global__ void __launch_bounds__(256) reduce_kw_kernel(const float* __restrict__ src, const double* __restrict__ weight, long long K,
                                                        long long ld, int n_chunks, double* __restrict__ partial) {
    __shared__ double lds4[4];
    const long long row = blockIdx.y;
    const int chunk = blockIdx.x;
    const long long k0 = K * chunk / n_chunks, k1 = K * (chunk + 1) / n_chunks;
    const float* r = src + row * ld;
    double acc = 0.0;
    for (long long k = k0 + threadIdx.x; k < k1; k += 256) acc += weight[k] * (double)r[k];
    const double tot = block_sum_256(acc, lds4);
    if (threadIdx.x == 0) partial[row * n_chunks + chunk] = tot;
}

// partial[row * n_chunks + chunk] = sum over the chunk's share of k of w(k) * a(row, k)
//   COMPLEX_ABS = false: a = src_f32[row*ld + k]      COMPLEX_ABS = true: a = |src_c64[row*ld + k]|
//   mask (K bytes, zero-padded to a multiple of 16) optional: w = mask[k] != 0
// grid (n_chunks, rows), 256 threads.  Rows that start on 16 bytes (ld % 4 == 0 for floats, ld % 2 == 0 for complex) take
// 16-byte loads over the whole quads of the row; the last chunk adds the K % 4 pixels behind them one by one.
template <bool COMPLEX_ABS>
__global__ void __launch_bounds__(256) reduce_k_kernel(const void* __restrict__ src, const uint8_t* __restrict__ mask, long long K,
                                                       long long ld, int n_chunks, double* __restrict__ partial) {
    __shared__ double lds4[4];
    const long long row = blockIdx.y;
    const int chunk = blockIdx.x;
    // chunk boundaries on multiples of 4 elements
    const long long quads = (K + 3) / 4, whole = K / 4;
    const long long q0 = quads * chunk / n_chunks, q1 = quads * (chunk + 1) / n_chunks;
    const long long v1 = min(q1, whole);              // whole quads of this chunk: [q0, v1)
    const bool tail = (chunk == n_chunks - 1) && (whole * 4 < K);
    double acc = 0.0;
    if constexpr (!COMPLEX_ABS) {
        const float* r = reinterpret_cast<const float*>(src) + row * ld;
        if ((ld & 3) == 0) {
            const float4* r4 = reinterpret_cast<const float4*>(r);
            if (mask) {
                const uchar4* m4 = reinterpret_cast<const uchar4*>(mask);
                for (long long q = q0 + threadIdx.x; q < v1; q += 256) {
                    const float4 v = r4[q];
                    const uchar4 m = m4[q];
                    acc += (double)((m.x ? v.x : 0.f) + (m.y ? v.y : 0.f)) + (double)((m.z ? v.z : 0.f) + (m.w ? v.w : 0.f));
                }
            } else {
                for (long long q = q0 + threadIdx.x; q < v1; q += 256) {
                    const float4 v = r4[q];
                    acc += (double)(v.x + v.y) + (double)(v.z + v.w);
                }
            }
            if (tail) { const long long k = whole * 4 + threadIdx.x; if (k < K && (!mask || mask[k])) acc += (double)r[k]; }
        } else {
            const long long k1 = min(q1 * 4, K);
            for (long long k = q0 * 4 + threadIdx.x; k < k1; k += 256)
                if (!mask || mask[k]) acc += (double)r[k];
        }
    } else {
        const float2* r = reinterpret_cast<const float2*>(src) + row * ld;
        if ((ld & 1) == 0) {
            const float4* r4 = reinterpret_cast<const float4*>(r);
            const long long p0 = q0 * 2, p1 = min(q1 * 2, K / 2);
            for (long long q = p0 + threadIdx.x; q < p1; q += 256) {
                const float4 v = r4[q];
                const float a = (!mask || mask[2 * q]) ? sqrtf(v.x * v.x + v.y * v.y) : 0.f;
                const float b = (!mask || mask[2 * q + 1]) ? sqrtf(v.z * v.z + v.w * v.w) : 0.f;
                acc += (double)a + (double)b;
            }
            if ((K & 1) && chunk == n_chunks - 1 && threadIdx.x == 0 && (!mask || mask[K - 1])) {
                const float2 v = r[K - 1]; acc += (double)sqrtf(v.x * v.x + v.y * v.y);
            }
        } else {
            const long long k1 = min(q1 * 4, K);
            for (long long k = q0 * 4 + threadIdx.x; k < k1; k += 256)
                if (!mask || mask[k]) { const float2 v = r[k]; acc += (double)sqrtf(v.x * v.x + v.y * v.y); }
        }
    }
    const double s = block_sum_256(acc, lds4);
    if (threadIdx.x == 0) partial[row * n_chunks + chunk] = s;
}

// out[k] = scale * sum_{b in [b0,b1)} sum_{f in [f0,f1)} src[(b*F + f)*ld + k]        (float64 accumulation and output)
// thread = 4 consecutive k when K % 4 == 0 and ld % 4 == 0 (VEC), else one k
template <bool VEC>
__global__ void __launch_bounds__(256) reduce_bf_kernel(const float* __restrict__ src, long long F, long long K, long long ld, long long b0,
                                                        long long b1, long long f0, long long f1, double scale,
                                                        double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if constexpr (VEC) {
        if (i * 4 >= K) return;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (long long b = b0; b < b1; ++b) {
            const float4* p = reinterpret_cast<const float4*>(src + (b * F + f0) * ld) + i;
            const long long step = ld / 4;
            long long f = f0;
            for (; f + 4 <= f1; f += 4) {
                const float4 v0 = p[0], v1 = p[step], v2 = p[2 * step], v3 = p[3 * step];
                a0 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
                a1 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
                a2 += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
                a3 += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
                p += 4 * step;
            }
            for (; f < f1; ++f) {
                const float4 v = p[0];
                a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
                p += step;
            }
        }
        double* o = out + i * 4;
        o[0] = a0 * scale; o[1] = a1 * scale; o[2] = a2 * scale; o[3] = a3 * scale;
    } else {
        if (i >= K) return;
        double a = 0;
        for (long long b = b0; b < b1; ++b)
            for (long long f = f0; f < f1; ++f) a += (double)src[(b * F + f) * ld + i];
        out[i] = a * scale;
    }
}

// out[row * n + i] = src[row * ld + idx[i]]   (rows = B * F)
__global__ void __launch_bounds__(256) gather_k_kernel(const float* __restrict__ src, long long rows, long long ld,
                                                       const long long* __restrict__ idx, long long n, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * n) return;
    const long long row = t / n, i = t % n;
    out[t] = src[row * ld + idx[i]];
}

// complex64 -> complex128 (msl_download_wavefunction_c128): n dense output elements; input element i of the run of rows of K
// pixels at a pitch of ld pixels that starts at src
__global__ void __launch_bounds__(256) widen_c64_kernel(const float2* __restrict__ src, double2* __restrict__ dst, long long n,
                                                        long long K, long long ld) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long row = i / K;
        const float2 v = src[row * ld + (i - row * K)];
        dst[i] = make_double2((double)v.x, (double)v.y);
    }
}

}  // namespace msl
