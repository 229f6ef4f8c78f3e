// Detector binning and streaming TACAW.  The reference keeps every pixel of every exit-wave
// spectrum of every frame (calculators.py:161) and transforms the full array (tacaw_data.py:94-96): BASELINE config C5
// (256 probes x 1024 frames x 2048^2) cannot be represented that way.  Two reductions at the source:
//
//   bin_kernel           sums bx x by neighbouring k pixels of the fftshifted spectrum (coherent sum of the complex
//                        amplitudes: == wavefunction_data.reshape(.., wx/bx, bx, wy/by, by).sum(axes bx, by)) when a frame is
//                        stored -- the resident result shrinks by bx*by;
//   tacaw_fold_kernel    time -> frequency transform accumulated frame tile by frame tile for a chosen set of frequency
//                        bins:  A[p,f,k] += sum_{t in tile} Psi[p,t,k] exp(-2 pi i u_f t / T)   so only a ring of
//                        Tt frames and the F wanted bins are resident, never (P,T,wx,wy).  Subtracting the time mean
//                        (tacaw_data.py:94) only changes the u = 0 bin, which it zeroes, so the accumulation is exact.
//                        Alongside, S1 = sum_t Psi and S2 = sum_t |Psi|^2 give the frequency-integrated pattern
//                        sum_w I[p,w,k] = T S2 - |S1|^2 (Parseval) over ALL T bins without any of them being stored.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msl {

// out[(p * T + slot0 + f) * opitch + X*oy + Y] = sum_{i<bx, j<by} stage[((f*P + p) * wx + X*bx + i) * wy + Y*by + j]
__global__ void __launch_bounds__(256) bin_kernel(const float2* __restrict__ stage, float2* __restrict__ out, int P, int groups,
                                                  int T, int slot0, int wx, int wy, int bx, int by, long long opitch) {
    const int ox = wx / bx, oy = wy / by;
    const long long opix = (long long)ox * oy;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= opix * P * groups) return;
    const int img = (int)(i / opix);                  // f * P + p
    const long long o = i - (long long)img * opix;
    const int X = (int)(o / oy), Y = (int)(o - (long long)X * oy);
    const int f = img / P, p = img - f * P;
    const float2* src = stage + ((long long)img * wx + (long long)X * bx) * wy + (long long)Y * by;
    float sx = 0.f, sy = 0.f;
    for (int a = 0; a < bx; ++a)
        for (int b = 0; b < by; ++b) { const float2 v = src[(long long)a * wy + b]; sx += v.x; sy += v.y; }
    out[((long long)p * T + slot0 + f) * opitch + o] = make_float2(sx, sy);
}

#define MSL_FOLD_FCH 16        // frequency bins accumulated in registers per pass over the frame tile

struct FoldJob {
    const float2* wf;          // (P, ring, wfK): frame slots of the ring, K pixels at a pitch of wfK
    float2* acc;               // (P, F, K) accumulators
    double2* s1;               // (P, K)  sum_t Psi            (updated by the launch with f0 == 0)
    double* s2;                // (P, K)  sum_t |Psi|^2        (float64: T s2 - |s1|^2 cancels to the thermal part)
    const float2* tw;          // (T) exp(-2 pi i m / T)
    const int* bins;           // (F) unshifted FFT bin u_f of every accumulated frequency
    const float2* ref;         // (P, K) reference pattern subtracted from every frame before it is folded, or null
    long long K, wfK;
    int ring, first_slot, count, t0, T, F, f0;
};

// grid (ceil(K/256), P), 256 threads; thread = one k pixel of one probe; this launch handles bins [f0, f0 + FCH)
__global__ void __launch_bounds__(256) tacaw_fold_kernel(FoldJob job) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    const int p = blockIdx.y;
    if (k >= job.K) return;
    const int nf = min(MSL_FOLD_FCH, job.F - job.f0);
    float2 a[MSL_FOLD_FCH];
    float2* accp = job.acc + ((long long)p * job.F + job.f0) * job.K + k;
#pragma unroll
    for (int f = 0; f < MSL_FOLD_FCH; ++f) a[f] = (f < nf) ? accp[(long long)f * job.K] : make_float2(0.f, 0.f);
    const bool sums = (job.f0 == 0);
    double s1x = 0.0, s1y = 0.0, s2 = 0.0;
    const float2* src = job.wf + ((long long)p * job.ring + job.first_slot) * job.wfK + k;
    // Any time-independent offset only changes the u = 0 bin (sum_t exp(-2 pi i u t / T) = 0 otherwise), which the mean
    // subtraction zeroes anyway (tacaw_data.py:94): folding Psi_t - ref keeps the float32 accumulators at the size of the
    // thermal part instead of the Bragg amplitude, whose T terms would have to cancel.  S1 / S2 take the frames as they are.
    const float2 r = job.ref ? job.ref[(long long)p * job.K + k] : make_float2(0.f, 0.f);
    for (int i = 0; i < job.count; ++i) {
        float2 v = src[(long long)i * job.wfK];
        const int t = job.t0 + i;                    // uniform
        if (sums) { s1x += (double)v.x; s1y += (double)v.y; s2 += (double)v.x * v.x + (double)v.y * v.y; }
        v.x -= r.x; v.y -= r.y;
#pragma unroll
        for (int f = 0; f < MSL_FOLD_FCH; ++f) {
            if (f < nf) {
                const int m = (int)(((long long)job.bins[job.f0 + f] * t) % job.T);        // uniform: scalar loads and arithmetic
                const float2 w = job.tw[m];
                a[f].x = fmaf(v.x, w.x, fmaf(-v.y, w.y, a[f].x));
                a[f].y = fmaf(v.x, w.y, fmaf(v.y, w.x, a[f].y));
            }
        }
    }
#pragma unroll
    for (int f = 0; f < MSL_FOLD_FCH; ++f) if (f < nf) accp[(long long)f * job.K] = a[f];
    if (sums) {
        double2* q1 = job.s1 + (long long)p * job.K + k;
        double* q2 = job.s2 + (long long)p * job.K + k;
        const double2 o = *q1;
        *q1 = make_double2(o.x + s1x, o.y + s1y);
        *q2 += s2;
    }
}

// intensity[p,f,k] = |A[p,f,k]|^2 (0 for the u = 0 bin: the time mean is subtracted, tacaw_data.py:94); in place is fine
__global__ void __launch_bounds__(256) tacaw_stream_finish_kernel(const float2* __restrict__ acc, float* __restrict__ inten,
                                                                  const int* __restrict__ bins, long long F, long long K, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long f = (i / K) % F;
    const float2 v = acc[i];
    inten[i] = bins[f] == 0 ? 0.f : fmaf(v.x, v.x, v.y * v.y);
}

// out[p,k] = T * s2 - |s1|^2 = sum over all T frequency bins of |fft_t(Psi - <Psi>)|^2     (float64 on the way out)
__global__ void __launch_bounds__(256) tacaw_stream_total_kernel(const double2* __restrict__ s1, const double* __restrict__ s2, double T,
                                                                 long long total, double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const double2 a = s1[i];
    out[i] = T * s2[i] - (a.x * a.x + a.y * a.y);
}

}  // namespace msl
