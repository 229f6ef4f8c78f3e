// Stand-alone bench of the transposing slice-loop pass (pyslice_amd/csrc/rowt_pass.h) on BASELINE C3's launch shape (4 frames x
// 64 probes x 1024 lines x 1024 points, one work item of 16 lines x 64 probes per CU), without the library around it: checks
// sampled lines of the natural-order instantiation against a float64 DFT on the host, then times back-to-back launches and
// reports the STEADY state (the chip's clock follows its power draw with a time constant of about a second).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DMSL_CLOCK] [-DMSL_ABL2=bits] [-DMSL_STAGGER=n] [-DBENCH_R=16] \
//         -o tools/bin/rowt_bench tools/rowt_bench.hip
//   tools/bin/rowt_bench [launches [zero]]      zero = 1: all-zero waves and transmission functions (same instructions, less switching)
// -DMSL_CLOCK: in-kernel clock (s_memtime / s_memrealtime around the item loop); -DMSL_ABL2: ablations, see rowt_pass.h.
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../pyslice_amd/csrc/rowt_pass.h"
using namespace msl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define VARIANT 0
#ifndef BENCH_R
#define BENCH_R 32
#endif
#ifndef BENCH_FL
#define BENCH_FL 3
#endif
constexpr int R = BENCH_R, N = R * R;
#define KERNEL(IP, OP) rowT_pass_kernel<R, 16, IP, OP, BENCH_FL>
constexpr int CS_K = (R * R + 33) / 32 * 32 + 2;

typedef std::complex<double> cd;
static void dft(std::vector<cd>& x, bool inv) {          // in-place radix-2, double
    const int n = (int)x.size();
    for (int i = 1, j = 0; i < n; ++i) { int bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(x[i], x[j]); }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = (inv ? 2.0 : -2.0) * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const cd w(cos(ang * k), sin(ang * k));
                const cd u = x[i + k], v = x[i + k + len / 2] * w;
                x[i + k] = u + v; x[i + k + len / 2] = u - v;
            }
    }
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int launches = argc > 1 ? atoi(argv[1]) : 20;
    const bool zero = argc > 2 && atoi(argv[2]) == 1;          // all-zero wave functions and transmission functions: same instructions, less switching
    constexpr int P = 64, F = (R == 32) ? 4 : 64, IMG = P * F, PITCH = N + 16, GRID = (R == 32) ? 256 : 768;      // R = 16: three workgroups per CU
    const size_t img = (size_t)N * PITCH;
    float2 *in, *out, *trans, *pl, *tw;
    CK(hipMalloc(&in, img * IMG * 8)); CK(hipMalloc(&out, img * IMG * 8)); CK(hipMalloc(&trans, (size_t)F * N * N * 8));
    CK(hipMalloc(&pl, N * 8)); CK(hipMalloc(&tw, N * 8));
    std::vector<float2> h(img), t((size_t)F * N * N), tab(N), tww(N);
    for (size_t i = 0; i < img; ++i) h[i] = make_float2((float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f, (float)((i * 40503u) % 1000) * 1e-3f - 0.5f);
    if (zero) std::fill(h.begin(), h.end(), make_float2(0.f, 0.f));
    for (int p = 0; p < IMG; ++p) CK(hipMemcpy(in + p * img, h.data(), img * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < t.size(); ++i) { float a = (float)(i % 977) * 0.01f + (float)(i / ((size_t)N * N)); t[i] = make_float2(cosf(a), sinf(a)); }
    if (zero) std::fill(t.begin(), t.end(), make_float2(0.f, 0.f));
    for (int k = 0; k < N; ++k) { const int kk = k < N / 2 ? k : k - N; double a = -3e-5 * kk * kk; tab[k] = make_float2((float)(cos(a) / N), (float)(sin(a) / N)); }
    for (int k1 = 0; k1 < R; ++k1) for (int n2 = 0; n2 < R; ++n2) { double a = -2.0 * M_PI * ((k1 * n2) % N) / N; tww[k1 * R + n2] = make_float2((float)cos(a), (float)sin(a)); }
    CK(hipMemcpy(trans, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(pl, tab.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tw, tww.data(), N * 8, hipMemcpyHostToDevice));
    RowTJob job{};
    job.in = in; job.out = out; job.trans = trans; job.pl = pl; job.tw = tw; job.tw2 = nullptr;
    job.in_image_stride = job.out_image_stride = (long long)img; job.in_pitch = job.out_pitch = PITCH;
    job.n_lines = N; job.n_images = IMG; job.flags = P2_PRE_A | P2_POST_A; job.pchunk = (R == 32) ? P : 32;
    job.t_group = P; job.t_magic = (unsigned)((1ull << 32) / (unsigned)P + 1); job.t_stride = (long long)N * N;
    job.perm_shift = (R == 32) ? 2 : 1;
    const size_t lds = ((size_t)2 * N + (size_t)16 * CS_K) * 8;
    CK(hipFuncSetAttribute((const void*)KERNEL(false, false), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)KERNEL(true, true), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // ---- correctness: natural order in and out, sampled lines against float64 ----
    CK(hipMemset(out, 0, img * IMG * 8));
    hipLaunchKernelGGL((KERNEL(false, false)), dim3(GRID), dim3(16 * R), lds, 0, job);
    CK(hipDeviceSynchronize());
    {
        const int samples[][2] = {{0, 0}, {0, 1}, {1, 17}, {63, N - 1}, {64, 5}, {130, N / 2 + 3}, {IMG - 1, N - 2}, {200, 31}};
        double num = 0, den = 0, worst = 0;
        std::vector<float2> col(N);
        for (auto& s : samples) {
            const int p = s[0], L = s[1], f = p / P;
            std::vector<cd> x(N);
            for (int n = 0; n < N; ++n) { const float2 v = h[(size_t)L * PITCH + n]; x[n] = cd(v.x, v.y); }
            auto A = [&](std::vector<cd>& y) { dft(y, false); for (int k = 0; k < N; ++k) y[k] *= cd(tab[k].x, tab[k].y); dft(y, true); };
            A(x);
            for (int n = 0; n < N; ++n) { const float2 v = t[(size_t)f * N * N + (size_t)L * N + n]; x[n] *= cd(v.x, v.y); }
            A(x);
            CK(hipMemcpy2D(col.data(), 8, out + (size_t)p * img + L, (size_t)PITCH * 8, 8, N, hipMemcpyDeviceToHost));
            double ln = 0, ld = 0;
            for (int n = 0; n < N; ++n) { const cd d = cd(col[n].x, col[n].y) - x[n]; ln += std::norm(d); ld += std::norm(x[n]); }
            num += ln; den += ld; worst = std::max(worst, sqrt(ln / ld));
        }
        printf("variant %d R %d: natural-order kernel vs float64, %zu lines: rel-L2 %.3e (worst line %.3e)\n", VARIANT, R, sizeof(samples) / sizeof(samples[0]), sqrt(num / den), worst);
    }
    // ---- timing: interleaved order in and out (the passes between two transposing passes) ----
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((KERNEL(true, true)), dim3(GRID), dim3(16 * R), lds, 0, job);
    CK(hipDeviceSynchronize());
    // back-to-back launches in batches of 50 (one event pair per batch: no host gap inside a batch); the chip's clock follows its
    // power draw with a time constant of a second or so, so the steady state is the LAST third of a run of a few seconds
    const int batches = (launches + 49) / 50;
    std::vector<float> bt(batches);
    float best = 1e9f, sum = 0;
    for (int b = 0; b < batches; ++b) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 50; ++r) hipLaunchKernelGGL((KERNEL(true, true)), dim3(GRID), dim3(16 * R), lds, 0, job);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bt[b] = ms / 50; best = bt[b] < best ? bt[b] : best; sum += bt[b];
    }
    float tail = 0; int nt = 0;
    for (int b = batches - (batches + 2) / 3; b < batches; ++b) { tail += bt[b]; ++nt; }
    tail /= nt;
    printf("variant %d R %d: steady state (last third of %d launches) %.1f us per launch = %.3f of 8 TB/s; first batch %.1f, best batch %.1f\n", VARIANT, R, batches * 50, tail * 1e3,
           16.0 * N * N * IMG / (tail * 1e-3) / 8e12, bt[0] * 1e3, best * 1e3);
    sum = sum / batches * launches;
#ifdef MSL_CLOCK
    {   // in-kernel clock: shader cycles / 100 MHz ticks over the item loop, median over the workgroups of the last launch
        unsigned long long* clk; CK(hipMalloc(&clk, GRID * 16)); CK(hipMemset(clk, 0, GRID * 16));
        job.clk = clk;
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((KERNEL(true, true)), dim3(GRID), dim3(16 * R), lds, 0, job);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hc(2 * GRID); CK(hipMemcpy(hc.data(), clk, GRID * 16, hipMemcpyDeviceToHost));
        std::vector<double> ghz(GRID), us(GRID);
        for (int b = 0; b < GRID; ++b) { ghz[b] = (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1; us[b] = hc[2 * b + 1] * 0.01; }
        std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
        printf("variant %d: in-kernel clock median %.3f GHz (min %.3f max %.3f); item loop median %.1f us (min %.1f max %.1f)\n", VARIANT, ghz[GRID / 2], ghz[0], ghz[GRID - 1], us[GRID / 2], us[0], us[GRID - 1]);
        job.clk = nullptr;
    }
#endif
    const double bytes = 16.0 * N * N * IMG;
    printf("variant %d R %d: %d launches of %d images: mean %.1f us, best %.1f us  (%.2f TB/s = %.3f of 8 TB/s at the mean)\n", VARIANT, R, launches, IMG,
           sum / launches * 1e3, best * 1e3, bytes / (sum / launches * 1e-3) / 1e12, bytes / (sum / launches * 1e-3) / 8e12);
    return 0;
}
