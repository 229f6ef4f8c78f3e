// Stand-alone check and bench of the mixed-radix transposing pass (pyslice_amd/csrc/rowtm_pass.h) on a square grid of N = A * B
// points: sampled lines against a float64 DFT on the host, then back-to-back launches (steady state: last third).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DBENCH_A=24 -DBENCH_B=25 -o tools/bin/rowtm_bench tools/rowtm_bench.hip
//   tools/bin/rowtm_bench [launches [pchunk]]
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../pyslice_amd/csrc/rowtm_pass.h"
using namespace msl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#ifndef BENCH_A
#define BENCH_A 24
#endif
#ifndef BENCH_B
#define BENCH_B 25
#endif
#ifndef BENCH_G
#define BENCH_G 32
#endif
#ifdef BENCH_TWO
constexpr int A = BENCH_A, B = BENCH_B, G = 64, N = 2 * A * B, TLINES = 8;
#define KERNEL rowTM2_pass_kernel<A, B>
#else
constexpr int A = BENCH_A, B = BENCH_B, G = BENCH_G, N = A * B, TLINES = 16;
#define KERNEL rowTM_pass_kernel<A, B, G>
#endif
typedef std::complex<double> cd;

static void dft(std::vector<cd>& x, bool inv) {          // O(N^2), double
    const int n = (int)x.size();
    std::vector<cd> w(n), y(n);
    for (int k = 0; k < n; ++k) { const double a = (inv ? 2.0 : -2.0) * M_PI * k / n; w[k] = cd(cos(a), sin(a)); }
    for (int k = 0; k < n; ++k) { cd s = 0; for (int j = 0; j < n; ++j) s += x[j] * w[(int)(((long long)j * k) % n)]; y[k] = s; }
    x = y;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const int pchunk = argc > 2 ? atoi(argv[2]) : 8;
    constexpr int P = 64, F = 4, IMG = P * F, PITCH = N + 16;
    const size_t img = (size_t)N * PITCH;
    float2 *in, *out, *trans, *pl, *tw;
    CK(hipMalloc(&in, img * IMG * 8)); CK(hipMalloc(&out, img * IMG * 8)); CK(hipMalloc(&trans, (size_t)F * N * N * 8));
    CK(hipMalloc(&pl, N * 8)); CK(hipMalloc(&tw, (2 * N + 2 * A) * 8));
    std::vector<float2> h(img), t((size_t)F * N * N), tab(N), tww(2 * N + 2 * A);
    for (size_t i = 0; i < img; ++i) h[i] = make_float2((float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f, (float)((i * 40503u) % 1000) * 1e-3f - 0.5f);
    for (int p = 0; p < IMG; ++p) CK(hipMemcpy(in + p * img, h.data(), img * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < t.size(); ++i) { float a = (float)(i % 977) * 0.01f + (float)(i / ((size_t)N * N)); t[i] = make_float2(cosf(a), sinf(a)); }
    for (int k = 0; k < N; ++k) { const int kk = k < (N + 1) / 2 ? k : k - N; double a = -3e-5 * kk * kk; tab[k] = make_float2((float)(cos(a) / N), (float)(sin(a) / N)); }
#ifdef BENCH_TWO
    for (int k2 = 0; k2 < B; ++k2) for (int n1 = 0; n1 < 2 * A; ++n1) {
        const double a = -2.0 * M_PI * ((k2 * n1) % N) / N;
        tww[k2 * 2 * A + n1] = make_float2((float)cos(a), (float)sin(a));
        tww[N + (n1 % A) * 2 * B + 2 * k2 + n1 / A] = make_float2((float)cos(a), (float)sin(a));
    }
    for (int m = 0; m < A; ++m) { const double a = -2.0 * M_PI * m / (2 * A); tww[2 * N + m] = make_float2(1.f, 0.f); tww[2 * N + A + m] = make_float2((float)cos(a), (float)sin(a)); }
#else
    for (int k2 = 0; k2 < B; ++k2) for (int n1 = 0; n1 < A; ++n1) {
        const double a = -2.0 * M_PI * ((k2 * n1) % N) / N;
        tww[k2 * A + n1] = make_float2((float)cos(a), (float)sin(a));
        tww[N + n1 * B + k2] = make_float2((float)cos(a), (float)sin(a));
    }
#endif
    CK(hipMemcpy(trans, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(pl, tab.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tw, tww.data(), tww.size() * 8, hipMemcpyHostToDevice));
    RowTJob job{};
    job.in = in; job.out = out; job.trans = trans; job.pl = pl; job.tw = tw; job.tw2 = nullptr;
    job.in_image_stride = job.out_image_stride = (long long)img; job.in_pitch = job.out_pitch = PITCH;
    job.n_lines = N; job.n_images = IMG; job.flags = P2_PRE_A | P2_POST_A; job.pchunk = pchunk;
    job.t_group = P; job.t_magic = (unsigned)((1ull << 32) / (unsigned)P + 1); job.t_stride = (long long)N * N;
#ifdef BENCH_TWO
    const size_t lds = rowTM2_lds_bytes(A, B);
#else
    const size_t lds = rowTM_lds_bytes(A, B);
#endif
    const int lblocks = (N + TLINES - 1) / TLINES, GRID = std::min(256, lblocks * (IMG / pchunk));
    CK(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipMemset(out, 0, img * IMG * 8));
    hipLaunchKernelGGL((KERNEL), dim3(GRID), dim3(TLINES * G), lds, 0, job);
    CK(hipDeviceSynchronize());
    {
        const int samples[][2] = {{0, 0}, {0, 1}, {1, 17}, {63, N - 1}, {64, 5}, {130, N / 2 + 3}, {IMG - 1, N - 2}, {200, 31}, {77, N - 9}};
        double num = 0, den = 0, worst = 0;
        std::vector<float2> col(N);
        for (auto& s : samples) {
            const int p = s[0], L = s[1], f = p / P;
            std::vector<cd> x(N);
            for (int n = 0; n < N; ++n) { const float2 v = h[(size_t)L * PITCH + n]; x[n] = cd(v.x, v.y); }
            auto Aop = [&](std::vector<cd>& y) { dft(y, false); for (int k = 0; k < N; ++k) y[k] *= cd(tab[k].x, tab[k].y); dft(y, true); };
            Aop(x);
            for (int n = 0; n < N; ++n) { const float2 v = t[(size_t)f * N * N + (size_t)L * N + n]; x[n] *= cd(v.x, v.y); }
            Aop(x);
            CK(hipMemcpy2D(col.data(), 8, out + (size_t)p * img + L, (size_t)PITCH * 8, 8, N, hipMemcpyDeviceToHost));
            double ln = 0, ld = 0;
            for (int n = 0; n < N; ++n) { const cd d = cd(col[n].x, col[n].y) - x[n]; ln += std::norm(d); ld += std::norm(x[n]); }
            num += ln; den += ld; worst = std::max(worst, sqrt(ln / ld));
        }
        printf("N = (2x) %d x %d = %d (G %d, LDS %zu B): vs float64, %zu lines: rel-L2 %.3e (worst line %.3e)\n", A, B, N, G, lds, sizeof(samples) / sizeof(samples[0]), sqrt(num / den), worst);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int batches = (launches + 49) / 50;
    std::vector<float> bt(batches);
    for (int b = 0; b < batches; ++b) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 50; ++r) hipLaunchKernelGGL((KERNEL), dim3(GRID), dim3(TLINES * G), lds, 0, job);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bt[b] = ms / 50;
    }
    float tail = 0; int nt = 0;
    for (int b = batches - (batches + 2) / 3; b < batches; ++b) { tail += bt[b]; ++nt; }
    tail /= nt;
    printf("N = %d pchunk %d grid %d: steady state %.1f us per launch of %d images = %.3f of 8 TB/s; %.0f k slice-steps/s for the loop alone; first batch %.1f us\n", N, pchunk, GRID, tail * 1e3, IMG,
           16.0 * N * N * IMG / (tail * 1e-3) / 8e12, IMG / (tail * 1e-3) / 1e3, bt[0] * 1e3);
    return 0;
}
