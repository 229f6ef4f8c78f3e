// Stand-alone bench of the 2048-point transposing pass (rowTW_pass_kernel, BASELINE C5's grid) on bench.py's launch shape for
// `--grid 2048 --probes 16` (4 frames x 16 probes x 2048 lines, items of 8 lines x 16 probes): float64 check of the natural-order
// instantiation, steady-state timing of the paired-layout one, in-kernel clock (-DMSL_CLOCK), ablations (-DMSL_ABL2=bits, see
// rowt_pass.h).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DMSL_CLOCK] [-DMSL_ABL2=n] -o tools/bin/rowtw_bench tools/rowtw_bench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pyslice_amd/csrc/fft_pow2.h"
using namespace msl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef std::complex<double> cd;
static void dft(std::vector<cd>& x, bool inv) {
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
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    constexpr int N = 2048, P = 16, F = 4, IMG = P * F, PITCH = N + 16, GRID = 256;
    const size_t img = (size_t)N * PITCH;
    float2 *in, *out, *trans, *pl, *tw, *w64;
    CK(hipMalloc(&in, img * IMG * 8)); CK(hipMalloc(&out, img * IMG * 8)); CK(hipMalloc(&trans, (size_t)F * N * N * 8));
    CK(hipMalloc(&pl, (N / 2 + 1) * 8)); CK(hipMalloc(&tw, N * 8)); CK(hipMalloc(&w64, 64 * 8));
    std::vector<float2> h(img), t((size_t)F * N * N), tab(N), tww(N), ww(64);
    for (size_t i = 0; i < img; ++i) h[i] = make_float2((float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f, (float)((i * 40503u) % 1000) * 1e-3f - 0.5f);
    for (int p = 0; p < IMG; ++p) CK(hipMemcpy(in + p * img, h.data(), img * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < t.size(); ++i) { float a = (float)(i % 977) * 0.01f + (float)(i / ((size_t)N * N)); t[i] = make_float2(cosf(a), sinf(a)); }
    for (int k = 0; k < N; ++k) { const int kk = k < N / 2 ? k : k - N; double a = -1e-5 * kk * kk; tab[k] = make_float2((float)(cos(a) / N), (float)(sin(a) / N)); }
    for (int k1 = 0; k1 < 32; ++k1) for (int n2 = 0; n2 < 64; ++n2) { double a = -2.0 * M_PI * (k1 * n2) / 2048.0; tww[k1 * 64 + n2] = make_float2((float)cos(a), (float)sin(a)); }
    for (int m = 0; m < 32; ++m) { double a = -2.0 * M_PI * m / 64.0; ww[m] = make_float2(1.f, 0.f); ww[32 + m] = make_float2((float)cos(a), (float)sin(a)); }
    CK(hipMemcpy(trans, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(pl, tab.data(), (N / 2 + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tw, tww.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(w64, ww.data(), 64 * 8, hipMemcpyHostToDevice));
    RowTJob job{};
    job.in = in; job.out = out; job.trans = trans; job.pl = pl; job.tw = tw; job.tw2 = w64;
    job.in_image_stride = job.out_image_stride = (long long)img; job.in_pitch = job.out_pitch = PITCH;
    job.n_lines = N; job.n_images = IMG; job.flags = P2_PRE_A | P2_POST_A; job.pchunk = P;
    job.t_group = P; job.t_magic = (unsigned)((1ull << 32) / (unsigned)P + 1); job.t_stride = (long long)N * N;
    const size_t lds = ((size_t)N + 64 + N / 2 + 64 + (size_t)8 * (N + 1)) * 8;
    CK(hipFuncSetAttribute((const void*)rowTW_pass_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)rowTW_pass_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipMemset(out, 0, img * IMG * 8));
    hipLaunchKernelGGL((rowTW_pass_kernel<false, false>), dim3(GRID), dim3(512), lds, 0, job);
    CK(hipDeviceSynchronize());
    {
        const int samples[][2] = {{0, 0}, {0, 1}, {1, 17}, {15, N - 1}, {16, 5}, {40, N / 2 + 3}, {IMG - 1, N - 2}, {33, 31}};
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
        printf("2048: natural-order kernel vs float64, 8 lines: rel-L2 %.3e (worst line %.3e)\n", sqrt(num / den), worst);
    }
    if (argc > 2 && atoi(argv[2]) == 1) { job.in_pitch = 0; job.in_image_stride = 0; printf("2048: every line reads the SAME 16 KB (cache hits)\n"); }
    if (argc > 2 && atoi(argv[2]) == 2) { job.out_pitch = 0; job.out_image_stride = 0; printf("2048: every store goes to the same 128 KB\n"); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rowTW_pass_kernel<true, true>), dim3(GRID), dim3(512), lds, 0, job);
    CK(hipDeviceSynchronize());
    const int batches = (launches + 24) / 25;
    std::vector<float> bt(batches);
    for (int b = 0; b < batches; ++b) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 25; ++r) hipLaunchKernelGGL((rowTW_pass_kernel<true, true>), dim3(GRID), dim3(512), lds, 0, job);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bt[b] = ms / 25;
    }
    float tail = 0; int nt = 0;
    for (int b = batches - (batches + 2) / 3; b < batches; ++b) { tail += bt[b]; ++nt; }
    tail /= nt;
    printf("2048: steady state (last third of %d launches) %.1f us per launch of %d images = %.3f of 8 TB/s; first batch %.1f\n", batches * 25, tail * 1e3, IMG,
           16.0 * N * N * IMG / (tail * 1e-3) / 8e12, bt[0] * 1e3);
#ifdef MSL_CLOCK
    {
        unsigned long long* clk; CK(hipMalloc(&clk, GRID * 16)); CK(hipMemset(clk, 0, GRID * 16));
        job.clk = clk;
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((rowTW_pass_kernel<true, true>), dim3(GRID), dim3(512), lds, 0, job);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hc(2 * GRID); CK(hipMemcpy(hc.data(), clk, GRID * 16, hipMemcpyDeviceToHost));
        std::vector<double> ghz(GRID), us(GRID), cyc(GRID);
        for (int b = 0; b < GRID; ++b) { ghz[b] = (double)hc[2 * b] / (double)hc[2 * b + 1] * 0.1; us[b] = hc[2 * b + 1] * 0.01; cyc[b] = (double)hc[2 * b]; }
        std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end()); std::sort(cyc.begin(), cyc.end());
        printf("2048: in-kernel clock median %.3f GHz; item loop median %.1f us (min %.1f max %.1f); %.3f M cycles = %.0f per iteration of 8 lines\n", ghz[GRID / 2], us[GRID / 2], us[0], us[GRID - 1], cyc[GRID / 2] / 1e6, cyc[GRID / 2] / 64);
    }
#endif
    return 0;
}
