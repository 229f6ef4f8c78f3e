// Where does an iteration of the transposing pass spend its time?  Diagnostic build of rowT_pass_kernel<32,16>
// (-DMSL_STAMPS: s_memtime at the phase boundaries of every iteration, accumulated per wave) on BASELINE C3's shape:
// 64 probes x 1024 lines x 1024 points, chunks of 16 probes, 256 workgroups.  Prints cycles per phase and wave-iteration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DMSL_STAMPS -o tools/bin/rowt_timeline tools/rowt_timeline.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../pyslice_amd/csrc/fft_pow2.h"
using namespace msl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    constexpr int R = 32, N = 1024, P = 64, PITCH = N + 16, GRID = 256;
    const size_t img = (size_t)N * PITCH;
    float2 *in, *out, *trans, *pl, *tw; unsigned* stamps;
    CK(hipMalloc(&in, img * P * 8)); CK(hipMalloc(&out, img * P * 8)); CK(hipMalloc(&trans, (size_t)N * N * 8));
    CK(hipMalloc(&pl, N * 8)); CK(hipMalloc(&tw, N * 8)); CK(hipMalloc(&stamps, (size_t)GRID * 8 * MSL_NSTAMP * 4));
    std::vector<float2> h(img), t((size_t)N * N), tab(N), tww(N);
    for (size_t i = 0; i < img; ++i) h[i] = make_float2((float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f, (float)((i * 40503u) % 1000) * 1e-3f - 0.5f);
    for (int p = 0; p < P; ++p) CK(hipMemcpy(in + p * img, h.data(), img * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < t.size(); ++i) { float a = (float)(i % 977) * 0.01f; t[i] = make_float2(cosf(a), sinf(a)); }
    for (int k = 0; k < N; ++k) { double a = -1e-4 * k * k; tab[k] = make_float2((float)(cos(a) / N), (float)(sin(a) / N)); }
    for (int k1 = 0; k1 < R; ++k1) for (int n2 = 0; n2 < R; ++n2) { double a = -2.0 * M_PI * ((k1 * n2) % N) / N; tww[k1 * R + n2] = make_float2((float)cos(a), (float)sin(a)); }
    CK(hipMemcpy(trans, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(pl, tab.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tw, tww.data(), N * 8, hipMemcpyHostToDevice));
    RowTJob job{};
    job.in = in; job.out = out; job.trans = trans; job.pl = pl; job.tw = tw; job.tw2 = nullptr;
    job.in_image_stride = job.out_image_stride = (long long)img; job.in_pitch = job.out_pitch = PITCH;
    job.n_lines = N; job.n_images = P; job.flags = P2_PRE_A | P2_POST_A; job.pchunk = 16; job.stamps = stamps;
    constexpr int CS = (R * R + 33) / 32 * 32 + 2;
    const size_t lds = ((size_t)2 * N + (size_t)16 * CS) * 8;
    CK(hipFuncSetAttribute((const void*)rowT_pass_kernel<32, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
        CK(hipMemset(stamps, 0, (size_t)GRID * 8 * MSL_NSTAMP * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((rowT_pass_kernel<32, 16>), dim3(GRID), dim3(512), lds, 0, job);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    std::vector<unsigned> s((size_t)GRID * 8 * MSL_NSTAMP);
    CK(hipMemcpy(s.data(), stamps, s.size() * 4, hipMemcpyDeviceToHost));
    const char* names[MSL_NSTAMP] = {"loop top (cursor)", "wait for the prefetched line, v = vn", "t_k load (1 of 16 iterations) + cursor",
        "T1 head: register FFT + twiddles", "T1 tail: transpose + register FFT", "prefetch quarter 1 (issue)", "x P (LDS table)",
        "T2 head", "T2 tail", "prefetch quarter 2 (issue)", "x t_k (registers)", "T3 head", "T3 tail", "prefetch quarter 3 (issue)",
        "x P (LDS table)", "T4 head", "T4 tail", "prefetch quarter 4 (issue)", "write the line into the tile", "barrier 1 (tile complete)",
        "store phase: tile -> 128-byte segments", "barrier 2 (tile free)", "", ""};
    const double iters = 16.0;     // 64 line blocks x 4 probe chunks x 16 probes / 256 workgroups
    double tot = 0; std::vector<double> mean(MSL_NSTAMP, 0.0), early(MSL_NSTAMP, 0.0), lateh(MSL_NSTAMP, 0.0);
    for (int w = 0; w < GRID * 8; ++w) for (int i = 0; i < MSL_NSTAMP; ++i) {
        mean[i] += s[(size_t)w * MSL_NSTAMP + i];
        ((w % 8) < 4 ? early : lateh)[i] += s[(size_t)w * MSL_NSTAMP + i];
    }
    for (int i = 0; i < MSL_NSTAMP; ++i) { mean[i] /= GRID * 8 * iters; early[i] /= GRID * 4 * iters; lateh[i] /= GRID * 4 * iters; tot += mean[i]; }
    printf("stamped kernel: %.1f us per pass (shipped kernel: see bench.py); s_memtime ticks = shader cycles; %.0f cycles per wave-iteration\n", best * 1e3, tot);
    printf("%-46s %10s %7s   %10s %10s\n", "phase", "cycles", "share", "waves 0-3", "waves 4-7");
    for (int i = 0; i < 22; ++i) printf("%-46s %10.0f %6.1f%%   %10.0f %10.0f\n", names[i], mean[i], 100.0 * mean[i] / tot, early[i], lateh[i]);
    printf("effective clock: %.2f GHz (cycles per wave-iteration x 16 iterations / kernel time)\n", tot * iters / (best * 1e6));
    return 0;
}
