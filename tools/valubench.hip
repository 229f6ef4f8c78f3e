// VALU issue-rate microbenchmark (MI355X): independent v_fma_f32 vs v_pk_fma_f32 chains at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int ITERS>
__global__ void k_scalar(float* out, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ITERS>
__global__ void k_packed(float* out, float a, float b) {
    v2f x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    const v2f va = v2f{a, a}, vb = v2f{b, b};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * va + vb;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ITERS>
__global__ void k_add(float* out, float a) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = x[i] + a;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* out; (void)hipMalloc(&out, 256 * 2048 * 4 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int IT = 8192;
    for (int wps : {1, 2, 4, 8}) {                    // waves per SIMD
        int threads = 256;                           // 4 waves per block -> 1 per SIMD per block
        int blocks = 256 * wps;
        auto run = [&](const char* name, auto launch, double flops_per_thread) {
            launch(); (void)hipDeviceSynchronize();
            float best = 1e9;
            for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
            double total = flops_per_thread * threads * blocks;
            double instr_per_simd = (double)IT * 16 * wps;       // wave-instructions issued per SIMD (scalar kernels)
            printf("waves/SIMD %d  %-10s %8.3f ms  %7.1f TFLOP/s   (%.2f ns per wave-instr per SIMD)\n", wps, name, best, total / best / 1e9, best * 1e6 / instr_per_simd);
        };
        run("fma", [&] { hipLaunchKernelGGL(k_scalar<IT>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); }, 2.0 * 16 * IT);
        run("pk_fma", [&] { hipLaunchKernelGGL(k_packed<IT>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); }, 2.0 * 16 * IT);
        run("add", [&] { hipLaunchKernelGGL(k_add<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f); }, 1.0 * 16 * IT);
    }
    return 0;
}
