// Issue cost of single VALU instructions on gfx950, controlled with inline asm (no compiler folding):
// 16 independent accumulators per lane, W waves per SIMD.  Prints ns and (at the measured clock) cycles per
// wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define BODY(INSTR, CONSTRAINT_C)                                                                         \
    v2f x[16];                                                                                            \
    for (int i = 0; i < 16; ++i) x[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};        \
    v2f c = v2f{a, b};                                                                                    \
    for (int it = 0; it < IT; ++it) {                                                                     \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(INSTR : "+v"(x[i]) : CONSTRAINT_C(c)); \
    }                                                                                                     \
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
template <int IT> __global__ void k_pk_add(float* out, float a, float b) { BODY("v_pk_add_f32 %0, %0, %1", "v") }
template <int IT> __global__ void k_pk_mul(float* out, float a, float b) { BODY("v_pk_mul_f32 %0, %0, %1", "v") }
template <int IT> __global__ void k_pk_fma(float* out, float a, float b) { BODY("v_pk_fma_f32 %0, %0, %1, %1", "v") }
template <int IT> __global__ void k_pk_fma_sel(float* out, float a, float b) { BODY("v_pk_fma_f32 %0, %0, %1, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]", "v") }
template <int IT> __global__ void k_pk_add_s(float* out, float a, float b) { BODY("v_pk_add_f32 %0, %0, %1", "s") }
template <int IT> __global__ void k_add_lo(float* out, float a, float b) {
    float x[16]; for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)); }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = s; }
template <int IT> __global__ void k_fma(float* out, float a, float b) {
    float x[16]; for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); }
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = s; }
// dependent chain: one accumulator
template <int IT> __global__ void k_pk_fma_dep(float* out, float a, float b) {
    v2f x = v2f{threadIdx.x * 0.001f, 1.f}, c = v2f{a, b};
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.x + x.y; }
template <int IT> __global__ void k_fma_dep(float* out, float a, float b) {
    float x = threadIdx.x * 0.001f;
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x; }
int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int IT = 8192;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps, threads = 256;      // wps waves per SIMD
        auto run = [&](const char* name, auto launch) {
            launch(); (void)hipDeviceSynchronize();
            float best = 1e9;
            for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
            const double per_simd = (double)IT * 16 * wps;
            printf("waves/SIMD %d  %-28s %8.3f ms  %6.2f ns/instr/SIMD  (%.2f clk at 2.4 GHz)\n", wps, name, best, best * 1e6 / per_simd, best * 1e6 / per_simd * 2.4);
        };
        run("v_add_f32", [&] { hipLaunchKernelGGL(k_add_lo<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f, 0.25f); });
        run("v_fma_f32", [&] { hipLaunchKernelGGL(k_fma<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.25f); });
        run("v_pk_add_f32 (vgpr)", [&] { hipLaunchKernelGGL(k_pk_add<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f, 0.25f); });
        run("v_pk_add_f32 (sgpr pair)", [&] { hipLaunchKernelGGL(k_pk_add_s<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f, 0.25f); });
        run("v_pk_mul_f32", [&] { hipLaunchKernelGGL(k_pk_mul<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 1.001f); });
        run("v_pk_fma_f32", [&] { hipLaunchKernelGGL(k_pk_fma<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.25f); });
        run("v_pk_fma_f32 op_sel", [&] { hipLaunchKernelGGL(k_pk_fma_sel<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f, 0.25f); });
        run("v_fma_f32 dependent chain", [&] { hipLaunchKernelGGL(k_fma_dep<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.25f); });
        run("v_pk_fma_f32 dependent chain", [&] { hipLaunchKernelGGL(k_pk_fma_dep<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.999f, 0.25f); });
    }
    return 0;
}
