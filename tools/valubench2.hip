// Which packed-f32 forms are slow on gfx950?  Each kernel runs 16 independent complex accumulators per thread
// through ITERS rounds of one instruction pattern; 2 waves per SIMD (the occupancy of the slice-loop kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define INIT v2f x[16]; for (int i = 0; i < 16; ++i) x[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
#define FINI float s = 0; for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y; out[blockIdx.x * blockDim.x + threadIdx.x] = s;
template <int IT> __global__ void k_pk_add(float* out, float a) { INIT v2f c = v2f{a, -a};
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) x[i] = x[i] + c; } FINI }
template <int IT> __global__ void k_pk_add_vv(float* out, float a) { INIT
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) x[i] = x[i] + x[(i + 1) & 15]; } FINI }
template <int IT> __global__ void k_pk_mul_s(float* out, float a) { INIT v2f c = v2f{a, a};
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) x[i] = x[i] * c; } FINI }
template <int IT> __global__ void k_pk_cmul(float* out, float a, float b) { INIT      // complex multiply by a constant: pk_mul + pk_fma(op_sel)
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) x[i] = x[i] * v2f{a, a} + v2f{x[i].y, x[i].x} * v2f{b, -b}; } FINI }
template <int IT> __global__ void k_sc_cmul(float* out, float a, float b) { INIT      // scalar form: 4 ops
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) { float re = fmaf(x[i].x, a, x[i].y * b), im = fmaf(x[i].y, a, -x[i].x * b); x[i] = v2f{re, im}; } } FINI }
template <int IT> __global__ void k_sc_add(float* out, float a) { INIT
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) { x[i].x += a; x[i].y -= a; } } FINI }
template <int IT> __global__ void k_pk_bfly(float* out) { INIT                            // radix-2 butterflies: a+b, a-b
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { v2f a = x[i], b = x[i + 8]; x[i] = (a + b) * 0.5f; x[i + 8] = (a - b) * 0.5f; } } FINI }
template <int IT> __global__ void k_sc_bfly(float* out) { INIT
    for (int it = 0; it < IT; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { float ax = x[i].x, ay = x[i].y, bx = x[i + 8].x, by = x[i + 8].y;
        x[i].x = (ax + bx) * 0.5f; x[i].y = (ay + by) * 0.5f; x[i + 8].x = (ax - bx) * 0.5f; x[i + 8].y = (ay - by) * 0.5f; } } FINI }
int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    constexpr int IT = 4096;
    const int blocks = 256 * 2, threads = 256;      // 2 waves per SIMD
    auto run = [&](const char* name, auto launch, double instr_per_iter) {
        launch(); (void)hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        double per_simd = (double)IT * instr_per_iter * 2;      // wave-instructions per SIMD (2 waves)
        printf("%-34s %8.3f ms   %.2f ns per (expected) wave-instruction per SIMD\n", name, best, best * 1e6 / per_simd);
    };
    run("pk_add  x + const(sgpr)", [&] { hipLaunchKernelGGL(k_pk_add<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f); }, 16);
    run("pk_add  x + y (vgpr)", [&] { hipLaunchKernelGGL(k_pk_add_vv<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f); }, 16);
    run("pk_mul  x * const(sgpr)", [&] { hipLaunchKernelGGL(k_pk_mul_s<IT>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f); }, 16);
    run("pk cmul (pk_mul + pk_fma op_sel)", [&] { hipLaunchKernelGGL(k_pk_cmul<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.8f, 0.6f); }, 32);
    run("scalar cmul (4 ops)", [&] { hipLaunchKernelGGL(k_sc_cmul<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.8f, 0.6f); }, 64);
    run("scalar add (2 ops)", [&] { hipLaunchKernelGGL(k_sc_add<IT>, dim3(blocks), dim3(threads), 0, 0, out, 0.5f); }, 32);
    run("pk butterfly (add,sub,2 mul)", [&] { hipLaunchKernelGGL(k_pk_bfly<IT>, dim3(blocks), dim3(threads), 0, 0, out); }, 32);
    run("scalar butterfly (8 ops)", [&] { hipLaunchKernelGGL(k_sc_bfly<IT>, dim3(blocks), dim3(threads), 0, 0, out); }, 64);
    return 0;
}
