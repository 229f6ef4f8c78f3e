// Access-pattern microbenchmark for the slice-loop design (MI355X): linear streaming copy vs
// column-tile copy (W-byte row segments at a row pitch), the pattern of the FFT column pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T>
__global__ void copy_lin(const T* __restrict__ in, T* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// tile = COLS complex columns (COLS*8 bytes per row segment) x nrows; thread moves 16 B
template <int COLS>
__global__ void __launch_bounds__(256) copy_coltile(const float4* __restrict__ in, float4* __restrict__ out, int nrows,
                                                   int pitch /*complex elems*/, int tiles_per_image, size_t image_stride /*complex*/) {
    constexpr int TPR = COLS / 2;                 // threads per row segment (16 B each)
    constexpr int RPI = 256 / TPR;                // rows per iteration
    int img = blockIdx.x / tiles_per_image, tile = blockIdx.x % tiles_per_image;
    size_t base = (size_t)img * image_stride + (size_t)tile * COLS;      // complex index
    int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
#pragma unroll 4
    for (int r = tr; r < nrows; r += RPI) {
        size_t idx = (base + (size_t)r * pitch) / 2 + tc;
        out[idx] = in[idx];
    }
}

// same but with all loads of a tile issued before the stores (deep memory-level parallelism): 1024 rows, 16 cols, 256 thr -> 32 float4/thread
template <int COLS, int NROWS>
__global__ void __launch_bounds__(256) copy_coltile_regs(const float4* __restrict__ in, float4* __restrict__ out,
                                                        int pitch, int tiles_per_image, size_t image_stride) {
    constexpr int TPR = COLS / 2;
    constexpr int RPI = 256 / TPR;
    constexpr int NIT = NROWS / RPI;
    int img = blockIdx.x / tiles_per_image, tile = blockIdx.x % tiles_per_image;
    size_t base = (size_t)img * image_stride + (size_t)tile * COLS;
    int tr = threadIdx.x / TPR, tc = threadIdx.x % TPR;
    float4 v[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) v[i] = in[(base + (size_t)(tr + i * RPI) * pitch) / 2 + tc];
#pragma unroll
    for (int i = 0; i < NIT; ++i) out[(base + (size_t)(tr + i * RPI) * pitch) / 2 + tc] = v[i];
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int n = 1024, P = 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pad : {0, 16, 32}) {
        int pitch = n + pad;
        size_t image = (size_t)n * pitch;           // complex elems
        size_t total = image * P;
        float2 *a, *b;
        CK(hipMalloc(&a, total * 8)); CK(hipMalloc(&b, total * 8));
        CK(hipMemset(a, 1, total * 8)); CK(hipMemset(b, 0, total * 8));
        auto timeit = [&](const char* name, auto launch, double bytes) {
            launch(); (void)hipDeviceSynchronize();
            float best = 1e9, sum = 0; const int reps = 6;
            for (int r = 0; r < reps; ++r) {
                (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
            }
            printf("pitch %4d  %-34s best %8.3f ms  %7.1f GB/s   mean %7.1f GB/s\n", pitch, name, best, bytes / best / 1e6, bytes / (sum / reps) / 1e6);
        };
        double bytes = 2.0 * total * 8;
        if (pad == 0) {
            timeit("linear 16B/lane grid 2048x256", [&] { hipLaunchKernelGGL(copy_lin<float4>, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, total / 2); }, bytes);
            timeit("linear 16B/lane grid 8192x256", [&] { hipLaunchKernelGGL(copy_lin<float4>, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, total / 2); }, bytes);
            timeit("linear 8B/lane grid 8192x256", [&] { hipLaunchKernelGGL(copy_lin<float2>, dim3(8192), dim3(256), 0, 0, (const float2*)a, (float2*)b, total); }, bytes);
            timeit("linear 16B/lane one blk per 8KB row", [&] { hipLaunchKernelGGL(copy_lin<float4>, dim3(P * n / 4), dim3(256), 0, 0, (const float4*)a, (float4*)b, total / 2); }, bytes);
        }
        double tb = 2.0 * (double)n * n * P * 8;
        timeit("coltile  8 cols ( 64B seg)", [&] { hipLaunchKernelGGL(copy_coltile<8>, dim3(P * n / 8), dim3(256), 0, 0, (const float4*)a, (float4*)b, n, pitch, n / 8, image); }, tb);
        timeit("coltile 16 cols (128B seg)", [&] { hipLaunchKernelGGL(copy_coltile<16>, dim3(P * n / 16), dim3(256), 0, 0, (const float4*)a, (float4*)b, n, pitch, n / 16, image); }, tb);
        timeit("coltile 32 cols (256B seg)", [&] { hipLaunchKernelGGL(copy_coltile<32>, dim3(P * n / 32), dim3(256), 0, 0, (const float4*)a, (float4*)b, n, pitch, n / 32, image); }, tb);
        timeit("coltile 64 cols (512B seg)", [&] { hipLaunchKernelGGL(copy_coltile<64>, dim3(P * n / 64), dim3(256), 0, 0, (const float4*)a, (float4*)b, n, pitch, n / 64, image); }, tb);
        timeit("coltile_regs 16 cols all-loads-first", [&] { hipLaunchKernelGGL((copy_coltile_regs<16, 1024>), dim3(P * n / 16), dim3(256), 0, 0, (const float4*)a, (float4*)b, pitch, n / 16, image); }, tb);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
