// Instantiation lists of the mixed-radix transposing passes (rowtm_pass.h), split over six translation units (slice_mixed_a.hip:
// lengths A * B below 500, slice_mixed_b.hip: from 500, slice_mixed_c.hip: lengths 2 A * B on one wave per line, G = 64; _d / _e / _f: the
// same three groups for lengths with a factor 7) so that
// the library builds in parallel.  Every entry is (A, B, G): line length
// A * B in groups of G lanes.  Balanced factors keep most lanes busy in both layouts; A >= B: the prefetched line and the t_k line
// are B registers each; G = 16 where both factors allow it.
#pragma once
#include "rowtm_pass.h"

#define MSL_ROWTM_LIST_A(X) \
    X(15, 9, 16) X(12, 12, 16) X(15, 10, 16) X(16, 10, 16) X(15, 12, 16) X(16, 12, 16) X(20, 10, 32) X(18, 12, 32) X(15, 15, 16) X(16, 15, 16) \
    X(25, 10, 32) X(18, 15, 32) X(18, 16, 32) X(20, 15, 32) X(20, 16, 32) X(18, 18, 32) X(20, 18, 32) X(25, 15, 32) X(24, 16, 32) \
    X(20, 20, 32) X(27, 15, 32) X(24, 18, 32) X(25, 18, 32) X(24, 20, 32) X(27, 18, 32)
#define MSL_ROWTM_LIST_B(X) \
    X(25, 20, 32) X(27, 20, 32) X(24, 24, 32) X(25, 24, 32) X(25, 25, 32) X(32, 20, 32) X(27, 24, 32) X(27, 25, 32) X(30, 24, 32) \
    X(27, 27, 32) X(30, 25, 32) X(32, 24, 32) X(32, 25, 32) X(30, 27, 32) X(32, 27, 32) X(30, 30, 32) X(32, 30, 32)
#define MSL_ROWTM_LIST_C(X) \
    X(27, 18, 64) X(25, 20, 64) X(27, 20, 64) X(24, 24, 64) X(25, 24, 64) X(25, 25, 64) X(32, 20, 64) X(27, 24, 64) X(27, 25, 64) \
    X(30, 24, 64) X(27, 27, 64) X(30, 25, 64) X(32, 24, 64) X(32, 25, 64) X(30, 27, 64) X(32, 27, 64)

// lengths with a factor 7 (radix-7 register butterfly, fft_regs.h: dif7_level): A * B below / from 400, and 2 A * B on one wave per line
#define MSL_ROWTM_LIST_D(X) \
    X(14, 10, 16) X(21, 7, 32) X(14, 12, 16) X(25, 7, 32) X(21, 9, 32) X(14, 14, 16) X(15, 14, 16) X(16, 14, 16) X(18, 14, 32) X(20, 14, 32) X(21, 14, 32) X(21, 15, 32) X(21, 16, 32) X(25, 14, 32) X(21, 18, 32) X(28, 14, 32)
#define MSL_ROWTM_LIST_E(X) \
    X(21, 20, 32) X(21, 21, 32) X(28, 16, 32) X(24, 21, 32) X(25, 21, 32) X(28, 20, 32) X(27, 21, 32) X(28, 21, 32) X(30, 21, 32) X(28, 24, 32) X(28, 25, 32) X(28, 27, 32) X(28, 28, 32) X(30, 28, 32) X(32, 28, 32)
#define MSL_ROWTM_LIST_F(X) \
    X(25, 21, 64) X(28, 20, 64) X(27, 21, 64) X(28, 21, 64) X(30, 21, 64) X(28, 24, 64) X(28, 25, 64) X(28, 27, 64) X(28, 28, 64) X(30, 28, 64)

namespace msl {

template <int A, int B, int G>
static bool rowTM_launch_one(const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)rowTM_pass_kernel<A, B, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit);
    hipLaunchKernelGGL((rowTM_pass_kernel<A, B, G>), dim3(grid), dim3(16 * G), rowTM_lds_bytes(A, B), stream, job);
    return true;
}

bool rowTM_launch_a(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream);
bool rowTM_launch_b(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream);
bool rowTM_launch_c(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream);
bool rowTM_launch_d(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream);
bool rowTM_launch_e(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream);
bool rowTM_launch_f(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream);
template <int A, int B>
static bool rowTM2_launch_one(const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)rowTM2_pass_kernel<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit);
    hipLaunchKernelGGL((rowTM2_pass_kernel<A, B>), dim3(grid), dim3(512), rowTM2_lds_bytes(A, B), stream, job);
    return true;
}

}  // namespace msl
