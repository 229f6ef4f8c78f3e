// Mixed-radix transposing pass, line lengths 2 A * B on one wave per line (rowtm_launch.h).
#include "rowtm_launch.h"

namespace msl {

template <int A, int B>
static bool rowTM2_launch_one(const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)rowTM2_pass_kernel<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit);
    hipLaunchKernelGGL((rowTM2_pass_kernel<A, B>), dim3(grid), dim3(512), rowTM2_lds_bytes(A, B), stream, job);
    return true;
}

bool rowTM_launch_c(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
#define X(a, b, g) if (n == 2 * (a) * (b)) return rowTM2_launch_one<a, b>(job, grid, lds_limit, stream);
    MSL_ROWTM_LIST_C(X)
#undef X
    return false;
}

}  // namespace msl
