// Mixed-radix transposing pass, line lengths 2 A * B on one wave per line (rowtm_launch.h).
#include "rowtm_launch.h"

namespace msl {

bool rowTM_launch_c(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
#define X(a, b, g) if (n == 2 * (a) * (b)) return rowTM2_launch_one<a, b>(job, grid, lds_limit, stream);
    MSL_ROWTM_LIST_C(X)
#undef X
    return false;
}

}  // namespace msl
