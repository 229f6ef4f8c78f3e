// Mixed-radix transposing pass, line lengths A * B from 400 with a factor 7 (rowtm_launch.h).
#include "rowtm_launch.h"

namespace msl {

bool rowTM_launch_e(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
#define X(a, b, g) if (n == (a) * (b)) return rowTM_launch_one<a, b, g>(job, grid, lds_limit, stream);
    MSL_ROWTM_LIST_E(X)
#undef X
    return false;
}

}  // namespace msl
