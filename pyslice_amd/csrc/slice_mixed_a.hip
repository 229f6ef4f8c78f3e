// Mixed-radix transposing pass, line lengths below 500 (rowtm_launch.h), and the length -> (A, B, G) table of both halves.
#include "rowtm_launch.h"

namespace msl {

bool rowTM_factors(int n, int* A, int* B, int* G) {
#define X(a, b, g) if (n == (a) * (b)) { *A = (a); *B = (b); *G = (g); return true; }
    MSL_ROWTM_LIST_A(X)
    MSL_ROWTM_LIST_B(X)
    MSL_ROWTM_LIST_D(X)
    MSL_ROWTM_LIST_E(X)
#undef X
#define X(a, b, g) if (n == 2 * (a) * (b)) { *A = (a); *B = (b); *G = (g); return true; }
    MSL_ROWTM_LIST_C(X)
    MSL_ROWTM_LIST_F(X)
#undef X
    return false;
}

bool rowTM_launch_a(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
#define X(a, b, g) if (n == (a) * (b)) return rowTM_launch_one<a, b, g>(job, grid, lds_limit, stream);
    MSL_ROWTM_LIST_A(X)
#undef X
    return false;
}

bool rowTM_launch(int n, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
    int A = 0, B = 0, G = 0;
    if (!rowTM_factors(n, &A, &B, &G)) return false;
    if (n % 7 == 0) return G == 64 ? rowTM_launch_f(n, job, grid, lds_limit, stream)
                                   : (n < 400 ? rowTM_launch_d(n, job, grid, lds_limit, stream) : rowTM_launch_e(n, job, grid, lds_limit, stream));
    if (G == 64) return rowTM_launch_c(n, job, grid, lds_limit, stream);
    return n < 500 ? rowTM_launch_a(n, job, grid, lds_limit, stream) : rowTM_launch_b(n, job, grid, lds_limit, stream);
}

}  // namespace msl
