// Instantiations and launch of the transposing slice-loop pass (rowt_pass.h) -- a translation unit of its own so that the library
// builds in parallel (build_native.py); mslice.hip sees only rowT_launch / rowT_selftest.
#include "rowt_pass.h"

namespace msl {

template <int R, bool IN_P, bool OUT_P, int FL>
static bool launch_one(const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)rowT_pass_kernel<R, 16, IN_P, OUT_P, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit);
    hipLaunchKernelGGL((rowT_pass_kernel<R, 16, IN_P, OUT_P, FL>), dim3(grid), dim3(16 * R), rowT_lds_bytes(R), stream, job);
    return true;
}

// Which (line order in, line order out, propagation halves) occur: between two transposing passes both orders are interleaved and
// both halves run (1, 1, 3); the first pass of a stack reads natural order and has no half in front (0, 1, 2); the last transposing
// pass writes natural order -- for the in-place last pass of 256 / 1024 grids, both halves (1, 0, 3), or as the last pass itself
// (1, 0, 1); without the interleaved order (other kernels on the second axis, MSL_NO_INTERLEAVE, stacks of one or two slices) every
// flag combination occurs with natural order on both sides: (0, 0, 3) compiled, the others through the run-time form.
template <int R>
static bool launch_r(const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
    const bool in_p = job.flags & P2_IN_PAIRED, out_p = job.flags & P2_OUT_PAIRED;
    const int fl = job.flags & (P2_PRE_A | P2_POST_A);
    if (in_p && out_p && fl == 3) return launch_one<R, true, true, 3>(job, grid, lds_limit, stream);
    if (!in_p && out_p && fl == 2) return launch_one<R, false, true, 2>(job, grid, lds_limit, stream);
    if (in_p && !out_p && fl == 1) return launch_one<R, true, false, 1>(job, grid, lds_limit, stream);
    if (in_p && !out_p && fl == 3) return launch_one<R, true, false, 3>(job, grid, lds_limit, stream);
    if (!in_p && !out_p && fl == 3) return launch_one<R, false, false, 3>(job, grid, lds_limit, stream);
    if (!in_p && !out_p) return launch_one<R, false, false, -1>(job, grid, lds_limit, stream);
    return false;
}

bool rowT_launch(int R, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream) {
    if (R == 32) return launch_r<32>(job, grid, lds_limit, stream);
    if (R == 16) return launch_r<16>(job, grid, lds_limit, stream);
    return false;
}

// One wave runs the add-tid exchange on known data: lane (g, l) holds in register j the value 1000 g + 32 j + l (imaginary part
// negated) and must come back with 1000 g + 32 l + j in register j.  The exchange rests on an M0 write inside inline asm that the
// compiler's hazard recogniser does not see (exchange_addtid): a toolchain that schedules it differently would corrupt every
// transform silently.  tests/test_abi_and_host.py checks the ISA on the build machine, this checks the device at msl_create.
template <int R>
__global__ void __launch_bounds__(64) exchange_selftest_kernel(int* bad) {
    __shared__ __attribute__((aligned(16))) float scr[R * 68];
    const int lane = threadIdx.x, g = lane / R, l = lane % R;
    float2 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = make_float2((float)(1000 * g + 32 * j + l), -(float)(1000 * g + 32 * j + l));
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)scr);
    exchange_addtid<R>(v, scr, base, l, lane);
    int wrong = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const float want = (float)(1000 * g + 32 * l + j);
        wrong += (v[j].x != want) + (v[j].y != -want);
    }
    if (wrong) atomicAdd(bad, wrong);
}

int rowT_selftest(hipStream_t stream) {
    int* bad = nullptr;
    if (hipMalloc(&bad, sizeof(int)) != hipSuccess) return -1;
    int host = -1;
    bool ok = hipMemsetAsync(bad, 0, sizeof(int), stream) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(exchange_selftest_kernel<32>, dim3(1), dim3(64), 0, stream, bad);
        hipLaunchKernelGGL(exchange_selftest_kernel<16>, dim3(1), dim3(64), 0, stream, bad);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&host, bad, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
             hipStreamSynchronize(stream) == hipSuccess;
    }
    (void)hipFree(bad);
    return ok ? host : -1;
}

}  // namespace msl
