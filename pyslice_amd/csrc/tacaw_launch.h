// Launchers of the per-lane / wave-split TACAW time kernels (tacaw_time.h).  Their 70 instantiations live in translation units of
// their own -- tacaw_direct.hip, tacaw_split.hip (one block per wave), tacaw_split2.hip (two blocks per wave) -- so that the
// library builds in parallel (build_native.py); mslice.hip sees only these declarations.
#pragma once
#include <hip/hip_runtime.h>
#include "tacaw_regs.h"

namespace msl {

// is there a kernel for T frames?
bool time_direct_has(int T);                          // 2-3-5-smooth, TDIR_MIN <= T <= TDIR_MAX
int time_split_waves(int T, int* hb = nullptr);       // L of the T = L x TP shape (and the blocks per wave), or 0
bool time_split_fits(int T, long long npix);          // ... and do the 32-bit row offsets of one of its kernels cover an image of npix pixels?

// launch on `stream`; false: no kernel for job.T (nothing launched).  Errors of the launch itself: hipGetLastError().
bool time_direct_launch(const TimeJob& job, int n_cus, hipStream_t stream);
bool time_split_launch(const TimeJob& job, int n_cus, size_t lds_limit, hipStream_t stream);      // job.tw: W_T^n, n < T
bool time_split2_launch(const TimeJob& job, int L, int n_cus, size_t lds_limit, hipStream_t stream);   // (called by time_split_launch)

}  // namespace msl
