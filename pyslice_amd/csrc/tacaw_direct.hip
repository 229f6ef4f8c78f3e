// time_direct_kernel<T>: one lane per pixel, the whole time line in registers (tacaw_time.h) -- the 40 instantiations.
#include <algorithm>
#include "tacaw_launch.h"

namespace msl {

// frame counts with a per-lane kernel: the 2-3-5-7-smooth numbers in [TDIR_MIN, TDIR_MAX] (radix 7: fft_regs.h, dif7_level)
#define MSL_TDIR_LENGTHS(X) X(16) X(18) X(20) X(24) X(25) X(27) X(30) X(32) X(36) X(40) X(45) X(48) X(50) X(54) X(60) X(64) X(72) X(75) \
    X(80) X(81) X(90) X(96) X(100) X(108) X(120) X(125) X(128) \
    X(21) X(28) X(35) X(42) X(49) X(56) X(63) X(70) X(84) X(98) X(105) X(112) X(126)

bool time_direct_has(int T) { return T >= TDIR_MIN && T <= TDIR_MAX && fft_smooth7(T); }

template <int T>
static bool launch_t(const TimeJob& j, int n_cus, hipStream_t stream) {
    static_assert(fft_smooth7(T) && T >= TDIR_MIN && T <= TDIR_MAX, "no per-lane time kernel for this frame count");
    const long long tiles = ((long long)(j.npix + 255) / 256) * j.n_images;
    int per_cu = 1;                                  // 1 for the long lines (512 registers per lane), more for the short ones
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)time_direct_kernel<T>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    const int grid = (int)std::min<long long>(tiles, (long long)n_cus * per_cu);
    hipLaunchKernelGGL((time_direct_kernel<T>), dim3(grid), dim3(256), 0, stream, j);
    return true;
}

bool time_direct_launch(const TimeJob& j, int n_cus, hipStream_t stream) {
    switch (j.T) {
#define X(n) case n: return launch_t<n>(j, n_cus, stream);
        MSL_TDIR_LENGTHS(X)
#undef X
    }
    return false;
}

}  // namespace msl
