// shared by tacaw_split.hip / tacaw_split2.hip: launch of one time_split_kernel<TP, L, HB> instantiation
#pragma once
#include <algorithm>
#include "tacaw_launch.h"

namespace msl {

template <int TP, int L, int HB>
static bool launch_split_t(const TimeJob& j, int n_cus, size_t lds_limit, hipStream_t stream) {
    const size_t lds = tsplit_lds_bytes(TP, L, HB);
    const int threads = 64 * L / HB;
    const long long tiles = ((long long)(j.npix + 64 / HB - 1) / (64 / HB)) * j.n_images;
    (void)hipFuncSetAttribute((const void*)time_split_kernel<TP, L, HB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit);
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)time_split_kernel<TP, L, HB>, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const int grid = (int)std::min<long long>(tiles, (long long)n_cus * per_cu);
    hipLaunchKernelGGL((time_split_kernel<TP, L, HB>), dim3(grid), dim3(threads), lds, stream, j);
    return true;
}

}  // namespace msl
