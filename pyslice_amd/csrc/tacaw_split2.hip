// time_split_kernel<TP, L, 2>: two blocks per wave -- every 2 x TP and 4 x TP shape up to 512 frames, 8 x TP / 6 x TP above.
#include "tacaw_split.h"

namespace msl {

#define MSL_TSPLIT2_SHAPES(X) X(72, 2) X(75, 2) X(80, 2) X(81, 2) X(90, 2) X(96, 2) X(100, 2) X(108, 2) X(120, 2) X(125, 2) X(128, 2) \
    X(72, 4) X(75, 4) X(80, 4) X(81, 4) X(90, 4) X(96, 4) X(100, 4) X(108, 4) X(120, 4) X(125, 4) X(128, 4) \
    X(90, 6) X(72, 8) X(75, 8) X(80, 8) X(81, 8) X(90, 8) X(125, 6) X(96, 8) X(100, 8) X(108, 8) X(120, 8) X(125, 8) X(128, 8)

bool time_split2_launch(const TimeJob& j, int L, int n_cus, size_t lds_limit, hipStream_t stream) {
    const int T = j.T;
#define X(tp, l) if (T == (tp) * (l) && L == (l)) return launch_split_t<tp, l, 2>(j, n_cus, lds_limit, stream);
    MSL_TSPLIT2_SHAPES(X)
#undef X
    return false;
}

}  // namespace msl
