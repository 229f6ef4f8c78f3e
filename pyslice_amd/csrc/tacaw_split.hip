// time_split_kernel<TP, L, 1>: smooth frame counts 129 .. 512 whose shape keeps one block per wave (odd L, L = 6), and the rule
// that picks a shape for T.
#include "tacaw_split.h"

namespace msl {

// smooth frame counts 129 .. 512 as L x TP with TP <= 128: L = 2, else 4, else 6, else 3 or 5 (odd counts; two waves per SIMD from
// L = 5 on: TP <= 100).  L = 2 and 4 put two blocks on the halves of a wave (HB = 2, tacaw_split2.hip: 32-pixel tiles, workgroups
// of one or two waves, two to four of them per CU -- independent workgroups cover each other's barriers: T = 500 0.56 -> 0.61,
// T = 300 0.49 -> 0.62, T = 256 0.63 -> 0.69 in same-box A/Bs; L = 6 lost that way, 0.53 -> 0.42, and keeps a block per wave).
// 513 .. 1024 as 8 x TP (else 6 x TP), two blocks per wave as well.
#define MSL_TSPLIT_SHAPES(X) X(45, 3) X(75, 3) X(81, 3) X(45, 6) X(125, 3) X(81, 5) X(75, 6) X(81, 6)

int time_split_waves(int T, int* hb) {           // L (and the blocks per wave), or 0: no such kernel
    if (hb) *hb = 1;
    if (T <= TDIR_MAX || T > 1024 || !fft_smooth(T)) return 0;
    if (T <= 512) {
        for (int L : {2, 4, 6, 3, 5})
            if (T % L == 0 && T / L <= TDIR_MAX && (L <= 4 || T / L <= 100)) { if (hb && L <= 4 && L % 2 == 0) *hb = 2; return L; }
        return 0;
    }
    if (hb) *hb = 2;
    for (int L : {8, 6})
        if (T % L == 0 && T / L <= TDIR_MAX) return L;
    return 0;
}

bool time_split_launch(const TimeJob& j, int n_cus, size_t lds_limit, hipStream_t stream) {
    int HB = 1;
    const int T = j.T, L = time_split_waves(T, &HB);
    if (!L) return false;
    if (HB == 2) return time_split2_launch(j, L, n_cus, lds_limit, stream);
#define X(tp, l) if (T == (tp) * (l) && L == (l)) return launch_split_t<tp, l, 1>(j, n_cus, lds_limit, stream);
    MSL_TSPLIT_SHAPES(X)
#undef X
    return false;
}

}  // namespace msl
