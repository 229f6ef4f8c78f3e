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
// one block per wave for the 2 x TP / 4 x TP shapes too: images beyond the two-block kernels' offset range (2048^2 ...)
#define MSL_TSPLIT_BIG_SHAPES(X) X(72, 2) X(75, 2) X(80, 2) X(81, 2) X(90, 2) X(96, 2) X(100, 2) X(108, 2) X(120, 2) X(125, 2) X(128, 2) \
    X(72, 4) X(75, 4) X(80, 4) X(81, 4) X(90, 4) X(96, 4) X(100, 4) X(108, 4) X(120, 4) X(125, 4) X(128, 4)

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

// The buffer unit adds the lane offset and the scalar row offset in 32 bits (the sum wraps: measured).  One block per wave: pixel
// + up to 64 rows; two blocks: pixel + TP rows to the second block + up to (TP + 1) / 2 - 1 rows.
static bool fits_one(long long npix) { return 65ull * (unsigned long long)npix * 8ull < (1ull << 32); }
static bool fits_two(int TP, long long npix) { return (unsigned long long)(TP + (TP + 1) / 2) * (unsigned long long)npix * 8ull < (1ull << 32); }

bool time_split_fits(int T, long long npix) {
    int HB = 1;
    const int L = time_split_waves(T, &HB);
    if (!L) return false;
    if (HB == 1) return fits_one(npix);
    return fits_two(T / L, npix) || (T <= 512 && fits_one(npix));
}

bool time_split_launch(const TimeJob& j, int n_cus, size_t lds_limit, hipStream_t stream) {
    int HB = 1;
    const int T = j.T, L = time_split_waves(T, &HB);
    if (!L) return false;
    if (HB == 2 && fits_two(T / L, j.npix)) return time_split2_launch(j, L, n_cus, lds_limit, stream);
    if (!fits_one(j.npix)) return false;
    if (HB == 2) {
#define X(tp, l) if (T == (tp) * (l) && L == (l)) return launch_split_t<tp, l, 1>(j, n_cus, lds_limit, stream);
        MSL_TSPLIT_BIG_SHAPES(X)
#undef X
        return false;
    }
#define X(tp, l) if (T == (tp) * (l) && L == (l)) return launch_split_t<tp, l, 1>(j, n_cus, lds_limit, stream);
    MSL_TSPLIT_SHAPES(X)
#undef X
    return false;
}

}  // namespace msl
