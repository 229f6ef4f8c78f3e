// libmslice: C-ABI HIP implementation of the multislice hot path for MI355X (gfx950).
// Entry points are declared in include/mslice.h (each cites the reference function it replaces).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mslice.h"
#include "fft_generic.h"
#include "fft_pow2.h"
#include "rowtm_pass.h"
#include "potential.h"
#include "reduce.h"
#include "stream.h"
#include "tacaw_time.h"
#include "tacaw_launch.h"

using namespace msl;

namespace {

thread_local std::string g_create_error;

struct FftPlan {
    int M = 0;                  // length the Stockham stages run on (N, or a power of two >= 2N-1: Bluestein)
    float2* chirp = nullptr;
    float2* bfilt = nullptr;
    int N = 0;
    int n_stages = 0;
    int radix[MSL_MAX_STAGES] = {0};
    float2* tw = nullptr;       // device twiddles
    bool ok = false;
};

enum LaunchKind { K_ROW = 0, K_COL = 1, K_OTHER = 2, K_NKINDS = 3 };

struct EventSet {
    std::vector<hipEvent_t> ev;
    std::vector<int> kind;      // kind[i] = kind of the launch between ev[i] and ev[i+1]
    int used = 0;               // number of events recorded
    bool pending = false;
};

}  // namespace

struct msl_handle {
    msl_config cfg{};
    hipStream_t stream = nullptr;
    std::string err;
    // plans
    FftPlan plan_x, plan_y, plan_t;
    // four-step (register-resident) kernels: R = 32 (N=1024) or 16 (N=256); 0 = use the generic kernel
    int Rx = 0, Ry = 0;
    float2* tw4_x = nullptr;
    float2* tw4_y = nullptr;
    int n_cus = 256;
    // one-pass-per-slice path (transposing passes): second work buffer in (P, ny, nx+pad) layout, transposed
    // probes and the transposed transmission slices
    int wx = 0, wy = 0, wx0 = 0, wy0 = 0;   // k-window of the stored exit-wave spectra (fftshifted coordinates)
    size_t wpix = 0;               // stored pixels per exit-wave spectrum: the (binned) k-window or the whole grid
    size_t wpitch = 0;             // pixel pitch of the images of wf / intensity: wpix rounded up to 32 pixels (pad pixels are zero: to the
                                   // time kernels they are pixels like any other), include/mslice.h: msl_result_pitch
    size_t intensity_ld = 0;       // pixel pitch of the resident intensity buffer: wpitch after msl_tacaw, wpix after a stream
    int bx = 1, by = 1;            // detector binning: stored pixel = sum of bx x by neighbouring pixels of the window
    float2* bin_stage = nullptr;   // binning: full-resolution window of the frames of one launch sequence, (FB*P, wx, wy)
    // streaming TACAW
    float2* st_acc = nullptr; double2* st_s1 = nullptr; double* st_s2 = nullptr; float2* st_tw = nullptr; int* st_bins = nullptr;
    float2* st_ref = nullptr; bool st_have_ref = false;      // reference pattern subtracted before folding (msl_tacaw_stream_set_reference)
    int st_T = 0, st_F = 0; bool st_open = false;
    int64_t intensity_F = 0;       // frequency bins of the resident intensity buffer (T after msl_tacaw, n_bins after a stream)
    char* scratch = nullptr;       // reductions: partial sums / masks / index lists
    size_t scratch_bytes = 0;
    bool onepass = false;
    bool scheme_b = false;         // a direction of 2R^2 points: every pass transposes, first pass along y, final transpose if nz is odd
    // one-pass kernel per direction: R^2 register kernel, 2 R^2 (two) register kernel with its tables, or the generic LDS kernel
    // (breg: any length <= R^2/2 by Bluestein's chirp-z on the R^2 register FFTs, with its filter bf and chirp bw)
    // (breg2: lengths 513..1024 by the same scheme on the wave-per-line 2048-point FFT; tw = T[k1*64+n2], tw2 = W_64 table)
    struct OpDir { int R = 0; bool two = false; bool generic = false; bool mixed = false; float2* mtw = nullptr;   // mixed: smooth length A * B on rowTM_pass_kernel, mtw = its two twiddle tables
         bool breg = false; bool breg2 = false; bool breg4 = false; bool wave2k = false; float2* tw = nullptr; float2* tw2 = nullptr; float2* qf = nullptr;
        // chirp-z tables of an axis whose slice-loop kernel is not a chirp / convolution kernel (a power of two next to another length):
        // only the potential's inverse transform uses them (ifftTB_kernel / ifftTB2_kernel).  cz_R: 16 / 32 (M = R^2) or 64 (the 2048-point wave FFT)
        int cz_R = 0; float2* cz_tw = nullptr; float2* cz_tw2 = nullptr; float2* cz_bf = nullptr; float2* cz_bw = nullptr;
                   float2* ptab = nullptr; float2* bf = nullptr; float2* bw = nullptr; } opx, opy;
    OpDir opt;                     // chirp-z tables of the TACAW time axis (cz_* only), made for opt_T frames (time_cz_kernel)
    int opt_T = 0;
    float2* tsplit_tw = nullptr;   // W_T^n, n < T: cross-wave butterflies of time_split_kernel, made for tsplit_T frames
    int tsplit_T = 0;
    float2* psiT = nullptr;
    float2* psi0T = nullptr;
    bool need_psi0T = false;
    int keys_cap = 0;              // capacity of d_counts / d_start (slice x species bins)
    // pinned host staging for the per-frame inputs (species maps, Z, positions), two slots used alternately: the call
    // copies the caller's arrays here and returns; the H2D copies run on the stream (no pointer into caller memory is kept)
    struct HostStage { char* buf = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool used = false; } stage[2];
    unsigned stage_pos = 0;
    float2* transT = nullptr;
    int pitchT = 0;
    int debug_flags_mask = -1;
    int row_pchunk = 0;         // 0 = auto (MSL_ROW_PCHUNK, debug)
    int pitch = 0;              // row pitch (elements) of psi0/psi; > ny de-aliases the column pass's 128-byte segments
    // frame batching (msl_config.frame_batch): FB frames share one launch of every slice-loop kernel -- image index =
    // frame * P + probe, each frame with its own transmission stack.  Work buffers hold FB * P images (the probes are
    // replicated once per frame of the batch), trans / transT hold FB stacks; cur_batch = stack the next potential goes to.
    int FB = 1, cur_batch = 0;
    // device buffers
    float2* psi0 = nullptr;
    float2* psi = nullptr;
    float2* trans = nullptr;
    float* V = nullptr;
    float2* wf = nullptr;
    float* intensity = nullptr;
    size_t intensity_elems = 0;
    float2* pxt = nullptr;      // exp(-i pi lambda dz kx^2)/nx
    float2* pyt = nullptr;
    double* d_abcd = nullptr;
    double* d_lo = nullptr;
    double* d_hi = nullptr;
    bool have_kirkland = false, have_slices = false, have_probes = false, have_potential = false, have_exit = false;
    int frames_done = 0;
    // potential scratch (grown on demand)
    size_t atom_cap = 0;
    double* d_pos = nullptr;
    int* d_Z = nullptr;
    int* d_key = nullptr;
    int* d_order = nullptr;
    double* d_u1 = nullptr;
    double* d_u2 = nullptr;
    float2* d_ex = nullptr;
    float2* d_ey = nullptr;
    int* d_counts = nullptr;
    int* d_start = nullptr;
    int* d_z2s = nullptr;
    int* d_species = nullptr;
    float* d_ff = nullptr;
    int ff_species_cap = 0;
    int n_species = 0;
    int ff_species[104] = {0};     // species list the resident form-factor table was computed for (frame-invariant: computed once per run)
    int ff_n = 0;
    double* d_xy = nullptr;
    // counters
    msl_counters ctr{};
    double ms_kind[K_NKINDS] = {0, 0, 0};
    uint64_t n_kind[K_NKINDS] = {0, 0, 0};
    std::vector<EventSet> ring;
    int ring_pos = 0;
    EventSet* cur = nullptr;
    int lds_limit = 160 * 1024;
};

namespace {

int fail(msl_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

// Diagnostic switches (A/B runs, cross-checks of the tests) are read from the environment ONLY when MSL_DEBUG is set: a stray
// MSL_* variable in a user's environment must not change which kernels run.
const char* dbg_env(const char* name) { return getenv("MSL_DEBUG") ? getenv(name) : nullptr; }

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return fail(h, MSL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dalloc(msl_handle* h, T** p, size_t n) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (n == 0) return MSL_OK;
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) { *p = nullptr; return fail(h, MSL_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e)); }
    return MSL_OK;
}

bool factorize(int N, FftPlan& pl) {
    static const int cand[] = {8, 4, 2, 3, 5, 7, 11, 13};
    pl.N = N; pl.n_stages = 0;
    int n = N;
    for (int r : cand) {
        while (n % r == 0 && n > 1) {
            if (pl.n_stages >= MSL_MAX_STAGES) return false;
            pl.radix[pl.n_stages++] = r;
            n /= r;
        }
    }
    return n == 1;
}

// in-place radix-2 FFT in double on the host (Bluestein filter setup only)
void host_fft_pow2(std::vector<double>& re, std::vector<double>& im) {
    const int n = (int)re.size();
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (int len = 2; len <= n; len <<= 1) {
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double a = -2.0 * M_PI * k / len, c = cos(a), s = sin(a);
                const int u = i + k, v = i + k + len / 2;
                const double xr = re[v] * c - im[v] * s, xi = re[v] * s + im[v] * c;
                re[v] = re[u] - xr; im[v] = im[u] - xi; re[u] += xr; im[u] += xi;
            }
    }
}

int make_plan(msl_handle* h, FftPlan& pl, int N) {
    if (pl.ok && pl.N == N) return MSL_OK;
    pl.ok = false;
    if (N < 1) return fail(h, MSL_ERR_INVALID, "FFT length %d", N);
    int M = N;
    bool native = factorize(N, pl);
    if (!native) {
        // Bluestein: convolution length = power of two >= 2N-1
        M = 1;
        while (M < 2 * N - 1) M <<= 1;
        FftPlan tmp;
        if (!factorize(M, tmp)) return fail(h, MSL_ERR_UNSUPPORTED, "FFT length %d: no plan", N);
        pl.n_stages = tmp.n_stages;
        for (int i = 0; i < tmp.n_stages; ++i) pl.radix[i] = tmp.radix[i];
    }
    pl.N = N; pl.M = M;
    if ((size_t)M * 8 * 2 > (size_t)h->lds_limit)
        return fail(h, MSL_ERR_UNSUPPORTED, "FFT length %d%s does not fit the LDS-resident kernel", N,
                    native ? "" : " (Bluestein, prime factor > 13)");
    std::vector<float2> tw(M);
    for (int j = 0; j < M; ++j) {
        double a = -2.0 * M_PI * (double)j / (double)M;
        tw[j] = make_float2((float)cos(a), (float)sin(a));
    }
    int rc = dalloc(h, &pl.tw, (size_t)M);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(pl.tw, tw.data(), M * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!native) {
        std::vector<float2> chirp(N);
        std::vector<double> br(M, 0.0), bi(M, 0.0);
        for (int n = 0; n < N; ++n) {
            const long long q = ((long long)n * n) % (2LL * N);
            const double a = -M_PI * (double)q / (double)N;
            chirp[n] = make_float2((float)cos(a), (float)sin(a));
            br[n] = cos(a); bi[n] = -sin(a);                 // conj chirp
            if (n) { br[M - n] = br[n]; bi[M - n] = bi[n]; }
        }
        host_fft_pow2(br, bi);
        std::vector<float2> bf(M);
        for (int j = 0; j < M; ++j) bf[j] = make_float2((float)(br[j] / M), (float)(bi[j] / M));
        if ((rc = dalloc(h, &pl.chirp, (size_t)N))) return rc;
        if ((rc = dalloc(h, &pl.bfilt, (size_t)M))) return rc;
        HIPCHK(h, hipMemcpyAsync(pl.chirp, chirp.data(), N * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(pl.bfilt, bf.data(), M * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    pl.ok = true;
    return MSL_OK;
}

// ---- per-launch event timing -------------------------------------------------------------
int resolve_set(msl_handle* h, EventSet& s) {
    if (!s.pending) return MSL_OK;
    if (s.used >= 2) {
        HIPCHK(h, hipEventSynchronize(s.ev[s.used - 1]));
        for (int i = 0; i + 1 < s.used; ++i) {
            float ms = 0.f;
            HIPCHK(h, hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
            int k = s.kind[i];
            h->ms_kind[k] += ms;
            h->n_kind[k] += 1;
        }
    }
    s.used = 0; s.pending = false;
    return MSL_OK;
}

int begin_timed(msl_handle* h, int max_launches) {
    if (!h->cfg.launch_timing) { h->cur = nullptr; return MSL_OK; }
    if (h->ring.empty()) h->ring.resize(4);
    EventSet& s = h->ring[h->ring_pos];
    h->ring_pos = (h->ring_pos + 1) % (int)h->ring.size();
    int rc = resolve_set(h, s);
    if (rc) return rc;
    while ((int)s.ev.size() < max_launches + 1) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreate(&e));
        s.ev.push_back(e);
    }
    s.kind.assign(max_launches + 1, K_OTHER);
    s.used = 0;
    HIPCHK(h, hipEventRecord(s.ev[s.used++], h->stream));
    s.pending = true;
    h->cur = &s;
    return MSL_OK;
}

int mark_launch(msl_handle* h, int kind) {
    EventSet* s = h->cur;
    if (!s || s->used >= (int)s->ev.size()) return MSL_OK;
    s->kind[s->used - 1] = kind;
    HIPCHK(h, hipEventRecord(s->ev[s->used++], h->stream));
    return MSL_OK;
}

int resolve_all(msl_handle* h) {
    for (auto& s : h->ring) { int rc = resolve_set(h, s); if (rc) return rc; }
    h->cur = nullptr;
    return MSL_OK;
}

// ---- generic line-FFT launch ----------------------------------------------------------------
struct LineArgs {
    const float2* in = nullptr; float2* out = nullptr; float* out_real = nullptr;
    long long n_lines = 0; int lines_per_image = 1;
    long long in_es = 1, in_ls = 0, in_is = 0, out_es = 1, out_ls = 0, out_is = 0;
    int contiguous_lines = 0;
    int fft1 = 0, fft2 = 0;         // the common two-step form: [fft1] x m1 [fft2] store(x m2)
    int m1_kind = MUL_NONE, m2_kind = MUL_NONE; const float2* m1 = nullptr; const float2* m2 = nullptr;
    // general form (used when n_steps > 0): step i = [FFT fft[i]] then [x mul[i]]
    int n_steps = 0; int fft[MSL_GEN_STEPS] = {0, 0, 0, 0}; int mkind[MSL_GEN_STEPS] = {0, 0, 0, 0};
    const float2* mul[MSL_GEN_STEPS] = {nullptr, nullptr, nullptr, nullptr};
    int out_contiguous = -1;        // store mapping; -1 = same as contiguous_lines
    long long m1_ls = 0, m2_ls = 0;
    int store_mode = STORE_C64; int shift_n = 0, shift_r = 0; float scale = 1.f; float sigma = 0.f;
    int win_n0 = 0, win_nn = 0, win_r0 = 0, win_nr = 0;      // win_nn > 0: store only the window of the shifted output
    int group = 0; long long out_gs = 0, m_gs = 0;           // frame batching (LineJob)
};

int launch_lines(msl_handle* h, const FftPlan& pl, const LineArgs& a, int kind) {
    LineJob job{};
    job.in = a.in; job.out = a.out; job.out_real = a.out_real; job.tw = pl.tw; job.m2 = a.m2;
    job.n_lines = a.n_lines; job.in_es = a.in_es; job.in_ls = a.in_ls; job.in_is = a.in_is;
    job.out_es = a.out_es; job.out_ls = a.out_ls; job.out_is = a.out_is; job.m1_ls = a.m1_ls; job.m2_ls = a.m2_ls;
    job.N = pl.N; job.lines_per_image = a.lines_per_image; job.contiguous_lines = a.contiguous_lines;
    job.m2_kind = a.m2_kind;
    if (a.n_steps > 0) {
        job.n_steps = a.n_steps;
        for (int i = 0; i < a.n_steps; ++i) { job.fft[i] = a.fft[i]; job.mkind[i] = a.mkind[i]; job.mul[i] = a.mul[i]; }
    } else {
        job.n_steps = 2;
        job.fft[0] = a.fft1; job.mkind[0] = a.m1_kind; job.mul[0] = a.m1;
        job.fft[1] = a.fft2; job.mkind[1] = MUL_NONE; job.mul[1] = nullptr;
    }
    job.out_contiguous = a.out_contiguous < 0 ? a.contiguous_lines : a.out_contiguous;
    job.store_mode = a.store_mode; job.shift_n = a.shift_n; job.shift_r = a.shift_r;
    job.win_n0 = a.win_n0; job.win_nn = a.win_nn; job.win_r0 = a.win_r0; job.win_nr = a.win_nr;
    job.group = a.group; job.out_gs = a.out_gs; job.m_gs = a.m_gs;
    job.n_stages = pl.n_stages; for (int i = 0; i < pl.n_stages; ++i) job.radix[i] = pl.radix[i];
    job.scale = a.scale; job.sigma = a.sigma;
    const int N = pl.M;         // sizing follows the transform length (M > N for Bluestein lines)
    job.M = pl.M; job.chirp = pl.chirp; job.bfilt = pl.bfilt;
    bool has5 = false;
    for (int i = 0; i < pl.n_stages; ++i) has5 |= (pl.radix[i] == 5);
    // values per thread the plan's radices allow under either radix-5 policy (16 for powers of two, 14 with a 7, ...)
    auto elems_per_thread = [&](bool ceil5) {
        int e = MSL_GEN_E;
        for (int i = 0; i < pl.n_stages; ++i) e = std::min(e, gen_elems_per_thread(pl.radix[i], ceil5));
        return e;
    };
    int C = 1, nthreads = 64;
    size_t lds = 0;
    // tile of C lines, block size, and the resident waves per CU that result (128 VGPRs: at most 16)
    auto configure = [&](bool ceil5) -> long long {
        const int epl = elems_per_thread(ceil5);
        const int max_elems = epl * 1024;
        if (a.contiguous_lines) C = 16; else C = std::max(1, std::min(16, 8192 / N));
        C = (int)std::min<long long>(C, a.n_lines);
        job.npad = (a.contiguous_lines || job.out_contiguous) ? (N | 1) : N;      // odd pitch: conflict-free when lines are the fast index
        auto lds_need = [&](int c, bool tw) { return (size_t)MSL_GEN_HEADER + (size_t)c * job.npad * 8 + (tw ? (size_t)N * 8 : 0); };
        while (C > 1 && ((long long)C * N > max_elems || lds_need(C, false) > (size_t)h->lds_limit)) C >>= 1;
        if ((long long)C * N > max_elems || lds_need(C, false) > (size_t)h->lds_limit) return -1;
        job.C = C;
        job.tw_in_lds = lds_need(C, true) <= (size_t)h->lds_limit ? 1 : 0;
        lds = lds_need(C, job.tw_in_lds != 0);
        nthreads = (int)(((long long)C * N + epl - 1) / epl);
        nthreads = std::min(1024, std::max(64, (nthreads + 63) / 64 * 64));
        const int waves = nthreads / 64;
        const long long wgs = std::min<long long>((long long)(h->lds_limit / lds), 16 / waves);
        return std::max<long long>(1, wgs) * waves;              // resident waves per CU
    };
    bool ceil5 = false;
    long long score = configure(false);
    if (has5) {
        const long long score5 = configure(true);
        if (score5 > score) ceil5 = true; else score = configure(false);
    }
    if (score < 0) return fail(h, MSL_ERR_UNSUPPORTED, "line length %d too long for the LDS kernel", pl.N);
    long long tiles = (a.n_lines + C - 1) / C;
    if (tiles > 0x7fffffffLL) return fail(h, MSL_ERR_UNSUPPORTED, "too many FFT tiles");
    int rset = 0;
    for (int i = 0; i < pl.n_stages; ++i) {
        if (pl.radix[i] == 3 || pl.radix[i] == 5 || pl.radix[i] == 7) rset = std::max(rset, 1);
        if (pl.radix[i] > 8) rset = 2;
    }
    const dim3 grid((unsigned)tiles), block(nthreads);
    if (rset == 0) hipLaunchKernelGGL((line_fft_kernel<0, false>), grid, block, lds, h->stream, job);
    else if (rset == 1 && !ceil5) hipLaunchKernelGGL((line_fft_kernel<1, false>), grid, block, lds, h->stream, job);
    else if (rset == 1) hipLaunchKernelGGL((line_fft_kernel<1, true>), grid, block, lds, h->stream, job);
    else if (!ceil5) hipLaunchKernelGGL((line_fft_kernel<2, false>), grid, block, lds, h->stream, job);
    else hipLaunchKernelGGL((line_fft_kernel<2, true>), grid, block, lds, h->stream, job);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}

// row pass over (images x nx) rows of length ny; column pass over (images x ny) columns of length nx
LineArgs row_args(const msl_handle* h, const float2* in, float2* out, int images, int pitch) {
    LineArgs a;
    a.in = in; a.out = out;
    a.n_lines = (long long)images * h->cfg.nx; a.lines_per_image = h->cfg.nx;
    a.in_es = a.out_es = 1; a.in_ls = a.out_ls = pitch; a.in_is = a.out_is = (long long)h->cfg.nx * pitch;
    a.contiguous_lines = 0;
    return a;
}
LineArgs col_args(const msl_handle* h, const float2* in, float2* out, int images, int in_pitch, int out_pitch) {
    LineArgs a;
    a.in = in; a.out = out;
    a.n_lines = (long long)images * h->cfg.ny; a.lines_per_image = h->cfg.ny;
    a.in_es = in_pitch; a.out_es = out_pitch; a.in_ls = a.out_ls = 1;
    a.in_is = (long long)h->cfg.nx * in_pitch; a.out_is = (long long)h->cfg.nx * out_pitch;
    a.contiguous_lines = 1;
    return a;
}

// timing events of one call, destroyed on every exit path
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

// Probes per work item of the chunked kernels (a work item = 16 lines x a chunk of probes that share the t_k lines in
// registers; the grid is `slots` persistent workgroups).  The time of a launch is (rounds of work items over the slots) x
// (iterations per item): take the chunk that minimises it, the larger one on ties (fewer t_k loads).  Halving until every slot
// has an item, as round 1 did, wastes up to half a launch when the item count lands just above a multiple of the slots
// (300 lines x 64 probes: 304 items of 4 probes = 8 iteration times, 247 items of 5 probes = 5).
int choose_pchunk(long long line_blocks, int n_images, long long slots, int t_group) {
    int best = 1;
    long long best_cost = -1;
    for (int pc = 1; pc <= n_images; ++pc) {
        if (t_group > 0 && t_group % pc) continue;            // frame batching: a chunk stays inside one frame
        const long long items = line_blocks * ((n_images + pc - 1) / pc);
        const long long cost = ((items + slots - 1) / slots) * pc;
        if (best_cost < 0 || cost <= best_cost) { best = pc; best_cost = cost; }
    }
    return best;
}

// ---- four-step fast path ------------------------------------------------------------------------
int fast_radix(int n) { return n == 1024 ? 32 : (n == 256 ? 16 : 0); }

int make_tw4(msl_handle* h, float2** dst, int R);

// chirp-z tables for inverse transforms of n points (33 <= n <= 1024) on the register FFTs, see OpDir::cz_*
int make_cz_tables(msl_handle* h, msl_handle::OpDir& o, int n) {
    const bool wave = n > 512;
    const int R = wave ? 32 : (n <= 128 ? 16 : 32), M = wave ? 2048 : R * R, NH = M / 2;
    int r;
    if (wave) {
        std::vector<float2> T(M), W(64);
        for (int k1 = 0; k1 < 32; ++k1)
            for (int n2 = 0; n2 < 64; ++n2) {
                const double a = -2.0 * M_PI * (double)(k1 * n2) / (double)M;
                T[k1 * 64 + n2] = make_float2((float)cos(a), (float)sin(a));
            }
        for (int m = 0; m < 32; ++m) {
            const double a = -2.0 * M_PI * m / 64.0;
            W[m] = make_float2(1.f, 0.f);
            W[32 + m] = make_float2((float)cos(a), (float)sin(a));
        }
        if ((r = dalloc(h, &o.cz_tw, (size_t)M))) return r;
        if ((r = dalloc(h, &o.cz_tw2, (size_t)64))) return r;
        if (hipMemcpy(o.cz_tw, T.data(), M * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(o.cz_tw2, W.data(), 64 * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
            return fail(h, MSL_ERR_HIP, "chirp-z twiddle upload failed (%d points)", n);
    } else if ((r = make_tw4(h, &o.cz_tw, R))) {
        return r;
    }
    std::vector<float2> bw(NH, make_float2(0.f, 0.f)), bf(NH + 2, make_float2(0.f, 0.f));
    std::vector<double> cr(M, 0.0), ci(M, 0.0);
    for (int i = 0; i < n; ++i) {
        const long long q = ((long long)i * i) % (2LL * n);
        const double a = -M_PI * (double)q / (double)n;
        bw[i] = make_float2((float)cos(a), (float)sin(a));
        cr[i] = cos(a); ci[i] = -sin(a);
        if (i) { cr[M - i] = cr[i]; ci[M - i] = ci[i]; }
    }
    host_fft_pow2(cr, ci);
    for (int j = 0; j <= NH; ++j) bf[j] = make_float2((float)(cr[j] / M), (float)(ci[j] / M));
    if ((r = dalloc(h, &o.cz_bw, (size_t)NH))) return r;
    if ((r = dalloc(h, &o.cz_bf, (size_t)NH + 2))) return r;
    if (hipMemcpy(o.cz_bw, bw.data(), NH * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(o.cz_bf, bf.data(), (NH + 2) * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
        return fail(h, MSL_ERR_HIP, "chirp-z table upload failed (%d points)", n);
    o.cz_R = wave ? 64 : R;
    return MSL_OK;
}

int make_tw4(msl_handle* h, float2** dst, int R) {
    const int N = R * R;
    std::vector<float2> t(N);
    for (int k1 = 0; k1 < R; ++k1)
        for (int n2 = 0; n2 < R; ++n2) {
            double a = -2.0 * M_PI * (double)((k1 * n2) % N) / (double)N;
            t[k1 * R + n2] = make_float2((float)cos(a), (float)sin(a));
        }
    int rc = dalloc(h, dst, (size_t)N);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(*dst, t.data(), N * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MSL_OK;
}

template <int R>
int launch_row_fast_r(msl_handle* h, const RowJob& job, int kind) {
    constexpr int N = R * R, G = 256 / R;
    const size_t lds = (size_t)N * 16 + (size_t)G * R * (R + 1) * 4;
    const int per_cu = std::max(1, std::min(2, (int)((size_t)h->lds_limit / lds)));
    const long long slots = (long long)h->n_cus * per_cu;
    // probes per work item: as many as possible (t_z reuse) while still giving every slot an item
    RowJob j2 = job;
    const long long xg = job.nx / G;
    int pc = job.n_images;
    while (pc > 1 && xg * ((job.n_images + pc - 1) / pc) < slots) pc = (pc + 1) / 2;
    if (h->row_pchunk > 0) pc = std::min(h->row_pchunk, job.n_images);
    if (job.t_group > 0) while (job.t_group % pc) --pc;          // a chunk of probes shares one t_k line: stay inside a frame
    j2.pchunk = pc;
    const long long items = xg * ((job.n_images + pc - 1) / pc);
    const int grid = (int)std::min<long long>(items, slots);
    hipLaunchKernelGGL(row_pass_pf_kernel<R>, dim3(grid), dim3(256), lds, h->stream, j2);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}

template <int R>
int launch_col_fast_r(msl_handle* h, const ColJob& job, int kind) {
    constexpr int N = R * R, CS = R * (R + 1) + 1;
    const size_t lds = ((size_t)2 * N + (size_t)16 * CS) * 8;
    const long long tiles = (long long)(job.ny / 16) * job.n_images;
    const int per_cu = std::max(1, (int)((size_t)h->lds_limit / lds));
    const int grid = (int)std::min<long long>(tiles, (long long)h->n_cus * std::min(per_cu, 2));
    hipLaunchKernelGGL(col_pass_kernel<R>, dim3(grid), dim3(16 * R), lds, h->stream, job);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}

// column pass of the potential build on the rows kx <= nx/2 of a Hermitian spectrum
template <int R>
int launch_col_herm_r(msl_handle* h, const ColJob& job, int kind) {
    constexpr int N = R * R, CS = R * (R + 1) + 1;
    const size_t lds = ((size_t)2 * N + (size_t)16 * CS) * 8;
    const long long tiles = (long long)(job.ny / 16) * job.n_images;
    const int per_cu = std::max(1, (int)((size_t)h->lds_limit / lds));
    const int grid = (int)std::min<long long>(tiles, (long long)h->n_cus * std::min(per_cu, 2));
    (void)hipFuncSetAttribute((const void*)col_pass_kernel<R, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    hipLaunchKernelGGL((col_pass_kernel<R, 16, true>), dim3(grid), dim3(16 * R), lds, h->stream, job);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}
int launch_col_herm(msl_handle* h, const ColJob& job, int kind) {
    return h->Rx == 32 ? launch_col_herm_r<32>(h, job, kind) : launch_col_herm_r<16>(h, job, kind);
}

int launch_row_fast(msl_handle* h, const RowJob& job, int kind) {
    return h->Ry == 32 ? launch_row_fast_r<32>(h, job, kind) : launch_row_fast_r<16>(h, job, kind);
}
int launch_col_fast(msl_handle* h, const ColJob& job, int kind) {
    return h->Rx == 32 ? launch_col_fast_r<32>(h, job, kind) : launch_col_fast_r<16>(h, job, kind);
}

RowJob row_job(const msl_handle* h, float2* buf, int images, int pitch) {
    RowJob j{};
    j.psi = buf; j.trans = nullptr; j.py = nullptr; j.tw = h->tw4_y;
    j.image_stride = (long long)h->cfg.nx * pitch; j.pitch = pitch; j.nx = h->cfg.nx; j.n_images = images;
    return j;
}
ColJob col_job(const msl_handle* h, const float2* in, float2* out, int images, int in_pitch, int out_pitch) {
    ColJob j{};
    j.in = in; j.out = out; j.px = nullptr; j.tw = h->tw4_x;
    j.in_image_stride = (long long)h->cfg.nx * in_pitch; j.out_image_stride = (long long)h->cfg.nx * out_pitch;
    j.in_pitch = in_pitch; j.out_pitch = out_pitch; j.ny = h->cfg.ny; j.n_images = images; j.flags = 0; j.scale = 1.f;
    return j;
}

int fft2_inplace(msl_handle* h, float2* buf, int images, int dir, float scale, int pitch) {
    int rc;
    if (h->Ry) {
        RowJob r = row_job(h, buf, images, pitch);
        r.do_ifft = dir < 0; r.do_fft = dir > 0;
        if ((rc = launch_row_fast(h, r, K_OTHER))) return rc;
    } else {
        LineArgs r = row_args(h, buf, buf, images, pitch);
        r.fft1 = dir;
        if ((rc = launch_lines(h, h->plan_y, r, K_OTHER))) return rc;
    }
    if (h->Rx) {
        ColJob c = col_job(h, buf, buf, images, pitch, pitch);
        c.flags = dir > 0 ? COL_FWD : COL_INV; c.scale = scale;
        return launch_col_fast(h, c, K_OTHER);
    }
    LineArgs c = col_args(h, buf, buf, images, pitch, pitch);
    c.fft1 = dir; c.scale = scale;
    return launch_lines(h, h->plan_x, c, K_OTHER);
}

// Exit-wave epilogue, second half: FFT along x of the y-transformed exit waves in psi, fftshift of both axes and
// scatter into slot `slot` of the (P, T_local, wx, wy) result (calculators.py:284-290).  With a k-window only the
// columns inside it are transformed and only the rows inside it are stored.
// staged full-resolution windows of `groups` frames x P probes -> binned frame slots slot .. slot+groups-1
int bin_frames(msl_handle* h, int slot, int groups) {
    const msl_config& c = h->cfg;
    const long long total = (long long)h->wpix * c.n_probes * groups;
    hipLaunchKernelGGL(bin_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->bin_stage, h->wf, c.n_probes, groups,
                       c.n_frames, slot, h->wx, h->wy, h->bx, h->by, (long long)h->wpitch);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, K_OTHER);
}

int epilogue_x_pass(msl_handle* h, int slot, int groups = 1) {
    const msl_config& c = h->cfg;
    const int P = c.n_probes * groups;               // image = frame-of-batch * n_probes + probe -> wf[probe][slot + frame]
    const bool binned = h->bin_stage != nullptr;
    // binning: the full-resolution window of every image goes to the staging buffer (image-major), bin_kernel sums it
    // into the frame slots -- 16 B/pixel/(probe, frame) extra against 16 B/pixel/slice-step of the loop
    float2* dst = binned ? h->bin_stage : h->wf + (size_t)slot * h->wpitch;
    const long long out_is = binned ? (long long)h->wx * h->wy : (long long)c.n_frames * h->wpitch;
    const int og = binned ? 1 : groups;               // staged images stay image-major; bin_kernel regroups them by frame
    const bool windowed = (h->wx != c.nx) || (h->wy != c.ny);
    const bool fast_ok = h->Rx && (!windowed || (c.ny % 32 == 0 && h->wy % 32 == 0));
    if (fast_ok) {
        ColJob k = col_job(h, h->psi, dst, P, h->pitch, h->wy);
        k.flags = COL_FWD | COL_SHIFT; k.out_image_stride = out_is;
        if (og > 1) { k.out_group = c.n_probes; k.out_group_stride = (long long)h->wpitch; }
        if (windowed) { k.win_c0 = h->wy0; k.win_nc = h->wy; k.win_x0 = h->wx0; k.win_nx = h->wx; }
        int rc = launch_col_fast(h, k, K_OTHER);
        return (rc || !binned) ? rc : bin_frames(h, slot, groups);
    }
    LineArgs k = col_args(h, h->psi, dst, P, h->pitch, h->wy);
    k.fft1 = +1;
    k.out_is = out_is;
    if (og > 1) { k.group = c.n_probes; k.out_gs = (long long)h->wpitch; }
    k.shift_n = c.nx / 2; k.shift_r = c.ny / 2;
    if (windowed) { k.win_n0 = h->wx0; k.win_nn = h->wx; k.win_r0 = h->wy0; k.win_nr = h->wy; }
    int rc = launch_lines(h, h->plan_x, k, K_OTHER);
    return (rc || !binned) ? rc : bin_frames(h, slot, groups);
}

// ---- one-pass-per-slice path ------------------------------------------------------------------------
// psi0 (P, nx, pitch) -> psi0T (P, ny, pitchT): needed when the first pass of the slice loop runs along x
int transpose_probes(msl_handle* h) {
    const msl_config& c = h->cfg;
    // frame batching: every frame of a batch starts from the same probes -- one copy per frame
    const size_t group = (size_t)c.n_probes * c.nx * h->pitch;
    for (int f = 1; f < h->FB; ++f)
        HIPCHK(h, hipMemcpyAsync(h->psi0 + f * group, h->psi0, group * sizeof(float2), hipMemcpyDeviceToDevice, h->stream));
    if (!h->onepass || !h->need_psi0T) return MSL_OK;       // the first pass runs along y, on layout A
    dim3 grid((c.ny + 31) / 32, (c.nx + 31) / 32, c.n_probes * h->FB);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, h->stream, h->psi0, h->psi0T, c.nx, c.ny, h->pitch, h->pitchT,
                       (long long)c.nx * h->pitch, (long long)c.ny * h->pitchT);
    HIPCHK(h, hipGetLastError());
    return MSL_OK;
}

// slice s is used by a pass along x (needs t transposed) iff its distance to the last slice is odd
inline bool slice_is_transposed(const msl_handle* h, int s) {
    if (!h->onepass) return false;
    return h->scheme_b ? ((s & 1) != 0) : (((h->cfg.nz - 1 - s) & 1) != 0);
}

// natural t (nz,nx,ny) -> transposed copies of the odd-distance slices in transT (upload / set_beam paths)
int transpose_odd_slices(msl_handle* h) {
    if (!h->onepass) return MSL_OK;
    const msl_config& c = h->cfg;
    const size_t npix = (size_t)c.nx * c.ny;
    // the transposed slices are every second one: a single launch with the slice index on grid.z
    const int first = slice_is_transposed(h, 0) ? 0 : 1;
    const int count = (c.nz - first + 1) / 2;
    for (int z0 = 0; z0 < count; z0 += 65535) {
        const int nzb = std::min(65535, count - z0);
        dim3 grid((c.ny + 31) / 32, (c.nx + 31) / 32, nzb);
        const size_t off = (size_t)h->cur_batch * c.nz * npix + (size_t)(first + 2 * z0) * npix;
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, h->stream, h->trans + off, h->transT + off, c.nx, c.ny,
                           c.ny, c.nx, (long long)(2 * npix), (long long)(2 * npix));
    }
    HIPCHK(h, hipGetLastError());
    return MSL_OK;
}

// transposing pass on R^2-point lines (1024 / 256), rowt_pass.h; the kernels live in slice_pass.hip
template <int R>
int launch_rowT_r(msl_handle* h, RowTJob job, int kind) {
    constexpr int LINES = 16;
    const size_t lds = rowT_lds_bytes(R);
    // R = 32: ~235 VGPRs, 152 KB -> one workgroup per CU; R = 16: 138 VGPRs, 41 KB -> three
    const int cap = (R == 16) ? 3 : 2;
    const int per_cu = std::max(1, std::min(cap, (int)((size_t)h->lds_limit / lds)));
    const long long slots = (long long)h->n_cus * per_cu;
    const long long lb = job.n_lines / LINES;
    int pc = choose_pchunk(lb, job.n_images, slots, job.t_group);
    if (h->row_pchunk > 0) { pc = std::min(h->row_pchunk, job.n_images); if (job.t_group > 0) while (job.t_group % pc) --pc; }
    job.pchunk = pc;
    const long long items = lb * ((job.n_images + pc - 1) / pc);
    const int grid = (int)std::min<long long>(items, slots);
    if (!rowT_launch(R, job, grid, (size_t)h->lds_limit, h->stream)) return fail(h, MSL_ERR_STATE, "transposing pass: no kernel for flags %d", job.flags);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}

// lines of 2 R^2 = 512 points
template <bool IN_P, bool OUT_P>
int launch_rowT2_io(msl_handle* h, RowTJob job, int kind) {
    constexpr int R = 16, N2 = R * R, N = 2 * N2;
    const size_t lds = ((size_t)2 * N2 + N + (size_t)16 * (N + 2)) * 8;
    const int per_cu = std::max(1, std::min(2, (int)((size_t)h->lds_limit / lds)));
    const long long slots = (long long)h->n_cus * per_cu;
    const long long lb = job.n_lines / 16;
    int pc = choose_pchunk(lb, job.n_images, slots, job.t_group);
    if (h->row_pchunk > 0) { pc = std::min(h->row_pchunk, job.n_images); if (job.t_group > 0) while (job.t_group % pc) --pc; }
    job.pchunk = pc;
    const long long items = lb * ((job.n_images + pc - 1) / pc);
    const int grid = (int)std::min<long long>(items, slots);
    (void)hipFuncSetAttribute((const void*)rowT2_pass_kernel<R, IN_P, OUT_P>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    hipLaunchKernelGGL((rowT2_pass_kernel<R, IN_P, OUT_P>), dim3(grid), dim3(16 * R), lds, h->stream, job);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}
int launch_rowT2(msl_handle* h, const RowTJob& job, int kind) {
    const bool in_p = job.flags & P2_IN_PAIRED, out_p = job.flags & P2_OUT_PAIRED;
    if (in_p) return out_p ? launch_rowT2_io<true, true>(h, job, kind) : launch_rowT2_io<true, false>(h, job, kind);
    return out_p ? launch_rowT2_io<false, true>(h, job, kind) : launch_rowT2_io<false, false>(h, job, kind);
}

// lines of any length <= R^2/2: zero-padded cyclic convolution on the register FFTs
template <int R>
int launch_rowTB_r(msl_handle* h, RowTJob job, int kind) {
    constexpr int M = R * R, NH = M / 2, CS = R * (R + 1) + 2;
    const size_t lds = ((size_t)M + NH + 2 + (size_t)16 * CS) * 8;
    const int per_cu = std::max(1, std::min(R == 16 ? 4 : 1, (int)((size_t)h->lds_limit / lds)));
    const long long slots = (long long)h->n_cus * per_cu;
    const long long lb = (job.n_lines + 15) / 16;
    int pc = choose_pchunk(lb, job.n_images, slots, job.t_group);
    if (h->row_pchunk > 0) { pc = std::min(h->row_pchunk, job.n_images); if (job.t_group > 0) while (job.t_group % pc) --pc; }
    job.pchunk = pc;
    const long long items = lb * ((job.n_images + pc - 1) / pc);
    const int grid = (int)std::min<long long>(items, slots);
    (void)hipFuncSetAttribute((const void*)rowTB_pass_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    hipLaunchKernelGGL((rowTB_pass_kernel<R>), dim3(grid), dim3(16 * R), lds, h->stream, job);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}

// lines of 513..1024 points: the same convolution on the wave-per-line 2048-point register FFT
template <bool IN_P, bool OUT_P>
int launch_rowTB2_io(msl_handle* h, RowTJob job, int kind) {
    constexpr int M = 2048, NH = M / 2, RS = (32 * W2K_PITCH) / 2 + 1;
    const size_t lds = ((size_t)M + 64 + NH + 2 + (size_t)8 * RS) * 8;
    const long long slots = h->n_cus;
    const long long lb = (job.n_lines + 7) / 8;
    int pc = choose_pchunk(lb, job.n_images, slots, job.t_group);
    if (h->row_pchunk > 0) { pc = std::min(h->row_pchunk, job.n_images); if (job.t_group > 0) while (job.t_group % pc) --pc; }
    job.pchunk = pc;
    const long long items = lb * ((job.n_images + pc - 1) / pc);
    const int grid = (int)std::min<long long>(items, slots);
    (void)hipFuncSetAttribute((const void*)rowTB2_pass_kernel<IN_P, OUT_P>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    hipLaunchKernelGGL((rowTB2_pass_kernel<IN_P, OUT_P>), dim3(grid), dim3(512), lds, h->stream, job);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}

// 2048-point lines, one wave per line (fft2048_wave)
template <bool IN_P, bool OUT_P>
int launch_rowTW_io(msl_handle* h, RowTJob job, int kind) {
    constexpr int N = 2048;
    const size_t lds = ((size_t)N + 64 + N / 2 + 64 + (size_t)8 * (N + 1)) * 8;
    const long long slots = h->n_cus;
    const long long lb = job.n_lines / 8;
    int pc = choose_pchunk(lb, job.n_images, slots, job.t_group);
    if (h->row_pchunk > 0) { pc = std::min(h->row_pchunk, job.n_images); if (job.t_group > 0) while (job.t_group % pc) --pc; }
    job.pchunk = pc;
    const long long items = lb * ((job.n_images + pc - 1) / pc);
    const int grid = (int)std::min<long long>(items, slots);
    (void)hipFuncSetAttribute((const void*)rowTW_pass_kernel<IN_P, OUT_P>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    hipLaunchKernelGGL((rowTW_pass_kernel<IN_P, OUT_P>), dim3(grid), dim3(512), lds, h->stream, job);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}
int launch_rowTW(msl_handle* h, const RowTJob& job, int kind) {
    const bool in_p = job.flags & P2_IN_PAIRED, out_p = job.flags & P2_OUT_PAIRED;
    if (in_p) return out_p ? launch_rowTW_io<true, true>(h, job, kind) : launch_rowTW_io<true, false>(h, job, kind);
    return out_p ? launch_rowTW_io<false, true>(h, job, kind) : launch_rowTW_io<false, false>(h, job, kind);
}

// lines of a smooth length A * B (A, B <= 32; G = 16 / 32 lanes per line) or 2 A * B (G = 64: one wave per line, tiles of 8 lines):
// direct mixed-radix transform (rowtm_pass.h)
int launch_rowTM(msl_handle* h, const msl_handle::OpDir& o, RowTJob job, int kind) {
    const int n = (&o == &h->opx) ? h->cfg.nx : h->cfg.ny;
    int A = 0, B = 0, G = 0;
    if (!rowTM_factors(n, &A, &B, &G)) return fail(h, MSL_ERR_STATE, "mixed-radix pass: no kernel for %d points", n);
    const size_t lds = G == 64 ? rowTM2_lds_bytes(A, B) : rowTM_lds_bytes(A, B);
    if (lds > (size_t)h->lds_limit) return fail(h, MSL_ERR_STATE, "mixed-radix pass: %zu bytes of LDS for %d points", lds, n);
    // ~230 VGPRs: two waves per SIMD, i.e. one workgroup of 512 threads or two of 256 per CU
    const int per_cu = std::max(1, std::min(G == 16 ? 2 : 1, (int)((size_t)h->lds_limit / lds)));
    const long long slots = (long long)h->n_cus * per_cu;
    const int lines = G == 64 ? 8 : 16;
    const long long lb = (job.n_lines + lines - 1) / lines;
    int pc = choose_pchunk(lb, job.n_images, slots, job.t_group);
    if (h->row_pchunk > 0) { pc = std::min(h->row_pchunk, job.n_images); if (job.t_group > 0) while (job.t_group % pc) --pc; }
    job.pchunk = pc;
    job.tw = o.mtw; job.flags &= ~(P2_IN_PAIRED | P2_OUT_PAIRED);
    const long long items = lb * ((job.n_images + pc - 1) / pc);
    const int grid = (int)std::min<long long>(items, slots);
    if (!rowTM_launch(n, job, grid, (size_t)h->lds_limit, h->stream)) return fail(h, MSL_ERR_STATE, "mixed-radix pass: no kernel for %d points", n);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}

// one transposing pass along direction `o`: register kernels for R^2 and 2 R^2 points, else the generic LDS kernel
// running the same program (fft, x P, ifft, x t, fft, x P, ifft) with a transposing store
int launch_rowT_dir(msl_handle* h, const msl_handle::OpDir& o, RowTJob job, int kind) {
    if (o.mixed) return launch_rowTM(h, o, job, kind);
    if (o.generic) {
        const FftPlan& pl = (&o == &h->opx) ? h->plan_x : h->plan_y;
        LineArgs a;
        a.in = job.in; a.out = job.out;
        a.n_lines = (long long)job.n_images * job.n_lines; a.lines_per_image = job.n_lines;
        a.in_es = 1; a.in_ls = job.in_pitch; a.in_is = job.in_image_stride;
        a.out_es = job.out_pitch; a.out_ls = 1; a.out_is = job.out_image_stride;
        a.contiguous_lines = 0; a.out_contiguous = 1;
        a.m1_ls = pl.N;
        if (job.t_group > 0) { a.group = job.t_group; a.m_gs = job.t_stride; a.out_gs = (long long)job.t_group * job.out_image_stride; }
        int n = 0;
        if (job.flags & P2_PRE_A) {
            a.fft[n] = +1; a.mkind[n] = MUL_VEC; a.mul[n] = job.pl; ++n;
            a.fft[n] = -1; a.mkind[n] = MUL_ARRAY; a.mul[n] = job.trans; ++n;
        } else {
            a.fft[n] = 0; a.mkind[n] = MUL_ARRAY; a.mul[n] = job.trans; ++n;
        }
        if (job.flags & P2_POST_A) {
            a.fft[n] = +1; a.mkind[n] = MUL_VEC; a.mul[n] = job.pl; ++n;
            a.fft[n] = -1; a.mkind[n] = MUL_NONE; a.mul[n] = nullptr; ++n;
        }
        a.n_steps = n;
        return launch_lines(h, pl, a, kind);
    }
    job.tw = o.tw;
    if (o.wave2k) { job.tw2 = o.tw2; return launch_rowTW(h, job, kind); }
    if (!(o.two || (o.R && !o.breg && !o.breg2 && !o.breg4))) job.flags &= ~(P2_IN_PAIRED | P2_OUT_PAIRED);    // (only the 256 / 512 / 1024-point kernels read and write the interleaved order)
    if (o.breg4) {                          // 1025 .. 2047 points: cyclic convolution of length 4096, two waves per line
        job.n_line = (&o == &h->opx) ? h->cfg.nx : h->cfg.ny;
        job.tw2 = o.tw2; job.bf = o.qf; job.bw = o.bw; job.pl = nullptr;
        constexpr int RS = (32 * W2K_PITCH) / 2 + 1;
        const size_t lds = ((size_t)2048 + 64 + 2048 + 2052 + (size_t)8 * RS) * 8;
        job.pchunk = 1;
        const long long items2 = (long long)((job.n_lines + 3) / 4) * job.n_images;
        const int grid2 = (int)std::min<long long>(items2, (long long)h->n_cus);
        (void)hipFuncSetAttribute((const void*)rowTC2_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
        hipLaunchKernelGGL(rowTC2_pass_kernel, dim3(grid2), dim3(512), lds, h->stream, job);
        HIPCHK(h, hipGetLastError());
        return mark_launch(h, kind);
    }
    if (o.breg || o.breg2) {                // A as one cyclic convolution of length M (two FFTs); bf = its filter
        job.n_line = (&o == &h->opx) ? h->cfg.nx : h->cfg.ny;
        job.pl = nullptr; job.bf = o.qf; job.bw = nullptr;
        if (o.breg2) { job.tw2 = o.tw2; return launch_rowTB2_io<false, false>(h, job, kind); }
        return o.R == 32 ? launch_rowTB_r<32>(h, job, kind) : launch_rowTB_r<16>(h, job, kind);
    }
    if (o.two) {
        job.tw2 = o.tw2; job.pl = o.ptab;
        return launch_rowT2(h, job, kind);
    }
    return o.R == 32 ? launch_rowT_r<32>(h, job, kind) : launch_rowT_r<16>(h, job, kind);
}

// Slice loop when a grid length is 2 R^2: every pass transposes (there is no in-place kernel for those lengths).
// Pass k runs along y for even k and along x for odd k; after an odd number of slices one transpose brings the
// waves back to layout A, and the exit FFT is the stand-alone two-pass one.
int slice_loop_onepass_b(msl_handle* h, int fused_slot, int groups, int first_group) {
    const msl_config& c = h->cfg;
    const int P = c.n_probes * groups, nz = c.nz;           // images of this run: frames of the batch x probes
    const size_t toff = (size_t)first_group * c.nz * c.nx * c.ny;
    const size_t npix = (size_t)c.nx * c.ny;
    const long long isA = (long long)c.nx * h->pitch, isB = (long long)c.ny * h->pitchT;
    const bool fused = fused_slot >= 0;
    int rc;
    if ((rc = begin_timed(h, nz + 4))) return rc;
    for (int k = 0; k < nz; ++k) {
        RowTJob j{};
        j.flags = (k > 0 ? P2_PRE_A : 0) | (k < nz - 1 ? P2_POST_A : 0);
        if (h->debug_flags_mask >= 0) j.flags &= h->debug_flags_mask;
        j.n_images = P;
        if (groups > 1) { j.t_group = c.n_probes; j.t_magic = (unsigned)((1ull << 32) / (unsigned)c.n_probes + 1); j.t_stride = (long long)c.nz * npix; }
        if (h->opx.wave2k && h->opy.wave2k)               // work buffers between two of these passes: paired-lines layout
            j.flags |= (k > 0 ? P2_IN_PAIRED : 0) | (k < nz - 1 ? P2_OUT_PAIRED : 0);
        // 512-point lines next to 512 / 256 / 1024-point ones: interleaved line order between two passes (16-byte loads)
        auto il_kernel = [](const msl_handle::OpDir& o) { return o.R && !o.generic && !o.breg && !o.breg2 && !o.breg4 && !o.wave2k; };
        if (il_kernel(h->opx) && il_kernel(h->opy) && !dbg_env("MSL_NO_INTERLEAVE") && h->debug_flags_mask < 0) {
            j.flags |= (k > 0 ? P2_IN_PAIRED : 0) | (k < nz - 1 ? P2_OUT_PAIRED : 0);
            j.perm_shift = (((k & 1) ? h->opy : h->opx).R == 32) ? 2 : 1;       // radix of the kernel that reads this pass's output
        }
        if (!(k & 1)) {
            j.in = (k == 0) ? h->psi0 : h->psi; j.out = h->psiT;
            j.trans = h->trans + toff + (size_t)k * npix; j.pl = h->pyt;
            j.in_image_stride = isA; j.out_image_stride = isB; j.in_pitch = h->pitch; j.out_pitch = h->pitchT; j.n_lines = c.nx;
            rc = launch_rowT_dir(h, h->opy, j, K_ROW);
        } else {
            j.in = h->psiT; j.out = h->psi;
            j.trans = h->transT + toff + (size_t)k * npix; j.pl = h->pxt;
            j.in_image_stride = isB; j.out_image_stride = isA; j.in_pitch = h->pitchT; j.out_pitch = h->pitch; j.n_lines = c.ny;
            rc = launch_rowT_dir(h, h->opx, j, K_COL);
        }
        if (rc) return rc;
    }
    if (nz & 1) {
        dim3 grid((c.nx + 31) / 32, (c.ny + 31) / 32, P);
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, h->stream, h->psiT, h->psi, c.ny, c.nx, h->pitchT, h->pitch, isB, isA);
        HIPCHK(h, hipGetLastError());
        if ((rc = mark_launch(h, K_OTHER))) return rc;
    }
    if (fused) {
        if (h->Ry) {
            RowJob r = row_job(h, h->psi, P, h->pitch);
            r.do_fft = true;
            if ((rc = launch_row_fast(h, r, K_OTHER))) return rc;
        } else {
            LineArgs r = row_args(h, h->psi, h->psi, P, h->pitch);
            r.fft1 = +1;
            if ((rc = launch_lines(h, h->plan_y, r, K_OTHER))) return rc;
        }
        if ((rc = epilogue_x_pass(h, fused_slot, groups))) return rc;
    }
    h->cur = nullptr;
    h->ctr.slice_steps += (uint64_t)P * nz;
    h->ctr.frames += groups;
    h->ctr.algorithmic_bytes += (uint64_t)P * nz * 16ull * npix + (uint64_t)groups * nz * 8ull * npix + (fused ? (uint64_t)P * 16ull * npix : 0ull);
    return MSL_OK;
}

template <int R>
int launch_row2_r(msl_handle* h, Row2Job job, int kind) {
    constexpr int N = R * R, G = 256 / R;
    const size_t lds = (size_t)N * 16 + (size_t)G * R * (R + 1) * 4;
    const int per_cu = std::max(1, std::min(2, (int)((size_t)h->lds_limit / lds)));
    const long long slots = (long long)h->n_cus * per_cu;
    const long long xg = job.nx / G;
    int pc = choose_pchunk(xg, job.n_images, slots, job.t_group);
    if (h->row_pchunk > 0) { pc = std::min(h->row_pchunk, job.n_images); if (job.t_group > 0) while (job.t_group % pc) --pc; }
    job.pchunk = pc;
    const long long items = xg * ((job.n_images + pc - 1) / pc);
    const int grid = (int)std::min<long long>(items, slots);
    hipLaunchKernelGGL(row_pass2_kernel<R>, dim3(grid), dim3(256), lds, h->stream, job);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, kind);
}

// Slice loop with one HBM pass per slice (see fft_pow2.h).  Pass k runs along y when its distance to the last
// slice is even, else along x; all passes but the last write transposed, the last one is in place on layout A.
int slice_loop_onepass(msl_handle* h, int fused_slot, int groups, int first_group) {
    if (h->scheme_b) return slice_loop_onepass_b(h, fused_slot, groups, first_group);
    const msl_config& c = h->cfg;
    const int P = c.n_probes * groups, nz = c.nz;
    const size_t toff = (size_t)first_group * c.nz * c.nx * c.ny;
    const size_t npix = (size_t)c.nx * c.ny;
    const long long isA = (long long)c.nx * h->pitch, isB = (long long)c.ny * h->pitchT;
    const bool fused = fused_slot >= 0;
    int rc;
    if (nz == 1)
        HIPCHK(h, hipMemcpyAsync(h->psi, h->psi0, (size_t)P * isA * sizeof(float2), hipMemcpyDeviceToDevice, h->stream));
    if ((rc = begin_timed(h, nz + 2))) return rc;
    for (int k = 0; k < nz; ++k) {
        const bool last = (k == nz - 1);
        int flags = (k > 0 ? P2_PRE_A : 0) | (!last ? P2_POST_A : 0) | ((last && fused) ? P2_POST_F : 0);
        if (h->debug_flags_mask >= 0) flags &= h->debug_flags_mask;     // timing experiments only (MSL_DEBUG_FLAGS_MASK)
        if (last) {
            Row2Job j{};
            j.psi = h->psi; j.trans = h->trans + toff + (size_t)k * npix; j.py = h->pyt; j.tw = h->tw4_y;
            j.image_stride = isA; j.pitch = h->pitch; j.nx = c.nx; j.n_images = P; j.flags = flags;
            if (groups > 1) { j.t_group = c.n_probes; j.t_magic = (unsigned)((1ull << 32) / (unsigned)c.n_probes + 1); j.t_stride = (long long)c.nz * npix; }
            rc = h->Ry == 32 ? launch_row2_r<32>(h, j, K_ROW) : launch_row2_r<16>(h, j, K_ROW);
            if (rc) return rc;
            break;
        }
        const bool along_y = !slice_is_transposed(h, k);
        RowTJob j{};
        // between two transposing passes the work buffer is in the interleaved line order (16-byte loads in the reader): the first
        // pass reads the probes, the last transposing pass (k = nz - 2) writes for the in-place pass, both in natural order
        if (!dbg_env("MSL_NO_INTERLEAVE") && h->debug_flags_mask < 0) flags |= (k > 0 ? P2_IN_PAIRED : 0) | (k < nz - 2 ? P2_OUT_PAIRED : 0);
        j.flags = flags; j.n_images = P;
        j.perm_shift = ((along_y ? h->Rx : h->Ry) == 32) ? 2 : 1;             // radix of the kernel that reads this pass's output: log2(R' / 8)
        if (groups > 1) { j.t_group = c.n_probes; j.t_magic = (unsigned)((1ull << 32) / (unsigned)c.n_probes + 1); j.t_stride = (long long)c.nz * npix; }
        if (along_y) {
            j.in = (k == 0) ? h->psi0 : h->psi; j.out = h->psiT;
            j.trans = h->trans + toff + (size_t)k * npix; j.pl = h->pyt; j.tw = h->tw4_y;
            j.in_image_stride = isA; j.out_image_stride = isB; j.in_pitch = h->pitch; j.out_pitch = h->pitchT; j.n_lines = c.nx;
            rc = h->Ry == 32 ? launch_rowT_r<32>(h, j, K_ROW) : launch_rowT_r<16>(h, j, K_ROW);
        } else {
            j.in = (k == 0) ? h->psi0T : h->psiT; j.out = h->psi;
            j.trans = h->transT + toff + (size_t)k * npix; j.pl = h->pxt; j.tw = h->tw4_x;
            j.in_image_stride = isB; j.out_image_stride = isA; j.in_pitch = h->pitchT; j.out_pitch = h->pitch; j.n_lines = c.ny;
            rc = h->Rx == 32 ? launch_rowT_r<32>(h, j, K_COL) : launch_rowT_r<16>(h, j, K_COL);
        }
        if (rc) return rc;
    }
    if (fused && (rc = epilogue_x_pass(h, fused_slot, groups))) return rc;
    h->cur = nullptr;
    h->ctr.slice_steps += (uint64_t)P * nz;
    h->ctr.frames += groups;
    h->ctr.algorithmic_bytes += (uint64_t)P * nz * 16ull * npix + (uint64_t)groups * nz * 8ull * npix + (fused ? (uint64_t)P * 16ull * npix : 0ull);
    return MSL_OK;
}

// n atoms of a frame group; `rows` rows of the sorted phase tables (the atoms plus the padding of every bin to SF_ALIGN rows)
int ensure_atoms(msl_handle* h, size_t n, size_t rows) {
    if (rows <= h->atom_cap) return MSL_OK;
    (void)n;
    size_t cap = std::max<size_t>(rows, h->atom_cap * 3 / 2 + 1024);
    int rc;
    if ((rc = dalloc(h, &h->d_pos, cap * 3))) return rc;
    if ((rc = dalloc(h, &h->d_Z, cap))) return rc;
    if ((rc = dalloc(h, &h->d_key, cap))) return rc;
    if ((rc = dalloc(h, &h->d_order, cap))) return rc;
    if ((rc = dalloc(h, &h->d_u1, cap))) return rc;
    if ((rc = dalloc(h, &h->d_u2, cap))) return rc;
    // phase tables: the quadrant kernel reads the columns 0 .. n/2 only
    if ((rc = dalloc(h, &h->d_ex, cap * (size_t)(h->cfg.nx / 2 + 1)))) return rc;
    if ((rc = dalloc(h, &h->d_ey, cap * (size_t)(h->cfg.ny / 2 + 1)))) return rc;
    h->atom_cap = cap;
    return MSL_OK;
}

// The slice loop (generic kernels).  fused_slot < 0: leave real-space exit waves in psi.
int slice_loop(msl_handle* h, int fused_slot, int groups, int first_group) {
    if (h->onepass) return slice_loop_onepass(h, fused_slot, groups, first_group);
    const msl_config& c = h->cfg;
    const int P = c.n_probes * groups, nz = c.nz;
    const size_t npix = (size_t)c.nx * c.ny;
    const size_t toff = (size_t)first_group * c.nz * npix;
    HIPCHK(h, hipMemcpyAsync(h->psi, h->psi0, (size_t)P * c.nx * h->pitch * sizeof(float2), hipMemcpyDeviceToDevice, h->stream));
    int rc = begin_timed(h, 2 * nz + 2);
    if (rc) return rc;
    const bool fused = fused_slot >= 0;
    for (int z = 0; z < nz; ++z) {
        const bool last = (z == nz - 1);
        if (h->Ry) {
            RowJob r = row_job(h, h->psi, P, h->pitch);
            r.do_ifft = z > 0; r.trans = h->trans + toff + (size_t)z * npix;
            r.do_fft = (!last || fused); r.py = last ? nullptr : h->pyt;
            if (groups > 1) { r.t_group = c.n_probes; r.t_magic = (unsigned)((1ull << 32) / (unsigned)c.n_probes + 1); r.t_stride = (long long)c.nz * npix; }
            if ((rc = launch_row_fast(h, r, K_ROW))) return rc;
        } else {
            LineArgs r = row_args(h, h->psi, h->psi, P, h->pitch);
            r.fft1 = (z > 0) ? -1 : 0;
            r.m1_kind = MUL_ARRAY; r.m1 = h->trans + toff + (size_t)z * npix; r.m1_ls = c.ny;
            if (groups > 1) { r.group = c.n_probes; r.m_gs = (long long)c.nz * npix; r.out_gs = (long long)c.n_probes * r.out_is; }
            if (!last) { r.fft2 = +1; r.m2_kind = MUL_VEC; r.m2 = h->pyt; }
            else if (fused) { r.fft2 = +1; }
            if ((rc = launch_lines(h, h->plan_y, r, K_ROW))) return rc;
        }
        if (!last) {
            if (h->Rx) {
                ColJob k = col_job(h, h->psi, h->psi, P, h->pitch, h->pitch);
                k.px = h->pxt; k.flags = COL_FWD | COL_MULPX | COL_INV;
                if ((rc = launch_col_fast(h, k, K_COL))) return rc;
            } else {
                LineArgs k = col_args(h, h->psi, h->psi, P, h->pitch, h->pitch);
                k.fft1 = +1; k.m1_kind = MUL_VEC; k.m1 = h->pxt; k.fft2 = -1;
                if ((rc = launch_lines(h, h->plan_x, k, K_COL))) return rc;
            }
        }
    }
    if (fused) {
        // epilogue: fft along x, fftshift both axes, scatter into (P, T_local, nx, ny)
        if ((rc = epilogue_x_pass(h, fused_slot, groups))) return rc;
    }
    h->cur = nullptr;
    h->ctr.slice_steps += (uint64_t)P * nz;
    h->ctr.frames += groups;
    h->ctr.algorithmic_bytes += (uint64_t)P * nz * 32ull * npix + (uint64_t)groups * nz * 8ull * npix + (fused ? (uint64_t)P * 16ull * npix : 0ull);
    return MSL_OK;
}

// Fresnel propagator, separable: P[kx,ky] = exp(-i pi lambda dz kx^2) * exp(-i pi lambda dz ky^2)
// (multislice.py:273-275), with the 1/(nx ny) of the inverse FFT folded in.
int fill_propagator(msl_handle* h) {
    const msl_config& c = h->cfg;
    auto fill = [&](float2* dst, int n, double d) -> int {
        std::vector<float2> v(n);
        for (int m = 0; m < n; ++m) {
            int f = (m < (n + 1) / 2) ? m : m - n;
            double k = f * (1.0 / (n * d));
            double ph = -M_PI * c.wavelength * c.dz * k * k;
            v[m] = make_float2((float)(cos(ph) / n), (float)(sin(ph) / n));
        }
        HIPCHK(h, hipMemcpyAsync(dst, v.data(), n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return MSL_OK;
    };
    int rc = fill(h->pxt, c.nx, c.dx);
    if (rc) return rc;
    if ((rc = fill(h->pyt, c.ny, c.dy))) return rc;
    // split-order copies for the 2R^2 kernels: entry [b*R^2 + k] = P[2k + b]
    auto fill_split = [&](float2* dst, int n, double d) -> int {
        if (!dst) return MSL_OK;
        std::vector<float2> v(n);
        for (int b = 0; b < 2; ++b)
            for (int k = 0; k < n / 2; ++k) {
                const int m = 2 * k + b;
                const int f = (m < (n + 1) / 2) ? m : m - n;
                const double kk = f * (1.0 / (n * d));
                const double ph = -M_PI * c.wavelength * c.dz * kk * kk;
                v[b * (n / 2) + k] = make_float2((float)(cos(ph) / n), (float)(sin(ph) / n));
            }
        HIPCHK(h, hipMemcpyAsync(dst, v.data(), n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return MSL_OK;
    };
    if (h->opx.two && (rc = fill_split(h->opx.ptab, c.nx, c.dx))) return rc;
    if (h->opy.two && (rc = fill_split(h->opy.ptab, c.ny, c.dy))) return rc;
    // any-length register kernels: filter of the zero-padded cyclic convolution that IS the propagation along one axis
    auto fill_padded = [&](const msl_handle::OpDir& o, int n, double d) -> int {
        const int NH = o.breg2 ? 1024 : o.R * o.R / 2;
        // convolution form: a = ifft_n(P) (float64), wrapped to the cyclic length M = 2 NH, filter = FFT_M(q) / M, first half + 1
        {
            const int M = 2 * NH;
            std::vector<double> pr(n), pi(n), er(n), ei(n), qr(M, 0.0), qi(M, 0.0);
            for (int m = 0; m < n; ++m) {
                const int f = (m < (n + 1) / 2) ? m : m - n;
                const double k = f * (1.0 / (n * d));
                const double ph = -M_PI * c.wavelength * c.dz * k * k;
                pr[m] = cos(ph); pi[m] = sin(ph);
                const double a = 2.0 * M_PI * (double)m / (double)n;
                er[m] = cos(a); ei[m] = sin(a);
            }
            for (int j = 0; j < n; ++j) {                      // a[j] = (1/n) sum_m P[m] e^{+2 pi i m j / n}
                double sr = 0.0, si = 0.0;
                long long t = 0;
                for (int m = 0; m < n; ++m) {
                    sr += pr[m] * er[t] - pi[m] * ei[t];
                    si += pr[m] * ei[t] + pi[m] * er[t];
                    t += j; if (t >= n) t -= n;
                }
                sr /= n; si /= n;
                qr[j] = sr; qi[j] = si;                         // lag +j
                if (j) { qr[M - n + j] = sr; qi[M - n + j] = si; }   // lag j - n  (M - (n - j))
            }
            host_fft_pow2(qr, qi);
            std::vector<float2> qf(NH + 2, make_float2(0.f, 0.f));
            for (int j = 0; j <= NH; ++j) qf[j] = make_float2((float)(qr[j] / M), (float)(qi[j] / M));
            HIPCHK(h, hipMemcpyAsync(o.qf, qf.data(), (NH + 2) * sizeof(float2), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        return MSL_OK;
    };
    if ((h->opx.breg || h->opx.breg2) && (rc = fill_padded(h->opx, c.nx, c.dx))) return rc;
    if ((h->opy.breg || h->opy.breg2) && (rc = fill_padded(h->opy, c.ny, c.dy))) return rc;
    // 1025..2047 points: filter of the cyclic convolution of length 4096 in the split order of rowTC_pass_kernel
    auto fill_conv4 = [&](const msl_handle::OpDir& o, int n, double d) -> int {
        const int M = 4096;
        std::vector<double> pr(n), pi(n), er(n), ei(n), qr(M, 0.0), qi(M, 0.0);
        for (int m = 0; m < n; ++m) {
            const int f = (m < (n + 1) / 2) ? m : m - n;
            const double k = f * (1.0 / (n * d));
            const double ph = -M_PI * c.wavelength * c.dz * k * k;
            pr[m] = cos(ph); pi[m] = sin(ph);
            const double a = 2.0 * M_PI * (double)m / (double)n;
            er[m] = cos(a); ei[m] = sin(a);
        }
        for (int j = 0; j < n; ++j) {
            double sr = 0.0, si = 0.0;
            long long t = 0;
            for (int m = 0; m < n; ++m) {
                sr += pr[m] * er[t] - pi[m] * ei[t];
                si += pr[m] * ei[t] + pi[m] * er[t];
                t += j; if (t >= n) t -= n;
            }
            sr /= n; si /= n;
            qr[j] = sr; qi[j] = si;
            if (j) { qr[M - n + j] = sr; qi[M - n + j] = si; }
        }
        host_fft_pow2(qr, qi);
        std::vector<float2> qf(2052, make_float2(0.f, 0.f));
        for (int k = 0; k <= 1024; ++k) qf[k] = make_float2((float)(qr[2 * k] / M), (float)(qi[2 * k] / M));
        for (int k = 0; k < 1024; ++k) qf[1026 + k] = make_float2((float)(qr[2 * k + 1] / M), (float)(qi[2 * k + 1] / M));
        HIPCHK(h, hipMemcpyAsync(o.qf, qf.data(), qf.size() * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return MSL_OK;
    };
    if (h->opx.breg4 && (rc = fill_conv4(h->opx, c.nx, c.dx))) return rc;
    if (h->opy.breg4 && (rc = fill_conv4(h->opy, c.ny, c.dy))) return rc;
    return MSL_OK;
}

}  // namespace

// The per-lane / wave-split time kernels (70 instantiations) are compiled in translation units of their own (tacaw_direct.hip,
// tacaw_split.hip, tacaw_split2.hip: built in parallel with this file); tacaw_launch.h declares their launchers.
static int launch_time_direct(msl_handle* h, const TimeJob& j) {
    if (!time_direct_launch(j, h->n_cus, h->stream)) return fail(h, MSL_ERR_UNSUPPORTED, "no per-lane time kernel for %d frames", j.T);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, K_OTHER);
}
static int launch_time_split(msl_handle* h, TimeJob j) {
    const int T = j.T;
    if (h->tsplit_T != T) {
        h->tsplit_T = 0;
        std::vector<float2> w(T);
        for (int n = 0; n < T; ++n) { const double a = -2.0 * M_PI * (double)n / (double)T; w[n] = make_float2((float)cos(a), (float)sin(a)); }
        int rc = dalloc(h, &h->tsplit_tw, (size_t)T);
        if (rc) return rc;
        HIPCHK(h, hipMemcpy(h->tsplit_tw, w.data(), (size_t)T * sizeof(float2), hipMemcpyHostToDevice));
        h->tsplit_T = T;
    }
    j.tw = h->tsplit_tw;
    if (!time_split_launch(j, h->n_cus, (size_t)h->lds_limit, h->stream)) return fail(h, MSL_ERR_UNSUPPORTED, "no wave-split time kernel for %d frames", T);
    HIPCHK(h, hipGetLastError());
    return mark_launch(h, K_OTHER);
}

extern "C" {

int msl_abi_version(void) { return MSL_ABI_VERSION; }

int msl_line_kernel_class(int32_t n) {
    if (n == 256 || n == 512 || n == 1024 || n == 2048) return 2;
    int A = 0, B = 0, G = 0;
    return rowTM_factors(n, &A, &B, &G) ? 1 : 0;
}

const char* msl_last_error(const msl_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int msl_create(const msl_config* cfg, msl_handle** out) {
    if (!cfg || !out) return fail(nullptr, MSL_ERR_INVALID, "msl_create: null argument");
    *out = nullptr;
    if (cfg->nx < 2 || cfg->ny < 2 || cfg->nz < 1 || cfg->n_probes < 1 || cfg->n_frames < 0)
        return fail(nullptr, MSL_ERR_INVALID, "msl_create: bad grid nx=%d ny=%d nz=%d P=%d T=%d", cfg->nx, cfg->ny, cfg->nz,
                    cfg->n_probes, cfg->n_frames);
    if (!(cfg->dx > 0) || !(cfg->dy > 0) || !(cfg->wavelength > 0))
        return fail(nullptr, MSL_ERR_INVALID, "msl_create: dx, dy, wavelength must be positive");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, MSL_ERR_HIP, "msl_create: no HIP device available (%s)", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, MSL_ERR_INVALID, "msl_create: device %d out of range (have %d)", cfg->device, ndev);
    if (cfg->window_nx < 0 || cfg->window_nx > cfg->nx || cfg->window_ny < 0 || cfg->window_ny > cfg->ny)
        return fail(nullptr, MSL_ERR_INVALID, "msl_create: k-window %d x %d outside the %d x %d grid", cfg->window_nx, cfg->window_ny,
                    cfg->nx, cfg->ny);
    msl_handle* h = new (std::nothrow) msl_handle();
    if (!h) return fail(nullptr, MSL_ERR_NOMEM, "msl_create: out of host memory");
    h->cfg = *cfg;
    h->FB = (cfg->frame_batch > 1 && !cfg->keep_potential) ? cfg->frame_batch : 1;
    // k-window, centred on the DC pixel of the fftshifted spectrum (index n/2): [n/2 - w/2, n/2 - w/2 + w)
    h->wx = cfg->window_nx ? cfg->window_nx : cfg->nx;
    h->wy = cfg->window_ny ? cfg->window_ny : cfg->ny;
    h->wx0 = cfg->nx / 2 - h->wx / 2;
    h->wy0 = cfg->ny / 2 - h->wy / 2;
    h->bx = cfg->bin_nx > 1 ? cfg->bin_nx : 1;
    h->by = cfg->bin_ny > 1 ? cfg->bin_ny : 1;
    if (h->wx % h->bx || h->wy % h->by) {
        const int rc_ = fail(nullptr, MSL_ERR_INVALID, "msl_create: stored spectrum %d x %d is not a multiple of the bin %d x %d", h->wx, h->wy, h->bx, h->by);
        delete h; return rc_;
    }
    h->wpix = (size_t)(h->wx / h->bx) * (h->wy / h->by);
    // every (probe, frame) image of the results starts on a 256-byte boundary: the time kernels read 128-byte row segments of
    // neighbouring pixels of every frame, and with an odd pixel count (501 x 491: 245 991) all but one frame in 16 straddled two lines
    h->wpitch = dbg_env("MSL_NO_RESULT_PITCH") ? h->wpix : ((h->wpix + 31) & ~(size_t)31);
    auto bail = [&](int rc) { g_create_error = h->err; msl_destroy(h); return rc; };
    if (hipSetDevice(cfg->device) != hipSuccess) return bail(fail(h, MSL_ERR_HIP, "hipSetDevice(%d) failed", cfg->device));
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(h, MSL_ERR_HIP, "hipStreamCreate failed"));
    (void)hipFuncSetAttribute((const void*)line_fft_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    (void)hipFuncSetAttribute((const void*)line_fft_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    (void)hipFuncSetAttribute((const void*)line_fft_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    (void)hipFuncSetAttribute((const void*)line_fft_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    (void)hipFuncSetAttribute((const void*)line_fft_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    int rc;
    {   // the lane <-> register exchange of every register kernel rests on inline asm the compiler cannot check: test it once per process
        static int exchange_ok = -1;
        if (exchange_ok < 0) exchange_ok = rowT_selftest(h->stream);
        if (exchange_ok != 0) return bail(fail(h, MSL_ERR_HIP, "msl_create: the add-tid LDS exchange self-test failed on this device / toolchain (%d values wrong)", exchange_ok));
    }
    if ((rc = make_plan(h, h->plan_x, cfg->nx))) return bail(rc);
    if ((rc = make_plan(h, h->plan_y, cfg->ny))) return bail(rc);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cus = prop.multiProcessorCount;
    }
    if (cfg->fft_path == 0) {
        // row kernel: rows of length ny, 256/R rows per workgroup;  column kernel: columns of length nx, 16 per tile
        int ry = fast_radix(cfg->ny), rx = fast_radix(cfg->nx);
        if (ry && cfg->nx % (256 / ry) == 0) { h->Ry = ry; if ((rc = make_tw4(h, &h->tw4_y, ry))) return bail(rc); }
        if (rx && cfg->ny % 16 == 0 && cfg->ny >= 32) { h->Rx = rx; if ((rc = make_tw4(h, &h->tw4_x, rx))) return bail(rc); }
        { const char* e = dbg_env("MSL_ROW_PCHUNK"); if (e) h->row_pchunk = atoi(e); }
        (void)hipFuncSetAttribute((const void*)row_pass_pf_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
        (void)hipFuncSetAttribute((const void*)row_pass_pf_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
        (void)hipFuncSetAttribute((const void*)col_pass_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
        (void)hipFuncSetAttribute((const void*)col_pass_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
    }
    const size_t npix = (size_t)cfg->nx * cfg->ny;
    {
        const char* e = dbg_env("MSL_PITCH_PAD");
        int pad = e ? atoi(e) : 16;
        if (pad < 0 || (pad & 1)) pad = 16;
        h->pitch = cfg->ny + ((h->Rx || h->Ry) ? pad : 0);
    }
    {
        const char* e = dbg_env("MSL_SLICE_PATH");           // 2 = force the two-pass four-step loop
        const bool want = cfg->fft_path == 0 && !(e && atoi(e) == 2);
        // n = line length of the direction, n_other = number of lines per image (the register kernels take 16 at a time)
        auto setup_dir = [&](msl_handle::OpDir& o, int n, int n_other, int Rfast, float2* tw4) -> int {
            const bool lines_ok = (n_other % 16 == 0);
            if (Rfast && lines_ok) { o.R = Rfast; o.two = false; o.tw = tw4; return MSL_OK; }
            const int R2 = (n == 512) ? 16 : (n == 2048 ? 32 : 0);
            const bool two_ok = R2 && lines_ok && want && !dbg_env("MSL_NO_TWO");
            if (n == 2048 && two_ok) {
                // 2048-point lines on the wave-per-line FFT (tables: T[k1*64+n2], W_64; the natural Fresnel table serves as is)
                o.R = 32; o.wave2k = true;
                std::vector<float2> T(2048), W(64);
                for (int k1 = 0; k1 < 32; ++k1)
                    for (int n2 = 0; n2 < 64; ++n2) {
                        const double a = -2.0 * M_PI * (double)(k1 * n2) / 2048.0;
                        T[k1 * 64 + n2] = make_float2((float)cos(a), (float)sin(a));
                    }
                for (int m = 0; m < 32; ++m) {
                    const double a = -2.0 * M_PI * m / 64.0;
                    W[m] = make_float2(1.f, 0.f);
                    W[32 + m] = make_float2((float)cos(a), (float)sin(a));
                }
                int r;
                if ((r = dalloc(h, &o.tw, (size_t)2048))) return r;
                if ((r = dalloc(h, &o.tw2, (size_t)64))) return r;
                if (hipMemcpy(o.tw, T.data(), 2048 * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(o.tw2, W.data(), 64 * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
                    return fail(h, MSL_ERR_HIP, "twiddle upload failed");
                return MSL_OK;
            }
            // smooth lengths A * B (A, B <= 32) or 2 A * B (up to 1728) with a compiled kernel: direct mixed-radix passes in the slice
            // loop (600^2: 163 k -> 330 k slice-steps/s, 1500^2: 18 k -> 55 k).  The tables of the convolution / generic branches below are made all the same: the potential's
            // inverse transform, the probes and the exit FFT of such a grid still run on those kernels.
            {
                int mA = 0, mB = 0, mG = 0;
                if (!two_ok && want && rowTM_factors(n, &mA, &mB, &mG) && !dbg_env("MSL_NO_MIXED") &&
                    (mG == 64 ? rowTM2_lds_bytes(mA, mB) : rowTM_lds_bytes(mA, mB)) <= (size_t)h->lds_limit) {
                    const int lanes1 = mG == 64 ? 2 * mA : mA;           // lanes of layout 1 (rowtm_pass.h)
                    std::vector<float2> T(2 * (size_t)n + (mG == 64 ? 2 * mA : 0));
                    for (int k2 = 0; k2 < mB; ++k2)
                        for (int n1 = 0; n1 < lanes1; ++n1) {
                            const double a = -2.0 * M_PI * (double)((k2 * n1) % n) / (double)n;
                            const float2 w = make_float2((float)cos(a), (float)sin(a));
                            T[k2 * lanes1 + n1] = w;
                            if (mG == 64) T[n + (n1 % mA) * 2 * mB + 2 * k2 + n1 / mA] = w;      // lane order of layout 2: [m 2B + 2 k2 + h], n1 = m + A h
                            else T[n + n1 * mB + k2] = w;
                        }
                    if (mG == 64)
                        for (int m = 0; m < mA; ++m) {
                            const double a = -2.0 * M_PI * (double)m / (double)(2 * mA);
                            T[2 * n + m] = make_float2(1.f, 0.f);
                            T[2 * n + mA + m] = make_float2((float)cos(a), (float)sin(a));
                        }
                    int r;
                    if ((r = dalloc(h, &o.mtw, T.size()))) return r;
                    if (hipMemcpy(o.mtw, T.data(), T.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
                        return fail(h, MSL_ERR_HIP, "twiddle upload failed");
                    o.mixed = true;
                }
            }
            // zero-padded cyclic convolution (or, MSL_CHIRPZ=1, chirp-z) on the register FFTs of length M = R^2 >= 2n - 1.  A line
            // costs the same whatever n is, so against the generic Stockham kernel (cost ~ n log n) it wins for n <= 128 (M = 256)
            // and from n ~ 190 up (M = 1024; 64 probes x 50 slices: 160^2 1.83 M vs 1.72 M slice-steps/s, 200^2 1.23 vs 1.31 M,
            // 240^2 0.84 vs 1.13 M), and everywhere the generic kernel would need its own LDS-resident Bluestein transform
            const bool smooth = ((&o == &h->opx) ? h->plan_x : h->plan_y).M == n;
            if (!two_ok && want && n >= 33 && n <= 512 && (n <= 128 || n >= 192 || !smooth) && !dbg_env("MSL_NO_BLUESTEIN_REG")) {
                const int Rb = (n <= 128) ? 16 : 32, M = Rb * Rb, NH = M / 2;
                o.R = Rb; o.breg = true;
                int r = make_tw4(h, &o.tw, Rb);
                if (r) return r;
                std::vector<float2> bw(NH, make_float2(0.f, 0.f)), bf(NH + 2, make_float2(0.f, 0.f));
                std::vector<double> cr(M, 0.0), ci(M, 0.0);
                for (int i = 0; i < n; ++i) {
                    const long long q = ((long long)i * i) % (2LL * n);
                    const double a = -M_PI * (double)q / (double)n;              // w[i] = exp(-i pi i^2 / n)
                    bw[i] = make_float2((float)cos(a), (float)sin(a));
                    cr[i] = cos(a); ci[i] = -sin(a);                              // conj chirp, wrapped to negative lags
                    if (i) { cr[M - i] = cr[i]; ci[M - i] = ci[i]; }
                }
                host_fft_pow2(cr, ci);
                for (int j = 0; j <= NH; ++j) bf[j] = make_float2((float)(cr[j] / M), (float)(ci[j] / M));
                if ((r = dalloc(h, &o.bw, (size_t)NH))) return r;
                if ((r = dalloc(h, &o.bf, (size_t)NH + 2))) return r;
                if ((r = dalloc(h, &o.qf, (size_t)NH + 2))) return r;
                if (hipMemcpy(o.bw, bw.data(), NH * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(o.bf, bf.data(), (NH + 2) * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
                    return fail(h, MSL_ERR_HIP, "chirp table upload failed");
                return MSL_OK;
            }
            if (!two_ok && want && n >= 513 && n <= 1024 && !dbg_env("MSL_NO_BLUESTEIN_REG")) {
                // 513..1024: the same on the wave-per-line 2048-point register FFT, every length (convolution form against the
                // Stockham kernel: 540^2 102 k -> 186 k slice-steps/s, 600^2 85 k -> 160 k, 768^2 73 k -> 130 k)
                constexpr int M = 2048, NH = 1024;
                o.R = 32; o.breg2 = true;
                std::vector<float2> T(M), W(64), bw(NH, make_float2(0.f, 0.f)), bf(NH + 2, make_float2(0.f, 0.f));
                for (int k1 = 0; k1 < 32; ++k1)
                    for (int n2 = 0; n2 < 64; ++n2) {
                        const double a = -2.0 * M_PI * (double)(k1 * n2) / (double)M;
                        T[k1 * 64 + n2] = make_float2((float)cos(a), (float)sin(a));
                    }
                for (int m = 0; m < 32; ++m) {
                    const double a = -2.0 * M_PI * m / 64.0;
                    W[m] = make_float2(1.f, 0.f);                                   // even lane of a pair: no twiddle
                    W[32 + m] = make_float2((float)cos(a), (float)sin(a));          // odd lane: W_64^m
                }
                std::vector<double> cr(M, 0.0), ci(M, 0.0);
                for (int i = 0; i < n; ++i) {
                    const long long q = ((long long)i * i) % (2LL * n);
                    const double a = -M_PI * (double)q / (double)n;
                    bw[i] = make_float2((float)cos(a), (float)sin(a));
                    cr[i] = cos(a); ci[i] = -sin(a);
                    if (i) { cr[M - i] = cr[i]; ci[M - i] = ci[i]; }
                }
                host_fft_pow2(cr, ci);
                for (int j = 0; j <= NH; ++j) bf[j] = make_float2((float)(cr[j] / M), (float)(ci[j] / M));
                int r;
                if ((r = dalloc(h, &o.tw, (size_t)M))) return r;
                if ((r = dalloc(h, &o.tw2, (size_t)64))) return r;
                if ((r = dalloc(h, &o.bw, (size_t)NH))) return r;
                if ((r = dalloc(h, &o.bf, (size_t)NH + 2))) return r;
                if ((r = dalloc(h, &o.qf, (size_t)NH + 2))) return r;
                if (hipMemcpy(o.tw, T.data(), M * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(o.tw2, W.data(), 64 * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(o.bw, bw.data(), NH * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(o.bf, bf.data(), (NH + 2) * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
                    return fail(h, MSL_ERR_HIP, "chirp table upload failed");
                return MSL_OK;
            }
            if (!two_ok && want && n >= 1025 && n <= 2047 && !dbg_env("MSL_NO_CONV4096") && !dbg_env("MSL_NO_BLUESTEIN_REG")) {
                // 1025..2047: cyclic convolution of length 4096 on pairs of 2048-point wave FFTs, the two branches of the radix-2
                // step on two waves (rowTC2_pass_kernel): 16 probes x 20 slices, slice-steps/s against the generic two-pass loop:
                // 1100^2 19.6 k -> 27.5 k, 1500^2 12.6 k -> 18.3 k, 2000^2 6.8 k -> 11.4 k.  MSL_CONV4096=1 selects the first
                // version, both branches in one wave (rowTC_pass_kernel): two 64-register line sets plus the transform's
                // temporaries spill 916 B per lane and it is no faster than the generic loop (20.0 k at 1100^2).
                constexpr int M = 2048;
                o.R = 32; o.breg4 = true;
                std::vector<float2> T(M), W(64), wq(M);
                for (int k1 = 0; k1 < 32; ++k1)
                    for (int n2 = 0; n2 < 64; ++n2) {
                        const double a = -2.0 * M_PI * (double)(k1 * n2) / (double)M;
                        T[k1 * 64 + n2] = make_float2((float)cos(a), (float)sin(a));
                    }
                for (int m = 0; m < 32; ++m) {
                    const double a = -2.0 * M_PI * m / 64.0;
                    W[m] = make_float2(1.f, 0.f);
                    W[32 + m] = make_float2((float)cos(a), (float)sin(a));
                }
                for (int i = 0; i < M; ++i) {
                    const double a = -2.0 * M_PI * (double)i / 4096.0;
                    wq[i] = make_float2((float)cos(a), (float)sin(a));
                }
                int r;
                if ((r = dalloc(h, &o.tw, (size_t)M))) return r;
                if ((r = dalloc(h, &o.tw2, (size_t)64))) return r;
                if ((r = dalloc(h, &o.bw, (size_t)M))) return r;
                if ((r = dalloc(h, &o.qf, (size_t)2052))) return r;
                if (hipMemcpy(o.tw, T.data(), M * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(o.tw2, W.data(), 64 * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(o.bw, wq.data(), M * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
                    return fail(h, MSL_ERR_HIP, "twiddle upload failed");
                return MSL_OK;
            }
            if (!two_ok) {
                // generic LDS kernel with a transposing store: tiles of >= 8 lines keep the stores at 64 bytes or more
                const int M = (&o == &h->opx) ? h->plan_x.M : h->plan_y.M;
                o.generic = want && M <= 1024 && !dbg_env("MSL_NO_GENERIC_ONEPASS");
                return MSL_OK;
            }
            o.R = R2; o.two = true;
            int r = make_tw4(h, &o.tw, R2);
            if (r) return r;
            std::vector<float2> t(R2 * R2);
            for (int m = 0; m < R2 * R2; ++m) {
                const double a = -2.0 * M_PI * (double)m / (double)n;
                t[m] = make_float2((float)cos(a), (float)sin(a));
            }
            if ((r = dalloc(h, &o.tw2, (size_t)R2 * R2))) return r;
            if (hipMemcpy(o.tw2, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess)
                return fail(h, MSL_ERR_HIP, "twiddle upload failed");
            return dalloc(h, &o.ptab, (size_t)n);
        };
        if ((rc = setup_dir(h->opx, cfg->nx, cfg->ny, h->Rx, h->tw4_x))) return bail(rc);
        if ((rc = setup_dir(h->opy, cfg->ny, cfg->nx, h->Ry, h->tw4_y))) return bail(rc);
        // a chirp / convolution axis next to another kind of axis of at most 1024 points: the other one gets chirp-z tables too, so
        // that the potential's inverse transform runs on the register kernels along both (512 x 300, 349 x 1024 ...)
        {
            auto cz_axis = [](const msl_handle::OpDir& o) { return o.breg || o.breg2; };
            if (want && cz_axis(h->opx) && !cz_axis(h->opy) && cfg->ny >= 33 && cfg->ny <= 1024 && (rc = make_cz_tables(h, h->opy, cfg->ny))) return bail(rc);
            if (want && cz_axis(h->opy) && !cz_axis(h->opx) && cfg->nx >= 33 && cfg->nx <= 1024 && (rc = make_cz_tables(h, h->opx, cfg->nx))) return bail(rc);
        }
        h->onepass = want && (h->opx.R || h->opx.generic) && (h->opy.R || h->opy.generic);
        h->scheme_b = h->onepass && (h->opx.two || h->opy.two || h->opx.generic || h->opy.generic || h->opx.breg || h->opy.breg ||
                                     h->opx.breg2 || h->opy.breg2 || h->opx.breg4 || h->opy.breg4 || h->opx.wave2k || h->opy.wave2k);
        if (h->pitch == cfg->ny && h->onepass) h->pitch = cfg->ny + 16;        // pad the work buffers of 2R^2 grids too
        if (h->onepass && (h->pitch & 1)) ++h->pitch;                          // even pitches: the transposed stores write two lines (16 bytes) at a time
        const size_t images = (size_t)cfg->n_probes * h->FB;
        if ((rc = dalloc(h, &h->psi0, (size_t)cfg->nx * h->pitch * images))) return bail(rc);
        if ((rc = dalloc(h, &h->psi, (size_t)cfg->nx * h->pitch * images))) return bail(rc);
        if (h->onepass) {
            h->pitchT = cfg->nx + 16 + (cfg->nx & 1);
            if (h->pitch - cfg->ny > 16 && !(cfg->nx & 1)) h->pitchT = cfg->nx + (h->pitch - cfg->ny);
            if ((rc = dalloc(h, &h->psiT, (size_t)cfg->ny * h->pitchT * images))) return bail(rc);
            // transposed probes: only when the first pass runs along x (alternating scheme with an even slice count)
            h->need_psi0T = !h->scheme_b && (cfg->nz % 2 == 0);
            if (h->need_psi0T && (rc = dalloc(h, &h->psi0T, (size_t)cfg->ny * h->pitchT * images))) return bail(rc);
            if ((rc = dalloc(h, &h->transT, npix * cfg->nz * h->FB))) return bail(rc);
            { const char* ev = dbg_env("MSL_DEBUG_FLAGS_MASK"); if (ev) h->debug_flags_mask = atoi(ev); }
            (void)hipFuncSetAttribute((const void*)row_pass2_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
            (void)hipFuncSetAttribute((const void*)row_pass2_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
        }
    }
    if ((rc = dalloc(h, &h->trans, npix * cfg->nz * h->FB))) return bail(rc);
    if (cfg->keep_potential && (rc = dalloc(h, &h->V, npix * cfg->nz))) return bail(rc);
    if (cfg->n_frames > 0) {
        if ((h->bx > 1 || h->by > 1) && (rc = dalloc(h, &h->bin_stage, (size_t)h->wx * h->wy * cfg->n_probes * h->FB))) return bail(rc);
        if ((rc = dalloc(h, &h->wf, h->wpitch * cfg->n_probes * cfg->n_frames))) return bail(rc);
        if (hipMemsetAsync(h->wf, 0, h->wpitch * cfg->n_probes * cfg->n_frames * sizeof(float2), h->stream) != hipSuccess)
            return bail(fail(h, MSL_ERR_HIP, "memset failed"));
    }
    if ((rc = dalloc(h, &h->pxt, (size_t)cfg->nx))) return bail(rc);
    if ((rc = dalloc(h, &h->pyt, (size_t)cfg->ny))) return bail(rc);
    if ((rc = dalloc(h, &h->d_lo, (size_t)cfg->nz))) return bail(rc);
    if ((rc = dalloc(h, &h->d_hi, (size_t)cfg->nz))) return bail(rc);
    if ((rc = dalloc(h, &h->d_abcd, (size_t)103 * 12))) return bail(rc);
    if ((rc = dalloc(h, &h->d_z2s, (size_t)104))) return bail(rc);
    if ((rc = dalloc(h, &h->d_species, (size_t)104))) return bail(rc);
    if ((rc = dalloc(h, &h->d_xy, (size_t)2 * cfg->n_probes))) return bail(rc);
    if ((rc = fill_propagator(h))) return bail(rc);
    *out = h;
    return MSL_OK;
}

int msl_destroy(msl_handle* h) {
    if (!h) return MSL_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& s : h->ring) for (auto e : s.ev) (void)hipEventDestroy(e);
    for (auto& st : h->stage) { if (st.ev) (void)hipEventDestroy(st.ev); if (st.buf) (void)hipHostFree(st.buf); }
    void* bufs[] = {h->psi0, h->psi, h->trans, h->V, h->wf, h->intensity, h->pxt, h->pyt, h->d_abcd, h->d_lo, h->d_hi,
                    h->d_pos, h->d_Z, h->d_key, h->d_order, h->d_u1, h->d_u2, h->d_ex, h->d_ey, h->d_counts, h->d_start,
                    h->d_z2s, h->d_species, h->d_ff, h->d_xy, h->plan_x.tw, h->plan_y.tw, h->plan_t.tw, h->tw4_x, h->tw4_y,
                    h->scratch, h->psiT, h->psi0T, h->transT, h->bin_stage, h->st_acc, h->st_s1, h->st_s2, h->st_tw, h->st_bins, h->st_ref, h->opx.tw2, h->opx.ptab, h->opy.tw2, h->opy.ptab,
                    ((h->opx.two || h->opx.breg || h->opx.breg2 || h->opx.breg4 || h->opx.wave2k) ? h->opx.tw : nullptr), ((h->opy.two || h->opy.breg || h->opy.breg2 || h->opy.breg4 || h->opy.wave2k) ? h->opy.tw : nullptr),
                    h->opx.bf, h->opx.bw, h->opy.bf, h->opy.bw, h->opx.qf, h->opy.qf, h->opx.mtw, h->opy.mtw, h->opx.cz_tw, h->opx.cz_tw2, h->opx.cz_bf, h->opx.cz_bw,
                    h->opy.cz_tw, h->opy.cz_tw2, h->opy.cz_bf, h->opy.cz_bw, h->opt.cz_tw, h->opt.cz_tw2, h->opt.cz_bf, h->opt.cz_bw, h->tsplit_tw, h->plan_x.chirp, h->plan_x.bfilt, h->plan_y.chirp, h->plan_y.bfilt, h->plan_t.chirp, h->plan_t.bfilt};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MSL_OK;
}

int msl_set_kirkland(msl_handle* h, const double* abcd) {
    if (!h || !abcd) return fail(h, MSL_ERR_INVALID, "msl_set_kirkland: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_abcd, abcd, 103 * 12 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_kirkland = true;
    h->n_species = 0;
    h->ff_n = 0;
    return MSL_OK;
}

int msl_set_slices(msl_handle* h, const double* lo, const double* hi) {
    if (!h || !lo || !hi) return fail(h, MSL_ERR_INVALID, "msl_set_slices: null argument");
    for (int s = 0; s < h->cfg.nz; ++s)
        if (!(hi[s] > lo[s]) || (s > 0 && lo[s] < lo[s - 1]))
            return fail(h, MSL_ERR_INVALID, "msl_set_slices: edges must be increasing with hi>lo (slice %d)", s);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_lo, lo, h->cfg.nz * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_hi, hi, h->cfg.nz * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_slices = true;
    return MSL_OK;
}

int msl_set_beam(msl_handle* h, double wavelength, double sigma, double dz) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (!(wavelength > 0)) return fail(h, MSL_ERR_INVALID, "msl_set_beam: wavelength must be positive");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    h->cfg.wavelength = wavelength; h->cfg.sigma = sigma; h->cfg.dz = dz;
    int rc = fill_propagator(h);
    if (rc) return rc;
    if (h->have_potential) {
        if (!h->V) return fail(h, MSL_ERR_STATE, "msl_set_beam: potential present but V not kept (keep_potential=0); rebuild the potential");
        const size_t n = (size_t)h->cfg.nx * h->cfg.ny * h->cfg.nz;
        hipLaunchKernelGGL(transmission_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->trans, h->V,
                           (long long)n, (float)sigma);
        HIPCHK(h, hipGetLastError());
        if ((rc = transpose_odd_slices(h))) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return MSL_OK;
}

int msl_resize_probes(msl_handle* h, int32_t n_probes) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (n_probes < 1) return fail(h, MSL_ERR_INVALID, "msl_resize_probes: need at least one probe");
    if (h->wf && n_probes != h->cfg.n_probes) return fail(h, MSL_ERR_STATE, "msl_resize_probes: handle owns a (P,T,nx,ny) result buffer");
    if (n_probes == h->cfg.n_probes) return MSL_OK;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int rc;
    const size_t images = (size_t)n_probes * h->FB;
    if ((rc = dalloc(h, &h->psi0, (size_t)h->cfg.nx * h->pitch * images))) return rc;
    if ((rc = dalloc(h, &h->psi, (size_t)h->cfg.nx * h->pitch * images))) return rc;
    if (h->onepass) {
        if ((rc = dalloc(h, &h->psiT, (size_t)h->cfg.ny * h->pitchT * images))) return rc;
        if (h->need_psi0T && (rc = dalloc(h, &h->psi0T, (size_t)h->cfg.ny * h->pitchT * images))) return rc;
    }
    if ((rc = dalloc(h, &h->d_xy, (size_t)2 * n_probes))) return rc;
    h->cfg.n_probes = n_probes;
    h->have_probes = false; h->have_exit = false;
    return MSL_OK;
}

int msl_shift_probes(msl_handle* h, const float* base, const double* xy, int32_t n_probes) {
    if (!h || !base || !xy) return fail(h, MSL_ERR_INVALID, "msl_shift_probes: null argument");
    if (n_probes != h->cfg.n_probes) return fail(h, MSL_ERR_INVALID, "msl_shift_probes: %d probes, handle has %d", n_probes, h->cfg.n_probes);
    const msl_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    const size_t npix = (size_t)c.nx * c.ny;
    float2* bk = nullptr;
    int rc = dalloc(h, &bk, npix);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(bk, base, npix * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_xy, xy, 2 * sizeof(double) * n_probes, hipMemcpyHostToDevice, h->stream));
    h->cur = nullptr;
    rc = fft2_inplace(h, bk, 1, +1, 1.0f, c.ny);
    if (rc == MSL_OK) {
        const long long total = (long long)npix * n_probes;
        hipLaunchKernelGGL(probe_ramp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->psi0, bk, h->d_xy,
                           n_probes, c.nx, c.ny, h->pitch, 1.0 / (c.nx * c.dx), 1.0 / (c.ny * c.dy));
        if (hipGetLastError() != hipSuccess) rc = fail(h, MSL_ERR_HIP, "probe_ramp_kernel launch failed");
    }
    if (rc == MSL_OK) rc = fft2_inplace(h, h->psi0, n_probes, -1, 1.0f / ((float)c.nx * (float)c.ny), h->pitch);
    if (rc == MSL_OK) rc = transpose_probes(h);
    hipError_t e = hipStreamSynchronize(h->stream);
    (void)hipFree(bk);
    if (rc) return rc;
    if (e != hipSuccess) return fail(h, MSL_ERR_HIP, "msl_shift_probes: %s", hipGetErrorString(e));
    h->have_probes = true;
    return MSL_OK;
}

int msl_set_probes(msl_handle* h, double mrad, const double* xy, int32_t n_probes) {
    if (!h || !xy) return fail(h, MSL_ERR_INVALID, "msl_set_probes: null argument");
    if (n_probes != h->cfg.n_probes) return fail(h, MSL_ERR_INVALID, "msl_set_probes: %d probes, handle has %d", n_probes, h->cfg.n_probes);
    if (mrad < 0) return fail(h, MSL_ERR_INVALID, "msl_set_probes: negative aperture");
    const msl_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    HIPCHK(h, hipMemcpyAsync(h->d_xy, xy, 2 * sizeof(double) * n_probes, hipMemcpyHostToDevice, h->stream));
    const long long total = (long long)c.nx * c.ny * n_probes;
    const double lx = c.nx * c.dx, ly = c.ny * c.dy;
    hipLaunchKernelGGL(probe_kspace_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->psi0, h->d_xy,
                       n_probes, c.nx, c.ny, h->pitch, 1.0 / lx, 1.0 / ly, 1.0 / (c.nx * c.dx), 1.0 / (c.ny * c.dy),
                       (mrad * 1e-3) / c.wavelength, mrad == 0 ? 1 : 0);
    HIPCHK(h, hipGetLastError());
    int rc = fft2_inplace(h, h->psi0, n_probes, -1, 1.0f / ((float)c.nx * (float)c.ny), h->pitch);
    if (rc) return rc;
    if ((rc = transpose_probes(h))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_probes = true;
    return MSL_OK;
}

int msl_upload_probes(msl_handle* h, const float* c64, int32_t n_probes) {
    if (!h || !c64) return fail(h, MSL_ERR_INVALID, "msl_upload_probes: null argument");
    if (n_probes != h->cfg.n_probes) return fail(h, MSL_ERR_INVALID, "msl_upload_probes: %d probes, handle has %d", n_probes, h->cfg.n_probes);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpy2DAsync(h->psi0, (size_t)h->pitch * sizeof(float2), c64, (size_t)h->cfg.ny * sizeof(float2),
                               (size_t)h->cfg.ny * sizeof(float2), (size_t)n_probes * h->cfg.nx, hipMemcpyHostToDevice, h->stream));
    { int rc = transpose_probes(h); if (rc) return rc; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_probes = true;
    return MSL_OK;
}

// Projected potentials + transmission functions of `count` MD frames (the same atoms, `count` sets of positions) into the batch
// slots first_slot .. first_slot + count - 1: ONE launch each of the atom preparation, the stable counting sort (keys = frame x
// slice x species), the two phase tables, the structure factor and the two inverse-transform passes for as many frames as the
// phase tables of a group may take (6 GB), instead of that sequence per frame.  The reference builds one Potential per frame
// (calculators.py:172-186, potentials.py:188-348); with its default single probe that build IS the frame (round 2: 0.43 of
// 0.62 ms at 512^2 x 100 slices, of which ~110 us were launches of 5-15 us kernels and four small copies per frame).
static int build_potentials(msl_handle* h, const double* pos, const int32_t* Z, int64_t n, int count, int first_slot, int32_t ax1,
                            int32_t ax2, int32_t axs) {
    const msl_config& c = h->cfg;
    const size_t npix = (size_t)c.nx * c.ny;
    // species present (sorted ascending, like np.unique)
    int z2s[104];
    for (int i = 0; i < 104; ++i) z2s[i] = -1;
    for (int64_t a = 0; a < n; ++a) {
        if (Z[a] < 1 || Z[a] > 103) return fail(h, MSL_ERR_INVALID, "msl_build_potential: atomic number %d out of 1..103", Z[a]);
        z2s[Z[a]] = 0;
    }
    int species[104], nsp = 0;
    for (int z = 1; z <= 103; ++z) if (z2s[z] == 0) { z2s[z] = nsp; species[nsp++] = z; }
    int rc;
    // with launch timing off the call only queues work: no event, no host wait (the frames of a run pipeline on the stream)
    const bool timed = h->cfg.launch_timing != 0;
    EventPair evp_; hipEvent_t &e0 = evp_.a, &e1 = evp_.b;
    if (timed) {
        HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
        HIPCHK(h, hipEventRecord(e0, h->stream));
    }
    if (nsp > h->ff_species_cap) {
        if ((rc = dalloc(h, &h->d_ff, npix * nsp))) return rc;
        h->ff_species_cap = nsp;
        h->ff_n = 0;
    }
    // frames per group: the phase tables of a group (n atoms x (nx/2 + 1 + ny/2 + 1) x 8 bytes per frame) stay under 6 GB
    const int cx = c.nx / 2 + 1, cy = c.ny / 2 + 1;             // table columns the quadrant kernel reads
    const size_t table_bytes_per_frame = std::max<size_t>(1, (size_t)n * (size_t)(cx + cy) * sizeof(float2));
    const int G = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, (size_t)6e9 / table_bytes_per_frame));
    const int keys_per_frame = c.nz * std::max(nsp, 1);
    const int nkeys_cap = keys_per_frame * G;
    if (nkeys_cap > h->keys_cap) {
        if ((rc = dalloc(h, &h->d_counts, (size_t)nkeys_cap + 1))) return rc;
        if ((rc = dalloc(h, &h->d_start, (size_t)nkeys_cap + 1))) return rc;
        h->keys_cap = nkeys_cap;
    }
    if ((rc = ensure_atoms(h, (size_t)n * G, (size_t)n * G + (size_t)(SF_ALIGN - 1) * nkeys_cap))) return rc;
    // R_s is Hermitian (real V): only the rows kx <= nx/2 are written and row-transformed when the inverse transform mirrors them
    // itself (every register-kernel path: col_pass_kernel<.., HERM>, ifftT2 / ifftTB / ifftTW with job.herm)
    const bool tw_axes = h->onepass && h->opx.wave2k && h->opy.wave2k && !h->V && h->transT;
    const bool t2_axes = h->onepass && h->opx.two && h->opy.two && h->opx.R == 16 && h->opy.R == 16 && !h->V;
    const bool tb_axes = h->onepass && !h->V && h->transT &&
                         (h->opx.breg || h->opx.breg2 || h->opx.cz_R) && (h->opy.breg || h->opy.breg2 || h->opy.cz_R);
    const bool herm_ifft = h->Rx && h->Ry;                      // four-step kernels on both axes (256 / 1024)
    const bool herm_tb = tb_axes, herm_t2 = t2_axes, herm_tw = tw_axes;
    const bool half_rows = herm_ifft || herm_tb || herm_t2 || herm_tw;
    const float vscale = (float)(1.0 / ((double)c.nx * c.ny) / (c.dx * c.dx * c.dy * c.dy));
    struct CzRef { int R; const float2 *tw, *tw2, *bf, *bw; };      // R = 64: the 2048-point wave FFT
    auto cz_of = [](const msl_handle::OpDir& o) -> CzRef {
        if (o.breg2) return {64, o.tw, o.tw2, o.bf, o.bw};
        if (o.breg) return {o.R, o.tw, nullptr, o.bf, o.bw};
        return {o.cz_R, o.cz_tw, o.cz_tw2, o.cz_bf, o.cz_bw};
    };
    const bool ifft_t2 = t2_axes;
    const bool ifft_tb = h->onepass && cz_of(h->opx).R && cz_of(h->opy).R && !h->V && h->transT;
    const bool ifft_tw = tw_axes;
    bool maps_sent = false;
    for (int f0 = 0; f0 < count; f0 += G) {
        const int g = std::min(G, count - f0);
        const int slot = first_slot + f0;
        float2* const TR = h->trans + (size_t)slot * c.nz * npix;                 // batch slots this group's stacks go to
        float2* const TRT = h->transT ? h->transT + (size_t)slot * c.nz * npix : nullptr;
        const int n_slices = c.nz * g, nkeys = keys_per_frame * g;
        const long long rows = (long long)n * g;
        bool recip_written = false;
        if (n > 0 && nsp > 0) {
            {
                msl_handle::HostStage& st = h->stage[h->stage_pos++ & 1];
                if (!st.ev) HIPCHK(h, hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
                if (st.used) HIPCHK(h, hipEventSynchronize(st.ev));            // the copies queued from this slot two calls ago
                const size_t off_sp = sizeof z2s, off_Z = (off_sp + sizeof species + 7) & ~(size_t)7;
                const size_t off_pos = (off_Z + (size_t)n * sizeof(int) + 7) & ~(size_t)7, need = off_pos + (size_t)rows * 3 * sizeof(double);
                if (need > st.bytes) {
                    if (st.buf) (void)hipHostFree(st.buf);
                    st.buf = nullptr; st.bytes = 0;
                    const size_t cap = need + need / 4;
                    if (hipHostMalloc((void**)&st.buf, cap, hipHostMallocDefault) != hipSuccess)
                        return fail(h, MSL_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed", cap);
                    st.bytes = cap;
                }
                memcpy(st.buf + off_pos, pos + (size_t)f0 * n * 3, (size_t)rows * 3 * sizeof(double));
                if (!maps_sent) {
                    memcpy(st.buf, z2s, sizeof z2s);
                    memcpy(st.buf + off_sp, species, sizeof species);
                    memcpy(st.buf + off_Z, Z, (size_t)n * sizeof(int));
                    HIPCHK(h, hipMemcpyAsync(h->d_z2s, st.buf, sizeof z2s, hipMemcpyHostToDevice, h->stream));
                    HIPCHK(h, hipMemcpyAsync(h->d_species, st.buf + off_sp, nsp * sizeof(int), hipMemcpyHostToDevice, h->stream));
                    HIPCHK(h, hipMemcpyAsync(h->d_Z, st.buf + off_Z, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
                    maps_sent = true;
                }
                HIPCHK(h, hipMemcpyAsync(h->d_pos, st.buf + off_pos, (size_t)rows * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
                HIPCHK(h, hipEventRecord(st.ev, h->stream));
                st.used = true;
            }
            HIPCHK(h, hipMemsetAsync(h->d_counts, 0, ((size_t)nkeys + 1) * sizeof(int), h->stream));
            const double lx = c.nx * c.dx, ly = c.ny * c.dy;
            // f_Z(q^2) depends on the grid and the species only (potentials.py:283-293 recomputes it per frame): build the table
            // when the species list changes, i.e. once per run
            if (nsp != h->ff_n || memcmp(species, h->ff_species, nsp * sizeof(int)) != 0) {
                long long tot = (long long)npix * nsp;
                hipLaunchKernelGGL(formfactor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->d_ff, h->d_abcd,
                                   h->d_species, nsp, c.nx, c.ny, 1.0 / lx, 1.0 / ly);
                memcpy(h->ff_species, species, nsp * sizeof(int));
                h->ff_n = nsp;
            }
            hipLaunchKernelGGL(atom_prep_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->stream, h->d_pos, h->d_Z,
                               (long long)n, g, h->d_z2s, h->d_lo, h->d_hi, c.nz, nsp, ax1, ax2, axs, 1.0 / lx, 1.0 / ly, h->d_key,
                               h->d_u1, h->d_u2, h->d_counts);
            // many atoms per (slice, species): the streaming structure-factor kernel, whose bins are padded to whole half-trips
            const bool sf_stream = (double)n / std::max(1, keys_per_frame) >= 128.0 && !dbg_env("MSL_SF_TILED");
            hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_counts, h->d_start, nkeys, sf_stream ? SF_ALIGN : 1);
            hipLaunchKernelGGL(bin_fill_kernel, dim3(nkeys), dim3(1024), 0, h->stream, h->d_key, (long long)n, keys_per_frame, h->d_start, h->d_order);
            HIPCHK(h, hipGetLastError());
            // atoms that fell into a slice: d_start[nkeys], read by the kernels themselves (grids sized for all atoms of the group)
            const int* n_sorted = h->d_start + nkeys;
            recip_written = true;
            const long long rows_pad = rows + (long long)(SF_ALIGN - 1) * nkeys;          // sorted rows at most: every bin padded to SF_ALIGN
            if (rows_pad > 0x7fffffffLL) return fail(h, MSL_ERR_INVALID, "msl_build_potentials: %lld padded table rows in one group exceed 2^31", rows_pad);
            const long long tx = rows_pad * cx, ty = rows_pad * cy;
            hipLaunchKernelGGL(phase_table_kernel, dim3((unsigned)((tx + 255) / 256)), dim3(256), 0, h->stream, h->d_ex, h->d_u1,
                               h->d_order, n_sorted, c.nx, cx, cx);
            hipLaunchKernelGGL(phase_table_kernel, dim3((unsigned)((ty + 255) / 256)), dim3(256), 0, h->stream, h->d_ey, h->d_u2,
                               h->d_order, n_sorted, c.ny, cy, cy);
            // matrix-core kernel over the quadrant of non-negative frequencies 0 .. n/2 in 32 x 32 tiles; on power-of-two grids
            // (n/2 + 1 = 32 k + 1) the Nyquist row / column goes to the edge kernel instead of a tile row of its own
            const bool edge_x = (c.nx % 2 == 0) && (cx % 32 == 1) && cx > 1, edge_y = (c.ny % 2 == 0) && (cy % 32 == 1) && cy > 1;
            const int tiles_x = edge_x ? cx / 32 : (cx + 31) / 32, tiles_y = edge_y ? cy / 32 : (cy + 31) / 32;
            const int n_tiles = tiles_x * tiles_y, wg_per_slice = (n_tiles + 3) / 4;
            const long long n_wg = (long long)wg_per_slice * ((n_slices + 7) / 8 * 8);
            if (n_wg > 0x7fffffffLL) return fail(h, MSL_ERR_UNSUPPORTED, "structure factor: too many workgroups");
            if (sf_stream) {
                // persistent form, one wave per SIMD: one workgroup per CU, a multiple of 8 (XCD-local slices)
                const int n_pers = std::max(8, h->n_cus / 8 * 8);
                if (dbg_env("MSL_SF_F32"))                   // the exact-f32 matrix instruction (A/B against the split-bf16 form)
                    hipLaunchKernelGGL(structure_factor_stream_kernel, dim3((unsigned)n_pers), dim3(256), 0, h->stream, TR, h->d_ex, h->d_ey, h->d_ff,
                                       h->d_start, nsp, c.nx, c.ny, tiles_y, n_tiles, (int)rows_pad, half_rows ? 0 : 1, cx, cy, n_slices);
                // two kx tiles per wave sharing the ey planes (potential.h) where a slice's tables outgrow an XCD's L2: 2048^2 x 50 potential
                // 8.67 -> 8.17 ms per frame, 1024^2 (C3) 3.72 -> 3.69 (kept on the one-tile kernel)
                else if ((n_tiles >= 512 || (tiles_x >= 2 && dbg_env("MSL_SF_TWO_TILES"))) && !dbg_env("MSL_SF_ONE_TILE"))
                    hipLaunchKernelGGL(structure_factor_stream_bf16x2_kernel, dim3((unsigned)n_pers), dim3(256), 4 * 128 * 64 * sizeof(float), h->stream, TR, h->d_ex, h->d_ey, h->d_ff,
                                       h->d_start, nsp, c.nx, c.ny, tiles_y, ((tiles_x + 1) / 2) * tiles_y, (int)rows_pad, half_rows ? 0 : 1, cx, cy, n_slices);
                else
                    hipLaunchKernelGGL(structure_factor_stream_bf16_kernel, dim3((unsigned)n_pers), dim3(256), 0, h->stream, TR, h->d_ex, h->d_ey, h->d_ff,
                                       h->d_start, nsp, c.nx, c.ny, tiles_y, n_tiles, (int)rows_pad, half_rows ? 0 : 1, cx, cy, n_slices);
            } else
                hipLaunchKernelGGL(structure_factor_quad_kernel, dim3((unsigned)n_wg), dim3(256), 0, h->stream, TR, h->d_ex, h->d_ey, h->d_ff,
                                   h->d_start, nsp, c.nx, c.ny, tiles_y, n_tiles, (int)rows_pad, half_rows ? 0 : 1, cx, cy, n_slices, wg_per_slice);
            if (edge_x || edge_y) {
                const int bins = (edge_x ? cy : 0) + (edge_y ? (edge_x ? cx - 1 : cx) : 0);
                for (int s0 = 0; s0 < n_slices; s0 += 65535) {
                    const int ns = std::min(65535, n_slices - s0);
                    hipLaunchKernelGGL(structure_factor_edge_kernel, dim3((bins + 127) / 128, ns), dim3(128), 0, h->stream,
                                       TR + (size_t)s0 * npix, h->d_ex, h->d_ey, h->d_ff, h->d_start + (size_t)s0 * nsp, nsp, c.nx, c.ny,
                                       edge_x ? 1 : 0, edge_y ? 1 : 0, half_rows ? 0 : 1, cx, cy);
                }
            }
            HIPCHK(h, hipGetLastError());
        }
        if (!recip_written) HIPCHK(h, hipMemsetAsync(TR, 0, npix * n_slices * sizeof(float2), h->stream));
        // V_s = Re ifft2(R_s) / (dx^2 dy^2);  t_s = exp(i sigma V_s)  -- in place over the (frames, nz, nx, ny) stacks of the group.
        // Slices a pass along x reads are kept transposed (scheme b: odd slices; alternating scheme: odd distance to the last one):
        // with several frames per launch the slice number is the image index modulo nz (slice_mod).
        h->cur = nullptr;
        if (ifft_tw) {
            auto passw = [&](const msl_handle::OpDir& o, IfftTBJob j) -> int {
                constexpr int N2 = 2048;
                const size_t lds = ((size_t)N2 + 64 + (size_t)8 * (N2 + 1)) * 8;
                const long long items = (long long)(j.n_lines / 8) * j.n_images;
                const int grid = (int)std::min<long long>(items, (long long)h->n_cus);
                j.tw = o.tw; j.tw2 = o.tw2;
                (void)hipFuncSetAttribute((const void*)ifftTW_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
                hipLaunchKernelGGL(ifftTW_kernel, dim3(grid), dim3(512), lds, h->stream, j);
                HIPCHK(h, hipGetLastError());
                return mark_launch(h, K_OTHER);
            };
            IfftTBJob a{};
            a.in = TR; a.out_t = TRT; a.out_rows = nullptr;
            a.in_is = a.out_t_is = (long long)npix; a.in_pitch = c.ny; a.out_t_pitch = c.nx; a.n_lines = c.nx; a.n_line = c.ny; a.n_images = n_slices;
            a.potential = 0; a.rows_parity = -1; a.slice_mod = c.nz;
            if (herm_tw) a.n_lines = (c.nx / 2 + 1 + 7) / 8 * 8;
            if ((rc = passw(h->opy, a))) return rc;
            IfftTBJob b{};
            b.in = TRT; b.out_t = TR; b.out_rows = TRT;
            b.in_is = b.out_t_is = b.out_rows_is = (long long)npix; b.in_pitch = c.nx; b.out_t_pitch = c.ny; b.out_rows_pitch = c.nx;
            b.n_lines = c.ny; b.n_line = c.nx; b.n_images = n_slices; b.potential = 1; b.herm = herm_tw ? 1 : 0; b.slice_mod = c.nz;
            b.rows_parity = slice_is_transposed(h, 1) ? 1 : 0;
            b.scale = vscale; b.sigma_over_pi = (float)(c.sigma / M_PI);
            if ((rc = passw(h->opx, b))) return rc;
        } else if (ifft_tb) {
            auto pass = [&](const CzRef& o, IfftTBJob j) -> int {
                if (o.R == 64) {                                // 513 .. 1024 points: the wave-per-line 2048-point FFT
                    constexpr int M2 = 2048, NH2 = 1024, RS = (32 * W2K_PITCH) / 2 + 1;
                    const size_t lds2 = ((size_t)M2 + 64 + NH2 + 2 + NH2 + (size_t)8 * RS) * 8;
                    const long long items2 = (long long)((j.n_lines + 7) / 8) * j.n_images;
                    const int grid2 = (int)std::min<long long>(items2, (long long)h->n_cus);
                    j.tw = o.tw; j.tw2 = o.tw2; j.bf = o.bf; j.bw = o.bw;
                    (void)hipFuncSetAttribute((const void*)ifftTB2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
                    hipLaunchKernelGGL(ifftTB2_kernel, dim3(grid2), dim3(512), lds2, h->stream, j);
                    HIPCHK(h, hipGetLastError());
                    return mark_launch(h, K_OTHER);
                }
                const int R = o.R, M = R * R, NH = M / 2, CS = R * (R + 1) + 2;
                const size_t lds = ((size_t)M + NH + 2 + NH + (size_t)16 * CS) * 8;
                const int per_cu = std::max(1, std::min(R == 16 ? 4 : 1, (int)((size_t)h->lds_limit / lds)));
                // first pass (no epilogue): two lines per group and tile round
                if (R == 32 && !j.potential && !dbg_env("MSL_NO_TWO_LINE_IFFT")) {
                    constexpr int CSN = 514;
                    const size_t lds2 = ((size_t)M + NH + 2 + NH + (size_t)32 * CSN) * 8;
                    const long long items2 = (long long)((j.n_lines + 31) / 32) * j.n_images;
                    const int grid2 = (int)std::min<long long>(items2, (long long)h->n_cus);
                    j.tw = o.tw; j.bf = o.bf; j.bw = o.bw;
                    (void)hipFuncSetAttribute((const void*)ifftTB_two_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
                    hipLaunchKernelGGL(ifftTB_two_kernel, dim3(grid2), dim3(512), lds2, h->stream, j);
                    HIPCHK(h, hipGetLastError());
                    return mark_launch(h, K_OTHER);
                }
                // second pass on a half spectrum: two real lines per transform (32-line work items)
                const bool pair = R == 32 && j.herm && j.potential && !dbg_env("MSL_NO_PAIRED_IFFT");
                const long long items = (long long)((j.n_lines + (pair ? 31 : 15)) / (pair ? 32 : 16)) * j.n_images;
                const int grid = (int)std::min<long long>(items, (long long)h->n_cus * per_cu);
                j.tw = o.tw; j.bf = o.bf; j.bw = o.bw;
                if (pair) {
                    (void)hipFuncSetAttribute((const void*)ifftTB_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
                    hipLaunchKernelGGL((ifftTB_kernel<32, true>), dim3(grid), dim3(512), lds, h->stream, j);
                } else if (R == 32) {
                    (void)hipFuncSetAttribute((const void*)ifftTB_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
                    hipLaunchKernelGGL(ifftTB_kernel<32>, dim3(grid), dim3(512), lds, h->stream, j);
                } else {
                    (void)hipFuncSetAttribute((const void*)ifftTB_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
                    hipLaunchKernelGGL(ifftTB_kernel<16>, dim3(grid), dim3(256), lds, h->stream, j);
                }
                HIPCHK(h, hipGetLastError());
                return mark_launch(h, K_OTHER);
            };
            IfftTBJob a{};
            a.in = TR; a.out_t = TRT; a.out_rows = nullptr;
            a.in_is = a.out_t_is = (long long)npix; a.in_pitch = c.ny; a.out_t_pitch = c.nx; a.n_lines = c.nx; a.n_line = c.ny; a.n_images = n_slices;
            a.potential = 0; a.rows_parity = -1; a.slice_mod = c.nz;
            if (herm_tb) a.n_lines = c.nx / 2 + 1;                  // the other rows are their mirror images (taken by the second pass's loads)
            if ((rc = pass(cz_of(h->opy), a))) return rc;
            IfftTBJob b{};
            b.in = TRT; b.out_t = TR; b.out_rows = TRT;
            b.in_is = b.out_t_is = b.out_rows_is = (long long)npix; b.in_pitch = c.nx; b.out_t_pitch = c.ny; b.out_rows_pitch = c.nx;
            b.n_lines = c.ny; b.n_line = c.nx; b.n_images = n_slices; b.potential = 1; b.herm = herm_tb ? 1 : 0; b.slice_mod = c.nz;
            b.rows_parity = slice_is_transposed(h, 1) ? 1 : 0;       // the slices a pass along x reads stay in TRT as rows
            b.scale = vscale; b.sigma_over_pi = (float)(c.sigma / M_PI);
            if ((rc = pass(cz_of(h->opx), b))) return rc;
        } else if (ifft_t2) {
            // 512 x 512 grids: two transposing inverse-FFT passes on the register kernels (TR -> TRT along y, TRT -> TR / TRT along
            // x with the potential epilogue; slices a pass along x reads stay in TRT as rows)
            auto pass = [&](const IfftT2Job& j) -> int {
                constexpr int R = 16, N2 = R * R, N = 2 * N2;
                const size_t lds = ((size_t)2 * N2 + (size_t)16 * (N + 1)) * 8;
                const int per_cu = std::max(1, std::min(2, (int)((size_t)h->lds_limit / lds)));
                // second pass on a half spectrum: two real lines per transform (32-line work items)
                const bool pair = j.herm && j.potential && j.n_lines % 32 == 0 && !dbg_env("MSL_NO_PAIRED_IFFT");
                const long long items = (long long)(j.n_lines / (pair ? 32 : 16)) * j.n_images;
                const int grid = (int)std::min<long long>(items, (long long)h->n_cus * per_cu);
                if (pair) {
                    (void)hipFuncSetAttribute((const void*)ifftT2_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
                    hipLaunchKernelGGL((ifftT2_kernel<16, true>), dim3(grid), dim3(256), lds, h->stream, j);
                    HIPCHK(h, hipGetLastError());
                    return mark_launch(h, K_OTHER);
                }
                (void)hipFuncSetAttribute((const void*)ifftT2_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
                hipLaunchKernelGGL(ifftT2_kernel<16>, dim3(grid), dim3(256), lds, h->stream, j);
                HIPCHK(h, hipGetLastError());
                return mark_launch(h, K_OTHER);
            };
            IfftT2Job a{};
            a.in = TR; a.out_t = TRT; a.out_rows = nullptr; a.tw = h->opy.tw; a.tw2 = h->opy.tw2;
            a.in_is = a.out_t_is = (long long)npix; a.in_pitch = c.ny; a.out_t_pitch = c.nx; a.n_lines = c.nx; a.n_images = n_slices;
            a.potential = 0; a.rows_parity = -1; a.slice_mod = c.nz;
            if (herm_t2) a.n_lines = (c.nx / 2 + 1 + 15) / 16 * 16;        // rows kx <= nx/2 in whole 16-line blocks (the surplus rows are never read)
            if ((rc = pass(a))) return rc;
            IfftT2Job b{};
            b.in = TRT; b.out_t = TR; b.out_rows = TRT; b.tw = h->opx.tw; b.tw2 = h->opx.tw2;
            b.in_is = b.out_t_is = b.out_rows_is = (long long)npix; b.in_pitch = c.nx; b.out_t_pitch = c.ny; b.out_rows_pitch = c.nx;
            b.n_lines = c.ny; b.n_images = n_slices; b.potential = 1; b.rows_parity = 1; b.slice_mod = c.nz;      // scheme b: slice s is read along x iff s is odd
            b.herm = herm_t2 ? 1 : 0;
            b.scale = vscale; b.sigma_over_pi = (float)(c.sigma / M_PI);
            if ((rc = pass(b))) return rc;
        } else if (h->Ry) {
            RowJob r = row_job(h, TR, n_slices, c.ny);
            r.do_ifft = 1;
            if (herm_ifft) { const int Gr = 256 / h->Ry; r.nx = (c.nx / 2 + 1 + Gr - 1) / Gr * Gr; }     // whole row groups (the surplus rows are never read)
            if ((rc = launch_row_fast(h, r, K_OTHER))) return rc;
        } else {
            LineArgs r = row_args(h, TR, TR, n_slices, c.ny);
            r.fft1 = -1;
            if ((rc = launch_lines(h, h->plan_y, r, K_OTHER))) return rc;
        }
        if (ifft_t2 || ifft_tb || ifft_tw) {
            // (both passes done above)
        } else if (h->Rx) {
            ColJob k = col_job(h, TR, TR, n_slices, c.ny, c.ny);
            k.flags = COL_INV | COL_POTENTIAL; k.scale = vscale; k.sigma = (float)c.sigma; k.out_real = h->V; k.slice_mod = c.nz;
            // (a kept V is written by the untransposed store only: with keep_potential the x-pass slices are transposed afterwards)
            if (h->onepass && !h->V) { k.flags |= COL_TPOT; k.tparity = h->scheme_b ? 0 : ((c.nz - 1) & 1); k.out_t = TRT; }
            if ((rc = herm_ifft ? launch_col_herm(h, k, K_OTHER) : launch_col_fast(h, k, K_OTHER))) return rc;
            if (h->onepass && h->V && (rc = transpose_odd_slices(h))) return rc;
        } else {
            LineArgs k = col_args(h, TR, TR, n_slices, c.ny, c.ny);
            k.fft1 = -1;
            k.scale = vscale;
            k.store_mode = STORE_POTENTIAL; k.out_real = h->V; k.sigma = (float)c.sigma;
            if ((rc = launch_lines(h, h->plan_x, k, K_OTHER))) return rc;
            const int saved = h->cur_batch;                         // one-pass loop on such a grid: x-pass slices transposed, stack by stack
            for (int f = 0; f < g && rc == MSL_OK; ++f) { h->cur_batch = slot + f; rc = transpose_odd_slices(h); }
            h->cur_batch = saved;
            if (rc) return rc;
        }
    }
    h->n_species = nsp;
    if (timed) {
        HIPCHK(h, hipEventRecord(e1, h->stream));
        HIPCHK(h, hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
        h->ctr.ms_potential += ms;
    }
    h->have_potential = true;
    return MSL_OK;
}

static int check_potential_args(msl_handle* h, const char* who, const double* pos, const int32_t* Z, int64_t n, int32_t ax1, int32_t ax2, int32_t axs) {
    if (!h || (n > 0 && (!pos || !Z))) return fail(h, MSL_ERR_INVALID, "%s: null argument", who);
    if (!h->have_kirkland) return fail(h, MSL_ERR_STATE, "%s: call msl_set_kirkland first", who);
    if (!h->have_slices) return fail(h, MSL_ERR_STATE, "%s: call msl_set_slices first", who);
    if (n < 0 || n > 0x7fffffff) return fail(h, MSL_ERR_INVALID, "%s: bad atom count", who);
    if (ax1 < 0 || ax1 > 2 || ax2 < 0 || ax2 > 2 || axs < 0 || axs > 2 || ((1 << ax1) | (1 << ax2) | (1 << axs)) != 7)
        return fail(h, MSL_ERR_INVALID, "%s: axes must be a permutation of 0,1,2", who);
    return MSL_OK;
}

int msl_build_potential(msl_handle* h, const double* pos, const int32_t* Z, int64_t n, int32_t ax1, int32_t ax2, int32_t axs) {
    int rc = check_potential_args(h, "msl_build_potential", pos, Z, n, ax1, ax2, axs);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    return build_potentials(h, pos, Z, n, 1, h->cur_batch, ax1, ax2, axs);
}

int msl_build_potentials(msl_handle* h, const double* pos, const int32_t* Z, int64_t n, int32_t count, int32_t ax1, int32_t ax2, int32_t axs) {
    int rc = check_potential_args(h, "msl_build_potentials", pos, Z, n, ax1, ax2, axs);
    if (rc) return rc;
    if (count < 1 || count > h->FB) return fail(h, MSL_ERR_INVALID, "msl_build_potentials: count %d outside [1,%d] (msl_config.frame_batch)", count, h->FB);
    if ((int64_t)n * count > 0x7fffffffLL) return fail(h, MSL_ERR_INVALID, "msl_build_potentials: more than 2^31 atoms in one batch");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    rc = build_potentials(h, pos, Z, n, count, 0, ax1, ax2, axs);
    if (rc == MSL_OK) h->cur_batch = count - 1;         // the current stack (MSL_BUF_TRANSMISSION, msl_propagate) is the last frame's, as after `count` single builds
    return rc;
}

int msl_upload_potential(msl_handle* h, const float* V) {
    if (!h || !V) return fail(h, MSL_ERR_INVALID, "msl_upload_potential: null argument");
    const msl_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    const size_t n = (size_t)c.nx * c.ny * c.nz;
    float* dst = h->V;
    float* tmp = nullptr;
    if (!dst) { int rc = dalloc(h, &tmp, n); if (rc) return rc; dst = tmp; }
    HIPCHK(h, hipMemcpyAsync(dst, V, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(transmission_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->trans + (size_t)h->cur_batch * n, dst,
                       (long long)n, (float)c.sigma);
    HIPCHK(h, hipGetLastError());
    { int rc = transpose_odd_slices(h); if (rc) return rc; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (tmp) (void)hipFree(tmp);
    h->have_potential = true;
    return MSL_OK;
}

static int run_loop(msl_handle* h, int slot, int groups = 1) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    const int first_group = groups > 1 ? 0 : h->cur_batch;
    if (!h->have_probes) return fail(h, MSL_ERR_STATE, "propagate: no probes (msl_set_probes / msl_upload_probes)");
    if (!h->have_potential) return fail(h, MSL_ERR_STATE, "propagate: no potential (msl_build_potential / msl_upload_potential)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->cfg.launch_timing) return slice_loop(h, slot, groups, first_group);         // queued; msl_synchronize / msl_download wait for it
    EventPair evp_; hipEvent_t &e0 = evp_.a, &e1 = evp_.b;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, hipEventRecord(e0, h->stream));
    int rc = slice_loop(h, slot, groups, first_group);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    h->ctr.ms_propagate += ms;
    return MSL_OK;
}

int msl_propagate(msl_handle* h) {
    int rc = run_loop(h, -1);
    if (rc == MSL_OK) h->have_exit = true;
    return rc;
}

int msl_propagate_frame(msl_handle* h, int32_t slot) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (!h->wf) return fail(h, MSL_ERR_STATE, "msl_propagate_frame: handle created with n_frames == 0");
    if (slot < 0 || slot >= h->cfg.n_frames) return fail(h, MSL_ERR_INVALID, "msl_propagate_frame: slot %d out of range [0,%d)", slot, h->cfg.n_frames);
    return run_loop(h, slot);
}

int msl_select_batch_slot(msl_handle* h, int32_t b) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (b < 0 || b >= h->FB) return fail(h, MSL_ERR_INVALID, "msl_select_batch_slot: slot %d out of range [0,%d)", b, h->FB);
    h->cur_batch = b;
    return MSL_OK;
}

int msl_frame_batch(const msl_handle* h) { return h ? h->FB : 0; }

int msl_propagate_frames(msl_handle* h, int32_t first_slot, int32_t count) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (!h->wf) return fail(h, MSL_ERR_STATE, "msl_propagate_frames: handle created with n_frames == 0");
    if (count < 1 || count > h->FB) return fail(h, MSL_ERR_INVALID, "msl_propagate_frames: count %d outside [1,%d] (msl_config.frame_batch)", count, h->FB);
    if (first_slot < 0 || first_slot + count > h->cfg.n_frames)
        return fail(h, MSL_ERR_INVALID, "msl_propagate_frames: slots [%d,%d) outside [0,%d)", first_slot, first_slot + count, h->cfg.n_frames);
    if (count == 1) { const int saved = h->cur_batch; h->cur_batch = 0; int rc = run_loop(h, first_slot, 1); h->cur_batch = saved; return rc; }
    return run_loop(h, first_slot, count);
}

int msl_tacaw(msl_handle* h, const void* d_src, void* d_dst, int64_t batch, int32_t T, int64_t npix) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    const msl_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    const float2* src = (const float2*)d_src;
    float* dst = (float*)d_dst;
    if (!src) {
        if (!h->wf) return fail(h, MSL_ERR_STATE, "msl_tacaw: no wavefunction buffer");
        // the pad pixels of an image are pixels like any other here (zeros in, zeros out)
        src = h->wf; batch = c.n_probes; T = c.n_frames; npix = (int64_t)h->wpitch;
        size_t need = (size_t)batch * T * npix;
        if (h->intensity_elems != need) {
            int rc = dalloc(h, &h->intensity, need);
            if (rc) return rc;
            h->intensity_elems = need;
        }
        h->intensity_F = T; h->intensity_ld = h->wpitch;
        dst = h->intensity;
    } else if (!dst) {
        return fail(h, MSL_ERR_INVALID, "msl_tacaw: src given without dst");
    }
    if (T < 2) return fail(h, MSL_ERR_INVALID, "msl_tacaw: needs at least 2 frames (got %d)", T);
    if (batch < 1 || npix < 1) return fail(h, MSL_ERR_INVALID, "msl_tacaw: bad batch/npix");
    if (npix > 0x7fffffffLL) return fail(h, MSL_ERR_UNSUPPORTED, "msl_tacaw: npix too large");
    // smooth counts up to 1024: the register network split over the waves of a workgroup (time_split_kernel)
    // 32-bit offsets: the buffer unit adds the lane offset and the scalar row offset in 32 bits (measured: the sum wraps) --
    // pixel + up to 64 rows with one block per wave, pixel + TP + (TP + 1) / 2 rows with two; time_split_launch falls back from the
    // two-block shape to the one-block shape of the same L where there is one (L = 2, 4: up to 512 frames)
    const bool split_t = c.fft_path == 0 && time_split_fits(T, npix)
                         && !dbg_env("MSL_TACAW_GENERIC") && !dbg_env("MSL_TACAW_CHIRPZ") && !(T == 1024 && dbg_env("MSL_TACAW_FOURSTEP"));
    // 1024 frames of images too large for that: the four-step column kernel (it served 256 and 1024 frames until the split kernel
    // overtook it: T = 256, 64 probes x 1024^2 40.2 -> 39.7 ms, 16 x 2048^2 44.7 -> 40.7 ms; T = 1024, 8 x 1024^2 33.7 -> 29.3 ms)
    const int Rt = (c.fft_path == 0 && T == 1024 && !split_t) ? fast_radix(T) : 0;
    const bool fast_t = Rt && (npix % 16 == 0) && npix >= 32 && !dbg_env("MSL_TACAW_GENERIC");
    // smooth frame counts from 16 to 128 (100 = 4.5.5 ...): a lane per pixel, the whole time line in its registers (time_direct_kernel)
    const bool direct_t = !fast_t && c.fft_path == 0 && time_direct_has(T) && (unsigned long long)((T + 1) / 2) * (unsigned long long)npix * 8ull < (1ull << 32)
                          && !dbg_env("MSL_TACAW_GENERIC") && !dbg_env("MSL_TACAW_CHIRPZ");
    const bool cz_t = !fast_t && !direct_t && !split_t && c.fft_path == 0 && T <= 512 && !dbg_env("MSL_TACAW_GENERIC");
    int rc = MSL_OK;
    float2* tw4_t = nullptr;
    if (fast_t) {
        if ((rc = make_tw4(h, &tw4_t, Rt))) return rc;
    } else if (direct_t || split_t) {
    } else if (cz_t) {
        if (h->opt_T != T) {
            h->opt_T = 0;
            if ((rc = make_cz_tables(h, h->opt, T < 33 ? 33 : T))) return rc;       // (the table builder's R rule starts at 33 points)
            if (T < 33) {                                                             // chirp of the real T: redo the two T-dependent tables
                const int M = 256, NH = 128;
                std::vector<float2> bw(NH, make_float2(0.f, 0.f)), bf(NH + 2, make_float2(0.f, 0.f));
                std::vector<double> cr(M, 0.0), ci(M, 0.0);
                for (int i = 0; i < T; ++i) {
                    const long long q = ((long long)i * i) % (2LL * T);
                    const double a = -M_PI * (double)q / (double)T;
                    bw[i] = make_float2((float)cos(a), (float)sin(a));
                    cr[i] = cos(a); ci[i] = -sin(a);
                    if (i) { cr[M - i] = cr[i]; ci[M - i] = ci[i]; }
                }
                host_fft_pow2(cr, ci);
                for (int j = 0; j <= NH; ++j) bf[j] = make_float2((float)(cr[j] / M), (float)(ci[j] / M));
                HIPCHK(h, hipMemcpy(h->opt.cz_bw, bw.data(), NH * sizeof(float2), hipMemcpyHostToDevice));
                HIPCHK(h, hipMemcpy(h->opt.cz_bf, bf.data(), (NH + 2) * sizeof(float2), hipMemcpyHostToDevice));
            }
            h->opt_T = T;
        }
    } else if ((rc = make_plan(h, h->plan_t, T))) {
        return rc;
    }
    EventPair evp_; hipEvent_t &e0 = evp_.a, &e1 = evp_.b;
    HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, hipEventRecord(e0, h->stream));
    h->cur = nullptr;
    if (fast_t) {
        // time lines are "columns" of a (T, npix) image per probe: 16 neighbouring pixels per tile
        ColJob j{};
        j.in = src; j.out = nullptr; j.px = nullptr; j.tw = tw4_t; j.out_real = dst;
        j.in_image_stride = j.out_image_stride = (long long)T * npix;
        j.in_pitch = j.out_pitch = (int)npix; j.ny = (int)npix; j.n_images = (int)batch;
        j.flags = COL_FWD | COL_INTENSITY; j.scale = 1.f;
        const int saved = h->Rx;
        h->Rx = Rt;
        rc = launch_col_fast(h, j, K_OTHER);
        h->Rx = saved;
        if (rc) { (void)hipFree(tw4_t); return rc; }
    } else if (direct_t) {
        TimeJob j{};
        j.in = src; j.out = dst;
        j.image_stride = (long long)T * npix; j.npix = (int)npix; j.n_images = (int)batch; j.T = T;
        if ((rc = launch_time_direct(h, j))) return rc;
    } else if (split_t) {
        TimeJob j{};
        j.in = src; j.out = dst;
        j.image_stride = (long long)T * npix; j.npix = (int)npix; j.n_images = (int)batch; j.T = T;
        if ((rc = launch_time_split(h, j))) return rc;
    } else if (cz_t) {
        TimeJob j{};
        j.in = src; j.out = dst; j.tw = h->opt.cz_tw; j.bf = h->opt.cz_bf; j.bw = h->opt.cz_bw;
        j.image_stride = (long long)T * npix; j.npix = (int)npix; j.n_images = (int)batch; j.T = T;
        auto launch = [&](auto r_c, auto cols_c, auto vec_c) -> int {
            constexpr int R = decltype(r_c)::value, COLS = decltype(cols_c)::value;
            constexpr bool VEC = decltype(vec_c)::value;
            constexpr int M = R * R, NH = M / 2;
            const size_t lds = ((size_t)M + NH + 2 + NH + (size_t)2 * COLS * tcz_stride(R, T)) * 8;          // two tile buffers
            const long long tiles = ((npix + COLS - 1) / COLS) * batch;
            const int per_cu = std::max(1, std::min(2, (int)((size_t)h->lds_limit / lds)));
            const int grid = (int)std::min<long long>(tiles, (long long)h->n_cus * per_cu);
            (void)hipFuncSetAttribute((const void*)time_cz_kernel<R, COLS, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_limit);
            hipLaunchKernelGGL((time_cz_kernel<R, COLS, VEC>), dim3(grid), dim3(COLS * R), lds, h->stream, j);
            HIPCHK(h, hipGetLastError());
            return mark_launch(h, K_OTHER);
        };
        using I16 = std::integral_constant<int, 16>; using I32 = std::integral_constant<int, 32>;
        const bool even = (npix % 2 == 0);
        if (h->opt.cz_R == 16) rc = even ? launch(I16{}, I32{}, std::true_type{}) : launch(I16{}, I32{}, std::false_type{});
        else rc = even ? launch(I32{}, I16{}, std::true_type{}) : launch(I32{}, I16{}, std::false_type{});
        if (rc) return rc;
    } else {
        LineArgs a;
        a.in = src; a.out = nullptr; a.out_real = dst;
        a.n_lines = (long long)batch * npix; a.lines_per_image = (int)npix;
        a.in_es = a.out_es = npix; a.in_ls = a.out_ls = 1; a.in_is = a.out_is = (long long)T * npix;
        a.contiguous_lines = 1; a.fft1 = +1; a.store_mode = STORE_INTENSITY; a.shift_n = T / 2;
        if ((rc = launch_lines(h, h->plan_t, a, K_OTHER))) return rc;
    }
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    h->ctr.ms_tacaw += ms;
    if (tw4_t) (void)hipFree(tw4_t);
    return MSL_OK;
}

static int ensure_scratch(msl_handle* h, size_t bytes);

int msl_tacaw_stream_begin(msl_handle* h, int32_t T_total, int32_t n_bins, const int32_t* bins) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (!h->wf) return fail(h, MSL_ERR_STATE, "msl_tacaw_stream_begin: handle created with n_frames == 0 (no frame ring)");
    if (T_total < 2) return fail(h, MSL_ERR_INVALID, "msl_tacaw_stream_begin: needs at least 2 frames (got %d)", T_total);
    if (!bins) n_bins = T_total;
    if (n_bins < 1 || n_bins > T_total) return fail(h, MSL_ERR_INVALID, "msl_tacaw_stream_begin: %d bins of %d", n_bins, T_total);
    std::vector<int> b(n_bins);
    for (int i = 0; i < n_bins; ++i) {
        b[i] = bins ? bins[i] : i;
        if (b[i] < 0 || b[i] >= T_total) return fail(h, MSL_ERR_INVALID, "msl_tacaw_stream_begin: bin %d outside [0,%d)", b[i], T_total);
    }
    const msl_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    const size_t PK = (size_t)c.n_probes * h->wpix;
    int rc;
    if ((rc = dalloc(h, &h->st_acc, PK * n_bins))) return rc;
    if ((rc = dalloc(h, &h->st_s1, PK))) return rc;
    if ((rc = dalloc(h, &h->st_s2, PK))) return rc;
    if ((rc = dalloc(h, &h->st_tw, (size_t)T_total))) return rc;
    if ((rc = dalloc(h, &h->st_bins, (size_t)n_bins))) return rc;
    std::vector<float2> tw(T_total);
    for (int m = 0; m < T_total; ++m) { const double a = -2.0 * M_PI * m / T_total; tw[m] = make_float2((float)cos(a), (float)sin(a)); }
    HIPCHK(h, hipMemcpyAsync(h->st_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->st_bins, b.data(), b.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->st_acc, 0, PK * n_bins * sizeof(float2), h->stream));
    HIPCHK(h, hipMemsetAsync(h->st_s1, 0, PK * sizeof(double2), h->stream));
    HIPCHK(h, hipMemsetAsync(h->st_s2, 0, PK * sizeof(double), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));            // the host vectors go out of scope
    h->st_T = T_total; h->st_F = n_bins; h->st_open = true; h->st_have_ref = false;
    return MSL_OK;
}

int msl_tacaw_stream_set_reference(msl_handle* h, const void* d_ref_c64, int32_t slot) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (!h->st_open) return fail(h, MSL_ERR_STATE, "msl_tacaw_stream_set_reference: no open stream (msl_tacaw_stream_begin)");
    const msl_config& c = h->cfg;
    if (!d_ref_c64 && (slot < 0 || slot >= c.n_frames))
        return fail(h, MSL_ERR_INVALID, "msl_tacaw_stream_set_reference: slot %d outside the ring of %d", slot, c.n_frames);
    HIPCHK(h, hipSetDevice(c.device));
    const size_t K = h->wpix;
    int rc;
    if (!h->st_ref && (rc = dalloc(h, &h->st_ref, (size_t)c.n_probes * K))) return rc;
    if (d_ref_c64) {
        HIPCHK(h, hipMemcpyAsync(h->st_ref, d_ref_c64, (size_t)c.n_probes * K * sizeof(float2), hipMemcpyDeviceToDevice, h->stream));
    } else {
        // frame slot `slot` of the (P, ring, K) buffer: P strided images
        HIPCHK(h, hipMemcpy2DAsync(h->st_ref, K * sizeof(float2), h->wf + (size_t)slot * h->wpitch, (size_t)c.n_frames * h->wpitch * sizeof(float2),
                                   K * sizeof(float2), c.n_probes, hipMemcpyDeviceToDevice, h->stream));
    }
    h->st_have_ref = true;
    return MSL_OK;
}

int msl_tacaw_stream_push(msl_handle* h, int32_t first_slot, int32_t count, int32_t t0) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (!h->st_open) return fail(h, MSL_ERR_STATE, "msl_tacaw_stream_push: no open stream (msl_tacaw_stream_begin)");
    const msl_config& c = h->cfg;
    if (count < 1 || first_slot < 0 || first_slot + count > c.n_frames)
        return fail(h, MSL_ERR_INVALID, "msl_tacaw_stream_push: slots [%d,%d) outside the ring of %d", first_slot, first_slot + count, c.n_frames);
    if (t0 < 0 || t0 + count > h->st_T) return fail(h, MSL_ERR_INVALID, "msl_tacaw_stream_push: times [%d,%d) outside [0,%d)", t0, t0 + count, h->st_T);
    HIPCHK(h, hipSetDevice(c.device));
    FoldJob j{};
    j.wf = h->wf; j.acc = h->st_acc; j.s1 = h->st_s1; j.s2 = h->st_s2; j.tw = h->st_tw; j.bins = h->st_bins;
    j.ref = h->st_have_ref ? h->st_ref : nullptr;
    j.K = (long long)h->wpix; j.wfK = (long long)h->wpitch; j.ring = c.n_frames; j.first_slot = first_slot; j.count = count; j.t0 = t0; j.T = h->st_T; j.F = h->st_F;
    const dim3 grid((unsigned)((h->wpix + 255) / 256), c.n_probes);
    for (int f0 = 0; f0 < h->st_F; f0 += MSL_FOLD_FCH) {
        j.f0 = f0;
        hipLaunchKernelGGL(tacaw_fold_kernel, grid, dim3(256), 0, h->stream, j);
    }
    HIPCHK(h, hipGetLastError());
    return MSL_OK;
}

int msl_tacaw_stream_finish_range(msl_handle* h, int32_t p0, int32_t count, void* d_dst_f32, double* total_host) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (!h->st_open) return fail(h, MSL_ERR_STATE, "msl_tacaw_stream_finish: no open stream");
    const msl_config& c = h->cfg;
    if (p0 < 0 || count < 0 || p0 + count > c.n_probes)
        return fail(h, MSL_ERR_INVALID, "msl_tacaw_stream_finish_range: probes [%d,%d) outside [0,%d)", p0, p0 + count, c.n_probes);
    if (!d_dst_f32 && count > 0 && (p0 != 0 || count != c.n_probes))
        return fail(h, MSL_ERR_INVALID, "msl_tacaw_stream_finish_range: a probe sub-range needs a destination (the handle's intensity buffer holds all probes)");
    HIPCHK(h, hipSetDevice(c.device));
    const size_t PK = (size_t)count * h->wpix, need = PK * h->st_F;
    int rc;
    float* dst = (float*)d_dst_f32;
    if (!dst && count > 0) {
        if (h->intensity_elems != need) {
            if ((rc = dalloc(h, &h->intensity, need))) return rc;
            h->intensity_elems = need;
        }
        h->intensity_F = h->st_F; h->intensity_ld = h->wpix;
        dst = h->intensity;
    }
    if (need) {
        hipLaunchKernelGGL(tacaw_stream_finish_kernel, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, h->stream,
                           h->st_acc + (size_t)p0 * h->st_F * h->wpix, dst, h->st_bins, (long long)h->st_F, (long long)h->wpix, (long long)need);
        HIPCHK(h, hipGetLastError());
    }
    if (total_host && PK) {
        if ((rc = ensure_scratch(h, PK * sizeof(double)))) return rc;
        hipLaunchKernelGGL(tacaw_stream_total_kernel, dim3((unsigned)((PK + 255) / 256)), dim3(256), 0, h->stream, h->st_s1 + (size_t)p0 * h->wpix,
                           h->st_s2 + (size_t)p0 * h->wpix, (double)h->st_T, (long long)PK, (double*)h->scratch);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(total_host, h->scratch, PK * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->st_acc); h->st_acc = nullptr;           // the accumulators are the big part: give them back
    h->st_open = false;
    return MSL_OK;
}

int msl_tacaw_stream_finish(msl_handle* h, double* total_PK) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    return msl_tacaw_stream_finish_range(h, 0, h->cfg.n_probes, nullptr, total_PK);
}

size_t msl_buffer_bytes(const msl_handle* h, msl_buffer what) {
    if (!h) return 0;
    const msl_config& c = h->cfg;
    const size_t npix = (size_t)c.nx * c.ny;
    switch (what) {
        case MSL_BUF_PROBES: case MSL_BUF_EXIT: return npix * c.n_probes * 8;
        case MSL_BUF_POTENTIAL: return h->V ? npix * c.nz * 4 : 0;
        case MSL_BUF_TRANSMISSION: return npix * c.nz * 8;
        case MSL_BUF_WAVEFUNCTION: return h->wf ? h->wpitch * c.n_probes * c.n_frames * 8 : 0;
        case MSL_BUF_INTENSITY: return h->intensity_elems * 4;
        case MSL_BUF_FORMFACTOR: return npix * h->n_species * 4;
        case MSL_BUF_STREAM_ACC: return (h->st_open && h->st_acc) ? h->wpix * c.n_probes * (size_t)h->st_F * 8 : 0;
        case MSL_BUF_STREAM_S1: return (h->st_open && h->st_s1) ? h->wpix * c.n_probes * 16 : 0;
        case MSL_BUF_STREAM_S2: return (h->st_open && h->st_s2) ? h->wpix * c.n_probes * 8 : 0;
        case MSL_BUF_STREAM_REF: return (h->st_open && h->st_have_ref) ? h->wpix * c.n_probes * 8 : 0;
    }
    return 0;
}

int64_t msl_result_pitch(const msl_handle* h, msl_buffer what) {
    if (!h) return 0;
    if (what == MSL_BUF_WAVEFUNCTION) return h->wf ? (int64_t)h->wpitch : 0;
    if (what == MSL_BUF_INTENSITY) return h->intensity ? (int64_t)h->intensity_ld : 0;
    return 0;
}

void* msl_device_ptr(msl_handle* h, msl_buffer what) {
    if (!h) return nullptr;
    switch (what) {
        case MSL_BUF_PROBES: return h->psi0;
        case MSL_BUF_EXIT: return h->psi;
        case MSL_BUF_POTENTIAL: return h->V;
        case MSL_BUF_TRANSMISSION: return h->trans ? h->trans + (size_t)h->cur_batch * h->cfg.nz * h->cfg.nx * h->cfg.ny : nullptr;
        case MSL_BUF_WAVEFUNCTION: return h->wf;
        case MSL_BUF_INTENSITY: return h->intensity;
        case MSL_BUF_FORMFACTOR: return h->d_ff;
        case MSL_BUF_STREAM_ACC: return h->st_open ? h->st_acc : nullptr;
        case MSL_BUF_STREAM_S1: return h->st_open ? h->st_s1 : nullptr;
        case MSL_BUF_STREAM_S2: return h->st_open ? h->st_s2 : nullptr;
        case MSL_BUF_STREAM_REF: return (h->st_open && h->st_have_ref) ? h->st_ref : nullptr;
    }
    return nullptr;
}

// ---- reductions over resident results (reduce.h) ------------------------------------------------------
static int ensure_scratch(msl_handle* h, size_t bytes) {
    if (bytes <= h->scratch_bytes) return MSL_OK;
    int rc = dalloc(h, &h->scratch, bytes);
    h->scratch_bytes = rc ? 0 : bytes;
    return rc;
}

// resolve (src, B, F, K, ld) for the TACAW reductions: NULL = the handle's intensity buffer
static int intensity_source(msl_handle* h, const char* who, const void** src, int64_t* B, int64_t* F, int64_t* K, int64_t* ld) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    if (!*src) {
        if (!h->intensity || h->intensity_elems == 0) return fail(h, MSL_ERR_STATE, "%s: no intensity (call msl_tacaw)", who);
        *src = h->intensity; *B = h->cfg.n_probes; *F = h->intensity_F; *K = (int64_t)h->wpix; *ld = (int64_t)h->intensity_ld;
    } else if (*ld == 0) {
        *ld = *K;
    }
    if (*B < 1 || *F < 1 || *K < 1) return fail(h, MSL_ERR_INVALID, "%s: bad shape (%lld,%lld,%lld)", who, (long long)*B, (long long)*F, (long long)*K);
    if (*ld < *K) return fail(h, MSL_ERR_INVALID, "%s: row pitch %lld below the row length %lld", who, (long long)*ld, (long long)*K);
    if (*B * *F > 0x7fffffffLL) return fail(h, MSL_ERR_UNSUPPORTED, "%s: more than 2^31 rows", who);
    return MSL_OK;
}

// sum over K of rows of a (rows, K) array (row pitch ld) with an optional host mask; float64 result per row on the host
static int reduce_rows(msl_handle* h, const void* src, bool complex_abs, int64_t rows, int64_t K, int64_t ld, const uint8_t* mask, double* out) {
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int64_t quads = (K + 3) / 4;
    int n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(quads / 1024, 4096 / rows), 64));
    const size_t mask_bytes = mask ? (((size_t)K + 15) & ~(size_t)15) + 16 : 0;
    const size_t part_bytes = (size_t)rows * n_chunks * sizeof(double);
    int rc = ensure_scratch(h, mask_bytes + part_bytes);
    if (rc) return rc;
    uint8_t* d_mask = mask ? (uint8_t*)h->scratch : nullptr;
    double* d_part = (double*)(h->scratch + mask_bytes);
    if (mask) {
        HIPCHK(h, hipMemsetAsync(d_mask, 0, mask_bytes, h->stream));
        HIPCHK(h, hipMemcpyAsync(d_mask, mask, (size_t)K, hipMemcpyHostToDevice, h->stream));
    }
    dim3 grid(n_chunks, (unsigned)rows);
    if (complex_abs) hipLaunchKernelGGL(reduce_k_kernel<true>, grid, dim3(256), 0, h->stream, src, d_mask, (long long)K, (long long)ld, n_chunks, d_part);
    else hipLaunchKernelGGL(reduce_k_kernel<false>, grid, dim3(256), 0, h->stream, src, d_mask, (long long)K, (long long)ld, n_chunks, d_part);
    HIPCHK(h, hipGetLastError());
    std::vector<double> part((size_t)rows * n_chunks);
    HIPCHK(h, hipMemcpyAsync(part.data(), d_part, part_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int64_t r = 0; r < rows; ++r) {
        double s = 0;
        for (int c = 0; c < n_chunks; ++c) s += part[(size_t)r * n_chunks + c];
        out[r] = s;
    }
    return MSL_OK;
}

int msl_tacaw_spectrum(msl_handle* h, const void* d_src_f32, int64_t B, int64_t F, int64_t K, int64_t ld, const uint8_t* mask, double* out) {
    if (!out) return fail(h, MSL_ERR_INVALID, "msl_tacaw_spectrum: null output");
    int rc = intensity_source(h, "msl_tacaw_spectrum", &d_src_f32, &B, &F, &K, &ld);
    if (rc) return rc;
    if (B * F > 65535) {                       // grid.y limit: go probe by probe
        if (F > 65535) return fail(h, MSL_ERR_UNSUPPORTED, "msl_tacaw_spectrum: more than 65535 frequencies");
        for (int64_t b = 0; b < B; ++b)
            if ((rc = reduce_rows(h, (const float*)d_src_f32 + b * F * ld, false, F, K, ld, mask, out + b * F))) return rc;
        return MSL_OK;
    }
    return reduce_rows(h, d_src_f32, false, B * F, K, ld, mask, out);
}

int msl_tacaw_spectrum_weighted(msl_handle* h, const void* d_src_f32, int64_t B, int64_t F, int64_t K, int64_t ld, const double* weight, double* out) {
    if (!out || !weight) return fail(h, MSL_ERR_INVALID, "msl_tacaw_spectrum_weighted: null argument");
    int rc = intensity_source(h, "msl_tacaw_spectrum_weighted", &d_src_f32, &B, &F, &K, &ld);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int64_t rows_per = std::min<int64_t>(B * F, 32768);
    const int n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(K / 4096, 4096 / std::max<int64_t>(1, rows_per)), 64));
    const size_t w_bytes = (size_t)K * sizeof(double), part_bytes = (size_t)rows_per * n_chunks * sizeof(double);
    if ((rc = ensure_scratch(h, w_bytes + part_bytes))) return rc;
    double* d_w = (double*)h->scratch;
    double* d_part = (double*)(h->scratch + w_bytes);
    HIPCHK(h, hipMemcpyAsync(d_w, weight, w_bytes, hipMemcpyHostToDevice, h->stream));
    std::vector<double> part((size_t)rows_per * n_chunks);
    for (int64_t r0 = 0; r0 < B * F; r0 += rows_per) {
        const int64_t rows = std::min<int64_t>(rows_per, B * F - r0);
        hipLaunchKernelGGL(reduce_kw_kernel, dim3(n_chunks, (unsigned)rows), dim3(256), 0, h->stream, (const float*)d_src_f32 + r0 * ld, d_w,
                           (long long)K, (long long)ld, n_chunks, d_part);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(part.data(), d_part, (size_t)rows * n_chunks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int64_t r = 0; r < rows; ++r) {
            double s = 0;
            for (int c = 0; c < n_chunks; ++c) s += part[(size_t)r * n_chunks + c];
            out[r0 + r] = s;
        }
    }
    return MSL_OK;
}

int msl_adf(msl_handle* h, const void* d_src_c64, int64_t B, int64_t T, int64_t K, int64_t ld, const uint8_t* mask, double* out) {
    if (!h || !out) return fail(h, MSL_ERR_INVALID, "msl_adf: null argument");
    if (!d_src_c64) {
        if (!h->wf) return fail(h, MSL_ERR_STATE, "msl_adf: no wavefunction buffer");
        d_src_c64 = h->wf; B = h->cfg.n_probes; T = h->cfg.n_frames; K = (int64_t)h->wpix; ld = (int64_t)h->wpitch;
    } else if (ld == 0) {
        ld = K;
    }
    if (B < 1 || T < 1 || K < 1 || ld < K) return fail(h, MSL_ERR_INVALID, "msl_adf: bad shape");
    if (T > 65535) return fail(h, MSL_ERR_UNSUPPORTED, "msl_adf: more than 65535 frames");
    std::vector<double> rows((size_t)T);
    for (int64_t b = 0; b < B; ++b) {
        int rc = reduce_rows(h, (const float2*)d_src_c64 + b * T * ld, true, T, K, ld, mask, rows.data());
        if (rc) return rc;
        double s = 0;
        for (double v : rows) s += v;
        out[b] = s / (double)T;                // mean over frames of the annulus sum (haadf_data.py:80)
    }
    return MSL_OK;
}

int msl_tacaw_diffraction(msl_handle* h, const void* d_src_f32, int64_t B, int64_t F, int64_t K, int64_t ld, int64_t b0, int64_t b1,
                          int64_t f0, int64_t f1, double scale, double* out) {
    if (!out) return fail(h, MSL_ERR_INVALID, "msl_tacaw_diffraction: null output");
    int rc = intensity_source(h, "msl_tacaw_diffraction", &d_src_f32, &B, &F, &K, &ld);
    if (rc) return rc;
    if (b0 < 0 || b1 > B || b0 >= b1 || f0 < 0 || f1 > F || f0 >= f1)
        return fail(h, MSL_ERR_INVALID, "msl_tacaw_diffraction: range [%lld,%lld) x [%lld,%lld) outside (%lld,%lld)", (long long)b0,
                    (long long)b1, (long long)f0, (long long)f1, (long long)B, (long long)F);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if ((rc = ensure_scratch(h, (size_t)K * sizeof(double)))) return rc;
    double* d_out = (double*)h->scratch;
    const bool vec = (K % 4 == 0) && (ld % 4 == 0);
    const long long threads = vec ? K / 4 : K;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (vec) hipLaunchKernelGGL(reduce_bf_kernel<true>, dim3(grid), dim3(256), 0, h->stream, (const float*)d_src_f32, (long long)F, (long long)K,
                                (long long)ld, (long long)b0, (long long)b1, (long long)f0, (long long)f1, scale, d_out);
    else hipLaunchKernelGGL(reduce_bf_kernel<false>, dim3(grid), dim3(256), 0, h->stream, (const float*)d_src_f32, (long long)F, (long long)K,
                            (long long)ld, (long long)b0, (long long)b1, (long long)f0, (long long)f1, scale, d_out);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, d_out, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MSL_OK;
}

int msl_tacaw_dispersion(msl_handle* h, const void* d_src_f32, int64_t B, int64_t F, int64_t K, int64_t ld, const int64_t* idx, int64_t n, float* out) {
    if (!out || !idx) return fail(h, MSL_ERR_INVALID, "msl_tacaw_dispersion: null argument");
    int rc = intensity_source(h, "msl_tacaw_dispersion", &d_src_f32, &B, &F, &K, &ld);
    if (rc) return rc;
    if (n < 1) return fail(h, MSL_ERR_INVALID, "msl_tacaw_dispersion: empty path");
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= K) return fail(h, MSL_ERR_INVALID, "msl_tacaw_dispersion: index %lld outside [0,%lld)", (long long)idx[i], (long long)K);
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t idx_bytes = (size_t)n * sizeof(int64_t), out_bytes = (size_t)B * F * n * sizeof(float);
    if ((rc = ensure_scratch(h, idx_bytes + out_bytes))) return rc;
    long long* d_idx = (long long*)h->scratch;
    float* d_out = (float*)(h->scratch + idx_bytes);
    HIPCHK(h, hipMemcpyAsync(d_idx, idx, idx_bytes, hipMemcpyHostToDevice, h->stream));
    const long long tot = (long long)B * F * n;
    hipLaunchKernelGGL(gather_k_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (const float*)d_src_f32,
                       (long long)(B * F), (long long)ld, d_idx, (long long)n, d_out);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MSL_OK;
}

int msl_download(msl_handle* h, msl_buffer what, void* dst, size_t bytes, int64_t first, int64_t count) {
    if (!h || !dst) return fail(h, MSL_ERR_INVALID, "msl_download: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const char* src = (const char*)msl_device_ptr(h, what);
    size_t total = msl_buffer_bytes(h, what);
    if (!src || total == 0) return fail(h, MSL_ERR_STATE, "msl_download: buffer %d not available", (int)what);
    if (what == MSL_BUF_EXIT && !h->have_exit) return fail(h, MSL_ERR_STATE, "msl_download: no exit waves (call msl_propagate)");
    if (what == MSL_BUF_TRANSMISSION && h->onepass && h->have_potential) {
        // the one-pass loop keeps every second slice transposed in its own buffer: restore the natural copies
        const size_t npix = (size_t)h->cfg.nx * h->cfg.ny;
        const int first = slice_is_transposed(h, 0) ? 0 : 1;
        const int count = (h->cfg.nz - first + 1) / 2;
        for (int z0 = 0; z0 < count; z0 += 65535) {
            const int nzb = std::min(65535, count - z0);
            dim3 grid((h->cfg.nx + 31) / 32, (h->cfg.ny + 31) / 32, nzb);
            const size_t off = (size_t)h->cur_batch * h->cfg.nz * npix + (size_t)(first + 2 * z0) * npix;
            hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, h->stream, h->transT + off, h->trans + off,
                               h->cfg.ny, h->cfg.nx, h->cfg.nx, h->cfg.ny, (long long)(2 * npix), (long long)(2 * npix));
        }
        HIPCHK(h, hipGetLastError());
    }
    size_t off = 0, len = total;
    if (count > 0 && what != MSL_BUF_WAVEFUNCTION && what != MSL_BUF_INTENSITY)
        return fail(h, MSL_ERR_INVALID, "msl_download: ranges only for wavefunction/intensity");
    if (what == MSL_BUF_WAVEFUNCTION || what == MSL_BUF_INTENSITY) {
        // (P, rows, ld) on the device -> dense (P, rows, wx*wy) on the host
        const size_t es = what == MSL_BUF_WAVEFUNCTION ? sizeof(float2) : sizeof(float);
        const size_t ld = what == MSL_BUF_WAVEFUNCTION ? h->wpitch : h->intensity_ld;
        const size_t rows_per_probe = total / es / ld / h->cfg.n_probes;
        size_t n_probes = h->cfg.n_probes;
        if (count > 0) {
            if (first < 0 || first + count > h->cfg.n_probes) return fail(h, MSL_ERR_INVALID, "msl_download: probe range out of bounds");
            off = (size_t)first * rows_per_probe * ld * es; n_probes = (size_t)count;
        }
        const size_t rows = n_probes * rows_per_probe;
        len = rows * h->wpix * es;
        if (bytes != len) return fail(h, MSL_ERR_INVALID, "msl_download: dst holds %zu bytes, buffer slice is %zu", bytes, len);
        if (ld == h->wpix) {
            HIPCHK(h, hipMemcpyAsync(dst, src + off, len, hipMemcpyDeviceToHost, h->stream));
        } else {
            HIPCHK(h, hipMemcpy2DAsync(dst, h->wpix * es, src + off, ld * es, h->wpix * es, rows, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return MSL_OK;
    }
    if (bytes != len) return fail(h, MSL_ERR_INVALID, "msl_download: dst holds %zu bytes, buffer slice is %zu", bytes, len);
    if ((what == MSL_BUF_PROBES || what == MSL_BUF_EXIT) && h->pitch != h->cfg.ny) {
        HIPCHK(h, hipMemcpy2DAsync(dst, (size_t)h->cfg.ny * sizeof(float2), src, (size_t)h->pitch * sizeof(float2),
                                   (size_t)h->cfg.ny * sizeof(float2), (size_t)h->cfg.n_probes * h->cfg.nx, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return MSL_OK;
    }
    HIPCHK(h, hipMemcpyAsync(dst, src + off, len, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MSL_OK;
}

static int frame_copy(msl_handle* h, int32_t slot, void* host, size_t bytes, bool to_host) {
    if (!h || !host) return fail(h, MSL_ERR_INVALID, "frame copy: null argument");
    if (!h->wf) return fail(h, MSL_ERR_STATE, "frame copy: handle created with n_frames == 0");
    const msl_config& c = h->cfg;
    if (slot < 0 || slot >= c.n_frames) return fail(h, MSL_ERR_INVALID, "frame copy: slot %d out of range [0,%d)", slot, c.n_frames);
    const size_t npix = h->wpix;                    // one (wx, wy) image of the stored window
    if (bytes != npix * c.n_probes * sizeof(float2)) return fail(h, MSL_ERR_INVALID, "frame copy: buffer holds %zu bytes, a frame is %zu", bytes, npix * c.n_probes * sizeof(float2));
    HIPCHK(h, hipSetDevice(c.device));
    // (P, T, wx, wy) device <-> (P, wx, wy) host: P strided blocks of one image
    float2* dev = h->wf + (size_t)slot * h->wpitch;
    if (to_host)
        HIPCHK(h, hipMemcpy2DAsync(host, npix * sizeof(float2), dev, (size_t)c.n_frames * h->wpitch * sizeof(float2), npix * sizeof(float2),
                                   c.n_probes, hipMemcpyDeviceToHost, h->stream));
    else
        HIPCHK(h, hipMemcpy2DAsync(dev, (size_t)c.n_frames * h->wpitch * sizeof(float2), host, npix * sizeof(float2), npix * sizeof(float2),
                                   c.n_probes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MSL_OK;
}

int msl_download_wavefunction_c128(msl_handle* h, int32_t n_frames_used, void* dst, size_t bytes) {
    if (!h || !dst) return fail(h, MSL_ERR_INVALID, "msl_download_wavefunction_c128: null argument");
    const msl_config& c = h->cfg;
    if (!h->wf) return fail(h, MSL_ERR_STATE, "msl_download_wavefunction_c128: no wavefunction buffer");
    if (n_frames_used < 1 || n_frames_used > c.n_frames) return fail(h, MSL_ERR_INVALID, "msl_download_wavefunction_c128: %d of %d frames", n_frames_used, c.n_frames);
    const size_t per_probe = (size_t)n_frames_used * h->wpix;
    if (bytes != (size_t)c.n_probes * per_probe * sizeof(double2))
        return fail(h, MSL_ERR_INVALID, "msl_download_wavefunction_c128: dst holds %zu bytes, the result has %zu", bytes, (size_t)c.n_probes * per_probe * sizeof(double2));
    HIPCHK(h, hipSetDevice(c.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // chunks of at most 256 MB of complex128 through the scratch buffer, probe by probe (a probe's used frames follow each other
    // at the image pitch); a chunk is a whole number of images or a piece of one
    size_t chunk = std::min<size_t>(per_probe, (size_t)(256u << 20) / sizeof(double2));
    if (chunk > h->wpix) chunk -= chunk % h->wpix;
    if (const char* e = dbg_env("MSL_C128_CHUNK")) chunk = std::max<size_t>(1, std::min<size_t>(chunk, (size_t)atoll(e)));      // (tests: several chunks per probe)
    int rc = ensure_scratch(h, chunk * sizeof(double2));
    if (rc) return rc;
    for (int p = 0; p < c.n_probes; ++p) {
        const float2* src = h->wf + (size_t)p * c.n_frames * h->wpitch;
        double2* out = (double2*)dst + (size_t)p * per_probe;
        for (size_t o = 0; o < per_probe; ) {
            // dense offset o = image o / wpix, pixel o % wpix; a piece that starts inside an image ends with it
            const size_t img = o / h->wpix, px = o % h->wpix;
            const size_t n = std::min(px ? std::min(chunk, h->wpix - px) : chunk, per_probe - o);
            const int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)h->n_cus * 8);
            hipLaunchKernelGGL(widen_c64_kernel, dim3(grid), dim3(256), 0, h->stream, src + img * h->wpitch + px, (double2*)h->scratch, (long long)n,
                               (long long)h->wpix, (long long)h->wpitch);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(out + o, h->scratch, n * sizeof(double2), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            o += n;
        }
    }
    return MSL_OK;
}

int msl_download_frame(msl_handle* h, int32_t slot, void* dst, size_t bytes) { return frame_copy(h, slot, dst, bytes, true); }
int msl_upload_frame(msl_handle* h, int32_t slot, const void* src, size_t bytes) { return frame_copy(h, slot, const_cast<void*>(src), bytes, false); }

int msl_synchronize(msl_handle* h) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MSL_OK;
}

int msl_get_counters(const msl_handle* hc, msl_counters* out) {
    msl_handle* h = const_cast<msl_handle*>(hc);
    if (!h || !out) return fail(h, MSL_ERR_INVALID, "msl_get_counters: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int rc = resolve_all(h);
    if (rc) return rc;
    h->ctr.slice_kernel_launches = h->n_kind[K_ROW] + h->n_kind[K_COL];
    h->ctr.ms_slice_kernels = h->ms_kind[K_ROW] + h->ms_kind[K_COL];
    h->ctr.row_launches = h->n_kind[K_ROW]; h->ctr.ms_row = h->ms_kind[K_ROW];
    h->ctr.col_launches = h->n_kind[K_COL]; h->ctr.ms_col = h->ms_kind[K_COL];
    *out = h->ctr;
    return MSL_OK;
}

int msl_reset_counters(msl_handle* h) {
    if (!h) return fail(h, MSL_ERR_INVALID, "null handle");
    int rc = resolve_all(h);
    if (rc) return rc;
    h->ctr = msl_counters{};
    for (int k = 0; k < K_NKINDS; ++k) { h->ms_kind[k] = 0; h->n_kind[k] = 0; }
    return MSL_OK;
}

int msl_fft2_host(msl_handle* h, const float* in, float* out, int32_t batch, int32_t dir) {
    if (!h || !in || !out) return fail(h, MSL_ERR_INVALID, "msl_fft2_host: null argument");
    if (batch < 1 || (dir != 1 && dir != -1)) return fail(h, MSL_ERR_INVALID, "msl_fft2_host: bad batch/dir");
    const msl_config& c = h->cfg;
    HIPCHK(h, hipSetDevice(c.device));
    const size_t n = (size_t)batch * c.nx * c.ny;
    float2* buf = nullptr;
    int rc = dalloc(h, &buf, n);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(buf, in, n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
    h->cur = nullptr;
    rc = fft2_inplace(h, buf, batch, dir, dir < 0 ? 1.0f / ((float)c.nx * (float)c.ny) : 1.0f, c.ny);
    if (rc == MSL_OK) {
        hipError_t e = hipMemcpyAsync(out, buf, n * sizeof(float2), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, MSL_ERR_HIP, "msl_fft2_host copy back: %s", hipGetErrorString(e));
    }
    (void)hipFree(buf);
    return rc;
}

}  // extern "C"
