// The transposing slice-loop pass for lines of N = R^2 points (R = 32 -> 1024, R = 16 -> 256): the dominant kernel of the
// multislice loop (reference Propagate, src/multislice/multislice.py:278-294), one HBM read and one write of the wave functions
// per slice.  A pass along one axis is
//
//     out^T = A . t_k . A  in          A = ifft . P . fft   (Fresnel propagation along the line axis, multislice.py:290-294)
//                                      t_k = exp(i sigma V_k) (transmission function, multislice.py:285-288)
//
// on contiguous lines (coalesced, the next line prefetched in registers), with the result written TRANSPOSED through an LDS tile
// of 16 lines so that HBM sees 128-byte segments; the next pass then again reads contiguous lines -- of the other axis.
// The kernel is a template only (no code here): its instantiations live in slice_pass.hip, a translation unit of its own.
#pragma once
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>
#include "fft_regs.h"
#include "kernel_util.h"

// Ablation switches of tools/rowt_bench.hip (timing experiments on this pass; wrong results): 1 = no prefetch loads, 2 = no
// global stores (the tile reads stay), 4 = no lane <-> register exchanges, 8 = no transforms at all, 16 = one table entry instead
// of the table reads, 32 = no tile write / barriers / store phase.  Never set in the library.
#ifndef MSL_ABL2
#define MSL_ABL2 0
#endif
#ifndef MSL_DIT_LCH
#define MSL_DIT_LCH 8           // leaf butterflies per chunk of table reads (4: 13 VGPRs fewer, 1 % slower)
#endif
#ifndef MSL_DIT_FENCE
#define MSL_DIT_FENCE 0         // butterflies between two scheduling barriers in the upper levels (0: none; 1, 2: no faster)
#endif
#ifndef MSL_STAGGER
#define MSL_STAGGER 4           // s_sleep units (64 cycles) between the iteration starts of the waves on different SIMDs (0: none)
#endif
#define MSL_IC(x) std::integral_constant<int, (x)>{}

namespace msl {

enum { P2_PRE_A = 1, P2_POST_A = 2, P2_POST_F = 4, P2_IN_PAIRED = 8, P2_OUT_PAIRED = 16 };

struct RowTJob {
    const float2* in;       // (P, n_lines, in_pitch): lines along the transform axis
    float2* out;            // (P, N, out_pitch): transposed
    const float2* trans;    // t_k in the input orientation, (n_lines, N) unpadded
    const float2* pl;       // (N) Fresnel factor along the line axis, 1/N folded in (split order for N = 2R^2)
    const float2* tw;
    const float2* tw2;      // N = 2R^2 only: W_N^m, m < R^2
    long long in_image_stride, out_image_stride;
    int in_pitch, out_pitch, n_lines, n_images, flags, pchunk;
    int t_group;            // frame batching: images [g t_group, (g+1) t_group) use the stack trans + g t_stride (0: one stack)
    unsigned t_magic;       // floor(2^32 / t_group) + 1
    long long t_stride;
    // rowTB_pass_kernel (lines of any length N <= R^2/2 by Bluestein's chirp-z on the register FFTs of length M = R^2):
    const float2* bf;       // (M/2 + 1) filter FFT_M(conj chirp, wrapped) / M -- an even sequence, first half stored
    const float2* bw;       // (M/2) chirp w[n] = exp(-i pi n^2 / N), zero for n >= N
    int n_line;             // N
    int perm_shift;         // rowT_pass_kernel<.., OUT_P>: log2(R' / 8), R' = radix of the kernel that reads the output lines
#ifdef MSL_CLOCK
    unsigned long long* clk;    // tools/rowt_bench.hip -DMSL_CLOCK: per workgroup, shader cycles and 100 MHz ticks spent in the item loop
#endif
};

// transmission stack of the frame that image p belongs to: job.trans + frame_off(job, p).  p is wave-uniform and the
// frame number p / t_group is computed in scalar registers only -- a multiply-high by t_magic = floor(2^32 / t_group) + 1,
// exact while p * t_group < 2^32 -- because these kernels run within a few VGPRs of the 256 that two waves per SIMD allow:
// a vector temporary here pushed rowT2_pass_kernel<16> into the AGPRs and halved its occupancy (81 -> 117 us per pass).
template <typename Job>
__device__ __forceinline__ long long frame_off(const Job& job, int p) {
    if (job.t_group <= 0) return 0;
    const unsigned up = (unsigned)__builtin_amdgcn_readfirstlane(p);
    const unsigned f = job.t_group == 1 ? up : __umulhi(up, job.t_magic);        // (the magic number of 1 does not fit 32 bits)
    return (long long)f * job.t_stride;
}

// ---- four-step transform on the decimation-in-time network of fft_regs.h (fused multiply-adds) --------------------------------
// A line of N = R^2 points lives in a group of R lanes x R registers, element n = reg R + lane before and after a transform:
// register FFT over `reg`, twiddles W_N^{lane k1}, lane <-> register exchange through the LDS, register FFT.  Every pointwise
// product of the pass -- the inter-FFT twiddles, the Fresnel factor, the transmission function -- sits in front of a register FFT
// and is handed to its leaf level as weights (10 instructions per weighted radix-2 leaf butterfly instead of 4 + 4 + 4), the
// twiddles therefore AFTER the exchange (the table is symmetric: the weight of element n2 in lane k1 is T[n2 R + k1]):
//     dit | X | dit(T) | dit(P) | X | dit(conj T) | dit(t_k) | X | dit(T) | dit(P) | X | dit(conj T)        (X = exchange)
// 388 + 7 x 484 instructions per 32 registers where the decimation-in-frequency form of rounds 1-3 (fourstep_split_addtid in
// fft_pow2.h, still used by the other line lengths) spends 8 x ~430 + 7 x 128: 10 % fewer, 3 % less time (the pass is bound by
// the chip's power limit, see DESIGN.md section 4.1, so instruction counts convert at about a third).

// Lane <-> register exchange with ds_write_addtid_b32 stores and 16-byte reads: the store address is M0 + offset + 4 * lane, so it
// needs no address register and runs at twice the rate of ds_write_b32 (128 B/clk: MI355X_MICROARCH.md, LDS).  The 64 / R line
// groups of a wave share one scratch of R rows x 68 floats: row k1 holds the k1-th register of all 64 lanes (group g at columns
// [g R, (g+1) R)), and lane (g, l) reads back row l, columns g R + n2, as R/4 ds_read_b128 (row pitch 68: 16-byte aligned,
// conflict-free); real parts first, then the imaginary parts through the same scratch.  wave_scratch: LDS address (bytes) of the
// wave's scratch, wave-uniform.  M0 is not used by anything else in these kernels, which tests/test_abi_and_host.py checks on the
// ISA (every write of M0 in the library is one of these s_mov), and msl_create runs a one-workgroup self-test of the exchange
// (msl_selftest_exchange); an s_mov to M0 needs a wait state before an add-tid instruction and the hazard recogniser does not
// see into inline asm, hence the s_nop.
template <int R>
__device__ __forceinline__ void exchange_addtid(float2 (&v)[R], const float* scratch_base, unsigned wave_scratch, int ln, int lane64) {
    static_assert(64 % R == 0 && R % 4 == 0, "R-lane groups inside one wave; rows are read four floats at a time");
    constexpr int PW = 68;
    if constexpr (MSL_ABL2 & 4) return;
    const float* rd = scratch_base + ln * PW + (lane64 / R) * R;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" :: "s"(wave_scratch) : "memory");
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(v[k1].x), "n"(k1 * PW * 4) : "memory");
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < R / 4; ++g) {
        const float4 q = *reinterpret_cast<const float4*>(rd + 4 * g);
        v[4 * g].x = q.x; v[4 * g + 1].x = q.y; v[4 * g + 2].x = q.z; v[4 * g + 3].x = q.w;
    }
    wave_lds_fence();
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 1" :: "s"(wave_scratch) : "memory");
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(v[k1].y), "n"(k1 * PW * 4) : "memory");
    wave_lds_fence();
#pragma unroll
    for (int g = 0; g < R / 4; ++g) {
        const float4 q = *reinterpret_cast<const float4*>(rd + 4 * g);
        v[4 * g].y = q.x; v[4 * g + 1].y = q.y; v[4 * g + 2].y = q.z; v[4 * g + 3].y = q.w;
    }
    wave_lds_fence();
}

// leaf butterflies C0 + J of the R-point network with the weights tab[n * STRIDE + ln] of their elements n fetched per chunk
template <int R, bool INV, int WMODE, int STRIDE, int C0, typename TabPtr, int... J>
__device__ __forceinline__ void dit_leaf_chunk(const float2* in, float2* v, TabPtr tab, int ln, std::integer_sequence<int, J...>) {
    constexpr int LR = dit_leaf_radix(R), C = R / LR;
    float2 w[sizeof...(J)][4];
    if constexpr (MSL_ABL2 & 16) {              // timing experiment: one table entry for all
        ((w[J][0] = tab[ln], w[J][1] = tab[ln], w[J][2] = tab[ln], w[J][3] = tab[ln]), ...);
    } else {
        ((w[J][0] = tab[(C0 + J) * STRIDE + ln], w[J][1] = tab[(C0 + J + C) * STRIDE + ln]), ...);
        if constexpr (LR == 4) ((w[J][2] = tab[(C0 + J + 2 * C) * STRIDE + ln], w[J][3] = tab[(C0 + J + 3 * C) * STRIDE + ln]), ...);
    }
    (dit_leaf<R, INV, WMODE, C0 + J>(in, v, w[J]), ...);
    (pin(v[C0 + J]), ...); (pin(v[C0 + J + C]), ...);
    if constexpr (LR == 4) { (pin(v[C0 + J + 2 * C]), ...); (pin(v[C0 + J + 3 * C]), ...); }
    __builtin_amdgcn_sched_barrier(0);
}
// the whole leaf level of in[] (may be v itself) into v[], element n weighted by tab[n * STRIDE + ln] (WMODE 2: conjugated);
// CH = leaf butterflies per chunk of table reads (one exposed LDS round trip per chunk, CH x leaf radix weights in registers)
template <int R, bool INV, int WMODE, int STRIDE, int CH, int C0, typename TabPtr>
__device__ __forceinline__ void dit_leaf_chunks(const float2* in, float2* v, TabPtr tab, int ln) {
    if constexpr (C0 < dit_leaf_count(R)) {
        dit_leaf_chunk<R, INV, WMODE, STRIDE, C0>(in, v, tab, ln, std::make_integer_sequence<int, CH>{});
        dit_leaf_chunks<R, INV, WMODE, STRIDE, CH, C0 + CH>(in, v, tab, ln);
    }
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------------
// Work item = (block of 16 lines, chunk of `pchunk` images that share t_k): the t_k lines stay in registers across the chunk.
//
// IN_P / OUT_P: between two transposing passes the work buffer holds every line in the INTERLEAVED order
//     position 2 R' (j >> 1) + 2 l + (j & 1)   <->   element R' j + l          (R' = radix of the kernel that reads the line)
// i.e. the two elements a lane of the reading kernel keeps in registers 2 jp and 2 jp + 1 sit next to each other: the reader
// fetches a line with R'/2 loads of 16 bytes per lane instead of R' loads of 8 (a wave's load covers 2 x 512 contiguous bytes;
// 8-byte accesses run at 0.54-0.70 of the 16-byte rate through the L1 / address path, MI355X_MICROARCH.md).  The writer keeps its
// 128-byte transposed segments: a tile is then not 16 consecutive lines but 8 lines l0 .. l0 + 7 of block j = 2 jp and the same 8
// of block 2 jp + 1 (tile row r = line R' (2 jp + (r & 1)) + l0 + (r >> 1)), whose 16 output elements are exactly positions
// 2 R' jp + 2 l0 .. + 15: the store addresses do not change at all, only which input lines (and t_k lines) form a tile.
// job.perm_shift = log2(R' / 8).
//
// FL: the P2_PRE_A / P2_POST_A bits of job.flags as a compile-time value (3 = both: every pass but the first and the last of a
// stack), or -1 = read them from the job.  With both halves unconditional the body needs 234 VGPRs; with branches around them the
// allocator spills the t_k line (272 bytes per lane) -- only the rarely used combinations are compiled that way (slice_pass.hip).
template <int R, int LINES, bool IN_P, bool OUT_P, int FL>
__global__ void __launch_bounds__(LINES * R, (R == 16) ? 3 : 2) rowT_pass_kernel(RowTJob job) {
    constexpr int N = R * R;
    constexpr int NT = LINES * R;
    // tile line pitch in float2: the R^2 positions of a line (the wave's exchange scratch, R x 68 floats over 64 / R lines, is
    // smaller); 2 mod 32: rows 16-byte aligned for the exchange's wide reads, conflict-free staging
    constexpr int CS = (R * R + 33) / 32 * 32 + 2;
    constexpr int TPS = LINES / 2;                    // threads (16 B = 2 lines each) per output segment of LINES*8 bytes
    constexpr int POS_PER_IT = NT / TPS;
    constexpr int NIT = N / POS_PER_IT;
    constexpr int LCHL = MSL_DIT_LCH < dit_leaf_count(R) ? MSL_DIT_LCH : dit_leaf_count(R);      // leaf butterflies per chunk of table reads
    constexpr int FN = MSL_DIT_FENCE;
    static_assert(LINES == 16, "16 lines = 128-byte transposed segments (8 lines / 64 bytes measured 1.6x slower)");
    static_assert(dit_leaf_count(R) % LCHL == 0, "whole chunks");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* tw = reinterpret_cast<float2*>(smem_raw);
    float2* pl = tw + N;
    float2* tile = pl + N;                            // LINES * CS
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += NT) { tw[i] = job.tw[i]; pl[i] = job.pl[i]; }
    __syncthreads();
    const int grp = tid / R, ln = tid % R;
    const int q = tid % TPS, r0 = tid / TPS;
    const int l64 = tid & 63;
    float2* myrow = tile + grp * CS;
    // add-tid exchange: the groups of a wave share the scratch that starts at the first of their tile rows
    const float* wscr = reinterpret_cast<const float*>(tile + (grp - grp % (64 / R)) * CS);
    const unsigned wscr_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(tile + (grp - grp % (64 / R)) * CS));
    const int lblocks = job.n_lines / LINES;
    const int PC = job.pchunk;
    const int pchunks = (job.n_images + PC - 1) / PC;
    const int n_items = lblocks * pchunks;
    const bool pre_a = (MSL_ABL2 & 8) ? false : FL >= 0 ? (FL & P2_PRE_A) != 0 : (job.flags & P2_PRE_A) != 0;
    const bool post_a = (MSL_ABL2 & 8) ? false : FL >= 0 ? (FL & P2_POST_A) != 0 : (job.flags & P2_POST_A) != 0;
    // work item = (line block lb, probe chunk pc), item = lb * pchunks + pc; the cursor (item, lb, pc, k) advances
    // incrementally -- a division per iteration costs ~0.3 us of scalar work on the critical path
    const int step_lb = (int)gridDim.x / pchunks, step_pc = (int)gridDim.x % pchunks;
    // input line of this thread's tile row in line block lbb
    auto line_of = [&](int lbb) {
        if constexpr (OUT_P) {
            const int sh = job.perm_shift;                              // blocks of 16 output elements per 2 R' chunk: R' / 8
            return (((lbb >> sh) * 2 + (grp & 1)) << (sh + 3)) + 8 * (lbb & ((1 << sh) - 1)) + (grp >> 1);
        } else {
            return lbb * LINES + grp;
        }
    };
    auto line_ptr = [&](int lbb, int pcc, int kk) {
        return job.in + (long long)(pcc * PC + kk) * job.in_image_stride + (long long)line_of(lbb) * job.in_pitch;
    };
    // registers [LO, HI) of the next line: 8-byte loads of elements j R + ln, or (interleaved input) 16-byte loads of the pairs
    auto load_regs = [&](float2 (&dst)[R], const float2* r, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        if constexpr (IN_P) {
            static_assert(LO % 2 == 0 && HI % 2 == 0, "register pairs");
#pragma unroll
            for (int jp = LO / 2; jp < HI / 2; ++jp) {
                const msl_f4v t = __builtin_nontemporal_load(reinterpret_cast<const msl_f4v*>(r + (2 * R * jp + 2 * ln)));
                dst[2 * jp] = make_float2(t.x, t.y); dst[2 * jp + 1] = make_float2(t.z, t.w);
            }
        } else {
#pragma unroll
            for (int j = LO; j < HI; ++j) dst[j] = ld_stream(r + (j * R + ln));
        }
    };
    int item = blockIdx.x;
    int lb = item / pchunks, pc = item - lb * pchunks, k = 0;
    float2 vn[R];
    if (item < n_items) load_regs(vn, line_ptr(lb, pc, 0), MSL_IC(0), MSL_IC(R));
    // The wait for the prefetched line.  vmcnt counts loads and stores together, in issue order, and inside the loop the 16 stores
    // of an iteration are YOUNGER than the loads of the next line: the wave may start on that line with its stores still in
    // flight (s_waitcnt vmcnt(16 + ...)).  The compiler's counter model merges the loop's back edge with the loop entry, where
    // the first line's loads are the youngest operations, takes the stricter of the two, and makes every iteration wait for
    // vmcnt(7) ... vmcnt(0) -- for its own stores to be acknowledged.  With nothing in flight at the loop entry the back edge
    // alone sets the counts (tests/test_abi_and_host.py checks them in the ISA); same-box A/B +0.5 %.
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), the other counters untouched
    float2 tv[R];
#ifdef MSL_CLOCK
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    while (item < n_items) {
        // The waves of a workgroup reach every lane <-> register exchange together and the LDS serves eight exchanges at once.
        // Waves on different SIMDs start an iteration MSL_STAGGER x 64 cycles apart (waves w and w + 4 share a SIMD and stay
        // together: delaying one against its partner only costs the delay): 3.7 % fewer cycles per pass, 1.3-1.5 % less time
        // (the clock the chip holds under its power limit falls as the work per cycle rises).
        if constexpr (MSL_STAGGER > 0)
            for (int i = __builtin_amdgcn_readfirstlane(tid >> 6) & 3; i > 0; --i) __builtin_amdgcn_s_sleep(MSL_STAGGER);
        float2 v[R];
        const int p = pc * PC + k;
        const int cur_lb = lb;
        if (k == 0) {
            const float2* trow = job.trans + frame_off(job, pc * PC) + (long long)line_of(lb) * N;
#pragma unroll
            for (int j = 0; j < R; ++j) tv[j] = ld_stream(trow + j * R + ln);       // read once per launch too (+0.4 %)
        }
        int nitem = item, nlb = lb, npc = pc, nk = k + 1;
        if (nk >= min(PC, job.n_images - pc * PC)) {
            nk = 0; nitem = item + (int)gridDim.x; nlb = lb + step_lb; npc = pc + step_pc;
            if (npc >= pchunks) { npc -= pchunks; ++nlb; }
        }
        // Prefetch of the next line: one sixteenth (R/16 registers, whole 16-byte pairs) after the leaf level and after the upper
        // levels of each of the eight register transforms, unconditional (past the last item the loads re-read the current line),
        // one address per iteration.  The memory pipeline accepts a wave's loads at the rate HBM returns data (about 32 KB in
        // flight per CU) and a burst blocks the in-order wave until it is accepted: per 1024^2 pass 322 us with all loads at the top
        // of the iteration, 306 us all after the third transform, 294 us in halves, 287 us in quarters (rounds 1-3), and with
        // the staggered waves the sixteenths are another 3 % against the quarters.
        const bool more = nitem < n_items;
        const float2* nptr = line_ptr(more ? nlb : lb, more ? npc : pc, more ? nk : k);
        auto pfx = [&](auto i_c) {
            constexpr int I = decltype(i_c)::value;
            constexpr int LO = (R * I / 16) & ~1, HI = (R * (I + 1) / 16) & ~1;
            if constexpr (HI > LO && !(MSL_ABL2 & 1)) {
                __builtin_amdgcn_sched_barrier(0);
                load_regs(vn, nptr, MSL_IC(LO), MSL_IC(HI));
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // one register transform with the table `TAB` folded into its leaf level, two prefetch slots
#define MSL_HALF_TAB(INV, WM, TAB, I0) do { dit_leaf_chunks<R, INV, WM, R, LCHL, 0>(v, v, TAB, ln); pfx(MSL_IC(I0)); \
                                            dit_upper<R, INV, FN>(v); pin_all(v); pfx(MSL_IC((I0) + 1)); } while (0)
        if (pre_a) {
            dit_leaves_plain<R, false, 0>(vn, v); pin_all(v);       // reads the prefetched line; vn is free from here on
            pfx(MSL_IC(0));
            dit_upper<R, false, FN>(v); pin_all(v);
            pfx(MSL_IC(1));
            exchange_addtid<R>(v, wscr, wscr_lds, ln, l64);
            MSL_HALF_TAB(false, 1, tw, 2);
            MSL_HALF_TAB(true, 1, pl, 4);
            exchange_addtid<R>(v, wscr, wscr_lds, ln, l64);
            MSL_HALF_TAB(true, 2, tw, 6);
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = vn[j];
            pin_all(v);
            pfx(MSL_IC(0)); pfx(MSL_IC(1)); pfx(MSL_IC(2)); pfx(MSL_IC(3)); pfx(MSL_IC(4)); pfx(MSL_IC(5)); pfx(MSL_IC(6)); pfx(MSL_IC(7));
        }
        if (post_a) {
            dit_leaves_regs<R, false, 1, 0>(v, v, tv); pin_all(v);
            pfx(MSL_IC(8));
            dit_upper<R, false, FN>(v); pin_all(v);
            pfx(MSL_IC(9));
            exchange_addtid<R>(v, wscr, wscr_lds, ln, l64);
            MSL_HALF_TAB(false, 1, tw, 10);
            MSL_HALF_TAB(true, 1, pl, 12);
            exchange_addtid<R>(v, wscr, wscr_lds, ln, l64);
            MSL_HALF_TAB(true, 2, tw, 14);
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = cmulf(v[j], tv[j]);
            pfx(MSL_IC(8)); pfx(MSL_IC(9)); pfx(MSL_IC(10)); pfx(MSL_IC(11)); pfx(MSL_IC(12)); pfx(MSL_IC(13)); pfx(MSL_IC(14)); pfx(MSL_IC(15));
        }
#undef MSL_HALF_TAB
        if constexpr (MSL_ABL2 & 32) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < R; ++j) acc += v[j].x + v[j].y;
            if (acc == 1.2345e-30f) job.out[tid] = make_float2(acc, acc);        // keeps the transforms alive
            item = nitem; lb = nlb; pc = npc; k = nk;
            continue;
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < R; ++j) myrow[j * R + ln] = v[j];
        lds_barrier();
        // uniform 64-bit base + per-thread 32-bit element offset, re-derived every iteration (the asm keeps the
        // compiler from hoisting 16 loop-invariant 64-bit addresses into registers for the whole kernel)
        float2* dst = job.out + (long long)p * job.out_image_stride + cur_lb * LINES;
        int off0 = 2 * q + r0 * job.out_pitch;
        asm volatile("" : "+v"(off0));
        const int ostep = POS_PER_IT * job.out_pitch;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pos = r0 + POS_PER_IT * i;
            const float2 a = tile[(2 * q) * CS + pos], b = tile[(2 * q + 1) * CS + pos];
#if MSL_ABL2 & 2
            if (a.x == 1.2345e-30f)                 // never true: the LDS reads stay, the store goes
#endif
            st_stream(dst + (off0 + i * ostep), a.x, a.y, b.x, b.y);
        }
        lds_barrier();
        item = nitem; lb = nlb; pc = npc; k = nk;
    }
#ifdef MSL_CLOCK
    if (tid == 0 && job.clk) {
        job.clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
        job.clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#endif
}

// LDS bytes of a workgroup of rowT_pass_kernel<R, 16, ...>
constexpr size_t rowT_lds_bytes(int R) { return ((size_t)2 * R * R + (size_t)16 * ((R * R + 33) / 32 * 32 + 2)) * 8; }

// launch of the instantiation for (R, job.flags) on `stream` (slice_pass.hip); false: R is not 16 or 32
bool rowT_launch(int R, const RowTJob& job, int grid, size_t lds_limit, hipStream_t stream);
// one workgroup exchanges known data through the add-tid scratch and checks it (msl_create); 0 = ok
int rowT_selftest(hipStream_t stream);

}  // namespace msl
