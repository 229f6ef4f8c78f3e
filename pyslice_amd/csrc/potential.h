// Kirkland projected-potential rasteriser (reciprocal-space form) and probe builder for gfx950.
//
// Reference algorithm (src/multislice/potentials.py:253-342):
//   R_s[kx,ky] = sum_Z f_Z(kx^2+ky^2) * sum_{a in Z, slice s} exp(-2 pi i (kx x_a + ky y_a))
//   V_s = Re ifft2(R_s) / (dx^2 dy^2)
// The double sum is a complex outer-product accumulation per (slice, species).  Phases are
// formed as integer-frequency x fractional-coordinate products reduced in float64 (naive fp32
// 2 pi k x loses ~2e-4 rad at k=5/A, x=100 A), then evaluated with sincospi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernel_util.h"

namespace msl {

__device__ __forceinline__ int signed_freq(int m, int n) { return (m < (n + 1) / 2) ? m : m - n; }

// f_Z(q^2) on the (nx,ny) grid for each species present; reference potentials.py:79-96.
// abcd: (103,3,4) doubles; species: n_species atomic numbers.
__global__ void formfactor_kernel(float* __restrict__ ff, const double* __restrict__ abcd,
                                  const int* __restrict__ species, int n_species, int nx, int ny,
                                  double inv_lx, double inv_ly) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long npix = (long long)nx * ny;
    if (i >= npix * n_species) return;
    int sp = (int)(i / npix);
    long long p = i - (long long)sp * npix;
    int mx = (int)(p / ny), my = (int)(p - (long long)mx * ny);
    double kx = signed_freq(mx, nx) * inv_lx, ky = signed_freq(my, ny) * inv_ly;
    double q2 = kx * kx + ky * ky;
    const double* t = abcd + (size_t)(species[sp] - 1) * 12;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s1 += t[j * 4 + 0] / (q2 + t[j * 4 + 1]);
        s2 += t[j * 4 + 2] * exp(-t[j * 4 + 3] * q2);
    }
    ff[i] = (float)(s1 + s2);
}

// Per atom of a group of G frames (a = frame * n + i; the species Z[i] are the same in every frame): species index and slice
// index -> sort key (frame * nz + slice) * n_species + species (or -1 when the atom belongs to no slice / unknown species), and
// fractional in-plane coordinates.  Slice rule: potentials.py:302-307.
__global__ void atom_prep_kernel(const double* __restrict__ pos, const int* __restrict__ Z, long long n, int n_frames,
                                 const int* __restrict__ z_to_species, const double* __restrict__ lo,
                                 const double* __restrict__ hi, int nz, int n_species, int ax1, int ax2, int axs,
                                 double inv_l1, double inv_l2, int* __restrict__ key, double* __restrict__ u1,
                                 double* __restrict__ u2, int* __restrict__ counts) {
    long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n * n_frames) return;
    const int frame = (int)(a / n);
    const long long i = a - (long long)frame * n;
    double zc = pos[a * 3 + axs];
    int zz = Z[i];
    int sp = (zz >= 1 && zz <= 103) ? z_to_species[zz] : -1;
    // binary search for the last slice with lo[s] <= zc, then test the reference's [lo,hi) masks
    int s = -1;
    if (zc >= lo[0]) {
        int l = 0, r = nz - 1;
        while (l < r) {
            int m = (l + r + 1) >> 1;
            if (lo[m] <= zc) l = m; else r = m - 1;
        }
        for (int c = l; c >= 0 && c >= l - 1; --c) {
            if (zc >= lo[c] && zc < hi[c]) { s = c; break; }
        }
    }
    int k = (s >= 0 && sp >= 0) ? (frame * nz + s) * n_species + sp : -1;
    key[a] = k;
    u1[a] = pos[a * 3 + ax1] * inv_l1;
    u2[a] = pos[a * 3 + ax2] * inv_l2;
    if (k >= 0) atomicAdd(&counts[k], 1);
}

// Exclusive scan of the bin counts (frames x slices x species of a group: up to a few thousand); one workgroup of 1024 threads,
// every thread a contiguous share, the shares' totals scanned through the LDS.  `align` (1 or SF_ALIGN = 8): every bin is rounded up
// to a multiple of that many rows -- the phase tables get zero rows there (order = -1) -- for structure_factor_stream_kernel, which
// walks whole half-trips of 8 atoms without masks.  start[nkeys] = rows of the sorted tables, padding included.
#define SF_ALIGN 16         // rows per trip of structure_factor_stream_bf16_kernel (the f32 kernel walks half-trips of 8)
__global__ void __launch_bounds__(1024) bin_scan_kernel(const int* __restrict__ counts, int* __restrict__ start, int nkeys, int align) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nkeys + 1023) / 1024;
    const int i0 = t * per, i1 = min(nkeys, i0 + per);
    int acc = 0;
    for (int i = i0; i < i1; ++i) acc += (counts[i] + align - 1) / align * align;
    part[t] = acc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {            // inclusive Hillis-Steele scan of the 1024 share totals
        const int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - acc;                               // exclusive prefix of this share
    for (int i = i0; i < i1; ++i) { start[i] = run; run += (counts[i] + align - 1) / align * align; }
    if (t == 1023) start[nkeys] = part[1023];
}

// Stable compaction: one 1024-thread workgroup per key collects its atoms in original order (deterministic sums).
// Every wave scans a contiguous share of the key's FRAME (n atoms) twice: once to count its matches, then -- after an exclusive
// prefix over the 16 waves -- to write them.
__global__ void __launch_bounds__(1024) bin_fill_kernel(const int* __restrict__ key, long long n, int keys_per_frame,
                                                        const int* __restrict__ start, int* __restrict__ order) {
    __shared__ int wave_count[16];
    const int mykey = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = start[mykey];
    if (start[mykey + 1] == base) return;                       // uniform for the workgroup
    const long long a_first = (long long)(mykey / keys_per_frame) * n;
    const long long per = ((n + 16 * 64 - 1) / (16 * 64)) * 64;   // atoms per wave, a multiple of 64
    const long long lo = a_first + per * wave, hi = (per * (wave + 1) < n) ? lo + per : a_first + n;
    int cnt = 0;
    for (long long a0 = lo; a0 < hi; a0 += 64) {
        const long long a = a0 + lane;
        cnt += __popcll(__ballot((a < hi) && (key[a] == mykey)));
    }
    if (lane == 0) wave_count[wave] = cnt;
    __syncthreads();
    int pos = base;
    for (int w = 0; w < wave; ++w) pos += wave_count[w];
    if (wave == 0) {                                            // padding rows of the bin (at most SF_ALIGN - 1)
        int total = 0;
        for (int w = 0; w < 16; ++w) total += wave_count[w];
        if (base + total + lane < start[mykey + 1]) order[base + total + lane] = -1;
    }
    for (long long a0 = lo; a0 < hi; a0 += 64) {
        const long long a = a0 + lane;
        const bool f = (a < hi) && (key[a] == mykey);
        const unsigned long long m = __ballot(f);
        if (f) order[pos + __popcll(m & ((1ull << lane) - 1ull))] = (int)a;
        pos += __popcll(m);
    }
}

// table[i][m] = exp(-2 pi i * f(m) * u[order[i]]),  f = signed FFT frequency index
// (n_sorted is read on the device -- the number of atoms that fell into a slice -- so that the host never waits for it)
// Only the first n_cols columns of every row of n are filled (the quadrant kernel reads m <= n/2).  The reduction t - rint(t)
// is odd in t, so the full table is exactly conjugate-symmetric: table[a][n-m] == conj(table[a][m]).
__global__ void phase_table_kernel(float2* __restrict__ table, const double* __restrict__ u,
                                   const int* __restrict__ order, const int* __restrict__ n_sorted_ptr, int n, int n_cols, int pitch) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)(*n_sorted_ptr) * n_cols;
    if (i >= total) return;
    int a, m;
    if (total < 0xffffffffLL) {                         // (uniform) 32-bit division: the 64-bit one was most of this kernel's instructions
        const unsigned iu = (unsigned)i;
        a = (int)(iu / (unsigned)n_cols); m = (int)(iu - (unsigned)a * (unsigned)n_cols);
    } else {
        a = (int)(i / n_cols); m = (int)(i - (long long)a * n_cols);
    }
    const int src = order[a];
    if (src < 0) { table[(long long)a * pitch + m] = make_float2(0.f, 0.f); return; }       // padding row of a bin
    double t = (double)signed_freq(m, n) * u[src];
    t -= rint(t);
    float sn, cs;
    sincospif((float)(-2.0 * t), &sn, &cs);
    table[(long long)a * pitch + m] = make_float2(cs, sn);
}

// R[s][kx][ky] = sum_species ff[sp][kx][ky] * sum_{atoms of (s,sp)} ex[a][kx] * ey[a][ky]          (potentials.py:323-330)
// S = sum_a ex_a (x) ey_a is a complex GEMM with a short inner dimension (atoms of one species in one slice) -- the one
// GEMM-shaped step of the path: gfx950's exact-f32 MFMA (v_mfma_f32_32x32x2_f32), one wave per 32 x 32 tile, two atoms per
// instruction (lanes 0-31 / 32-63).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The four real products an atom contributes, with x = ex[a][mx] = (c_x, -s_x), y = ey[a][my] = (c_y, -s_y):
//     A = x.x y.x,  B = x.y y.y,  C = x.x y.y,  D = x.y y.x
// give the bin and its three mirror images:
//     S[ mx,  my] = (A - B,  C + D)      S[-mx,  my] = (A + B,  C - D)
//     S[ mx, -my] = (A + B, -C + D)      S[-mx, -my] = (A - B, -C - D)
// (the phase table is exactly conjugate-symmetric, ex[a][n-m] = conj ex[a][m], and f_Z depends on mx^2, my^2 only).  Bins on the
// axes (m = 0) and on the Nyquist lines (2m = n) are their own mirror image.
// write_mx == 0: the inverse transform takes rows 0 .. nx/2 only and rebuilds row nx - mx as the conjugate of row mx, which is
// exact for every bin but the Nyquist column, where the grid's one frequency -ny/2 serves both signs and R[-mx] is NOT conj
// R[mx].  Re(ifft2) keeps the Hermitian part, (R[mx, ny/2] + conj R[-mx, ny/2]) / 2 = (A, D): stored there in that mode.
__device__ __forceinline__ void store_quad_bin(float2* __restrict__ out, int nx, int ny, int mx, int my, float A, float B, float C,
                                               float D, int write_mx) {
    const bool mir_x = write_mx && mx > 0 && 2 * mx != nx;
    const bool mir_y = my > 0 && 2 * my != ny;
    const float dre = A - B, sre = A + B, sim = C + D, dim = C - D;
    const bool herm_col = !write_mx && 2 * my == ny && 2 * mx != nx;
    out[(size_t)mx * ny + my] = herm_col ? make_float2(A, D) : make_float2(dre, sim);
    if (mir_x) out[(size_t)(nx - mx) * ny + my] = make_float2(sre, dim);
    if (mir_y) out[(size_t)mx * ny + (ny - my)] = make_float2(sre, -dim);
    if (mir_x && mir_y) out[(size_t)(nx - mx) * ny + (ny - my)] = make_float2(dre, -sim);
}

// Quadrant kernel: a wave accumulates A, B, C, D separately -- the same four MFMAs per atom pair as the complex product -- over a
// 32 x 32 tile of the non-negative frequencies only and writes up to four bins per accumulator element: a quarter of the
// arithmetic of the full grid, any nx, ny (loads clamped, stores guarded).
// Grid: one dimension, n_slices (= frames of the group x nz) x wg_per_slice workgroups of four tiles, dealt so that all
// workgroups of a slice run on ONE XCD (workgroups are dealt round-robin over the 8 XCDs: blockIdx % 8 share an L2): a slice's
// phase-table rows are then fetched into one L2 instead of all eight (round 2: 5.3 x the algorithmic traffic at 1024^2, 11 x
// at 2048^2).  tiles_x x tiles_y tiles cover the frequencies [0, 32 tiles); a Nyquist row / column left over by a power-of-two
// grid (n/2 + 1 = 32 k + 1) goes to structure_factor_edge_kernel instead of a 33rd tile row of which 1/32 is used.
__global__ void __launch_bounds__(256) structure_factor_quad_kernel(float2* __restrict__ recip,
                                                                    const float2* __restrict__ ex,
                                                                    const float2* __restrict__ ey,
                                                                    const float* __restrict__ ff,
                                                                    const int* __restrict__ start, int n_species,
                                                                    int nx, int ny, int tiles_y, int n_tiles, int n_rows,
                                                                    int write_mx, int px_pitch, int py_pitch, int n_slices,
                                                                    int wg_per_slice) {
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int s = (jj / wg_per_slice) * 8 + xcd;
    if (s >= n_slices) return;
    const int lane = threadIdx.x & 63;
    const int tile = (jj % wg_per_slice) * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int hx = nx / 2, hy = ny / 2;                     // largest non-negative frequency index of each axis
    const int kx0 = (tile / tiles_y) * 32, ky0 = (tile % tiles_y) * 32;
    const int i = lane & 31, kk = lane >> 5;
    const int lx = min(kx0 + i, hx), ly = min(ky0 + i, hy); // surplus rows / columns of the last tiles recompute the last valid one
    f32x16 tA = {0}, tB = {0}, tC = {0}, tD = {0};
    for (int sp = 0; sp < n_species; ++sp) {
        const int a0 = start[s * n_species + sp], a1 = start[s * n_species + sp + 1];
        if (a0 == a1) continue;
        f32x16 A = {0}, B = {0}, C = {0}, D = {0};
        // 8 atoms (4 MFMA k-steps) per half trip, two register sets in turn: the loads of one set fly while the sixteen MFMAs of
        // the other run.  Loads are unconditional (rows clamped to the table; rows past the species' last atom are zeroed in
        // registers just before use) so that no branch and no register copy ties a wait to the set in flight.
        auto load4 = [&](int a, float2 (&x)[4], float2 (&y)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = min(a + 2 * u, n_rows - 1);
                x[u] = ex[(size_t)r * px_pitch + lx];
                y[u] = ey[(size_t)r * py_pitch + ly];
            }
        };
        auto mfma16 = [&](int a, float2 (&x)[4], float2 (&y)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (a + 2 * u >= a1) { x[u] = make_float2(0.f, 0.f); y[u] = make_float2(0.f, 0.f); }
                A = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u].x, y[u].x, A, 0, 0, 0);
                B = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u].y, y[u].y, B, 0, 0, 0);
                C = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u].x, y[u].y, C, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u].y, y[u].x, D, 0, 0, 0);
            }
        };
        float2 xa[4], ya[4], xb[4], yb[4];
        load4(a0 + kk, xa, ya);
        for (int ab = a0; ab < a1; ab += 16) {                 // uniform trip count
            load4(ab + 8 + kk, xb, yb);
            mfma16(ab + kk, xa, ya);
            load4(ab + 16 + kk, xa, ya);
            if (ab + 8 < a1) mfma16(ab + 8 + kk, xb, yb);
        }
        const float* f = ff + (size_t)sp * nx * ny;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;      // C/D layout of the 32x32 MFMA
            const float w = f[(size_t)min(kx0 + row, hx) * ny + ly];
            tA[r] = fmaf(w, A[r], tA[r]); tB[r] = fmaf(w, B[r], tB[r]);
            tC[r] = fmaf(w, C[r], tC[r]); tD[r] = fmaf(w, D[r], tD[r]);
        }
    }
    float2* out = recip + (size_t)s * nx * ny;
    const int my = ky0 + i;
    if (my > hy) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mx = kx0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (mx > hx) continue;
        store_quad_bin(out, nx, ny, mx, my, tA[r], tB[r], tC[r], tD[r], write_mx);
    }
}

// The same arithmetic as a persistent kernel with ONE wave per SIMD and a continuous matrix-instruction stream, for slices with
// many atoms.  Counters of the kernel above (tools/sf_mfma_counters.sh): the matrix pipes are busy 56 % of its time at 512^2, 69 %
// at 1024^2 -- every (slice, species) is a handful of 16-atom trips between a latency-exposed first operand load and a
// latency-exposed weight load.  Here a wave walks ALL rows of a slice in half-trips of 8 (the sorted tables keep a slice's species
// contiguous; with `align` = 8 every bin is padded with zero rows to a multiple of 8: no masks, and a species boundary is a
// half-trip boundary) with its operand loads three half-trips ahead in four register sets; at a boundary the sums are flushed into
// the totals with the species' weights f_Z, which were fetched while it accumulated.  Work item q = (slice, group of 4 tiles),
// dealt so that the workgroups of an XCD (blockIdx % 8) walk the slices s = xcd + 8 m in order: a few slices in flight per L2.
// Measured against the tiled kernel (same box, potential per frame): 1024^2 x 200 slices (267 rows per bin) 4.42 -> 4.33 ms,
// 2048^2 x 50 (1 058 per bin) 11.85 -> 10.88 ms, 512^2 x 100 (66 per bin, +9 % padding) 0.312 -> 0.316 ms: used from 128 rows per
// bin on.  (A variant without running totals -- species flushed straight into the bins, read-modify-write, 2 waves per SIMD --
// spills: five inlined copies of the flush.)
__global__ void __launch_bounds__(256, 1) structure_factor_stream_kernel(float2* __restrict__ recip, const float2* __restrict__ ex,
                                                                      const float2* __restrict__ ey, const float* __restrict__ ff,
                                                                      const int* __restrict__ start, int n_species, int nx, int ny,
                                                                      int tiles_y, int n_tiles, int n_rows, int write_mx, int px_pitch,
                                                                      int py_pitch, int n_slices) {
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wg_x = gridDim.x >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hx = nx / 2, hy = ny / 2;
    const int i = lane & 31, kk = lane >> 5;
    const int G = (n_tiles + 3) / 4, Mx = (n_slices + 7) / 8;
    for (int q = jw; q < Mx * G; q += wg_x) {
        const int m = q / G, g = q - m * G;
        const int s = xcd + 8 * m, tile = 4 * g + wave;
        if (s >= n_slices || tile >= n_tiles) continue;
        const int kx0 = (tile / tiles_y) * 32, ky0 = (tile % tiles_y) * 32;
        const int lx = min(kx0 + i, hx), ly = min(ky0 + i, hy);
        const float2* exl = ex + lx;
        const float2* eyl = ey + ly;
        const int* st = start + s * n_species;
        const int S0 = st[0], S1 = st[n_species];
        const int H = (S1 - S0) >> 3;                   // whole half-trips: every bin is padded to a multiple of 8 rows
        f32x16 tA = {0}, tB = {0}, tC = {0}, tD = {0};
        f32x16 A = {0}, B = {0}, C = {0}, D = {0};
        // species in turn (empty ones skipped): sp accumulates until row seg1; spn is the next one with rows
        int sp = -1, seg1 = S0, spn = -1, segn1 = S0;
        auto next_of = [&](int& k, int& e) { const int from = e; do { ++k; if (k >= n_species) return; e = st[k + 1]; } while (e <= from); };
        float wc[16], wn[16];               // weights f_Z of the current species, and of the one after it (in flight)
        auto load_w = [&](int spc, float (&w)[16]) {
            const float* f = ff + (size_t)spc * nx * ny;
#pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = f[(size_t)min(kx0 + (r & 3) + 8 * (r >> 2) + 4 * kk, hx) * ny + ly];
        };
#pragma unroll
        for (int r = 0; r < 16; ++r) { wc[r] = 0.f; wn[r] = 0.f; }
        next_of(sp, seg1);
        spn = sp; segn1 = seg1;
        if (sp < n_species) { load_w(sp, wc); next_of(spn, segn1); if (spn < n_species) load_w(spn, wn); }
        auto flush = [&]() {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tA[r] = fmaf(wc[r], A[r], tA[r]); tB[r] = fmaf(wc[r], B[r], tB[r]);
                tC[r] = fmaf(wc[r], C[r], tC[r]); tD[r] = fmaf(wc[r], D[r], tD[r]);
                A[r] = 0.f; B[r] = 0.f; C[r] = 0.f; D[r] = 0.f;
                wc[r] = wn[r];
            }
            sp = spn; seg1 = segn1;
            if (sp < n_species) { next_of(spn, segn1); if (spn < n_species) load_w(spn, wn); }
        };
        auto load8 = [&](int h, float2 (&x)[4], float2 (&y)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = min(S0 + 8 * h + kk + 2 * u, n_rows - 1);
                x[u] = exl[(size_t)r * px_pitch];
                y[u] = eyl[(size_t)r * py_pitch];
            }
        };
        // (the accumulators are pinned to the accumulation registers by the asm constraints: with the builtin the register
        // allocator moved all 64 of them between the two files at every branch of this loop)
        auto mma = [&](f32x16& acc, float a, float b) { asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); };
        auto step = [&](int h, const float2 (&x)[4], const float2 (&y)[4]) {
            if (S0 + 8 * h == seg1) flush();                // the species ended with the previous half-trip
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mma(A, x[u].x, y[u].x);
                mma(B, x[u].y, y[u].y);
                mma(C, x[u].x, y[u].y);
                mma(D, x[u].y, y[u].x);
            }
        };
        float2 x0[4], y0[4], x1[4], y1[4], x2[4], y2[4], x3[4], y3[4];
        if (H > 0) { load8(0, x0, y0); load8(1, x1, y1); load8(2, x2, y2); }
        for (int h = 0; h < H; h += 4) {
            load8(h + 3, x3, y3); step(h, x0, y0);
            if (h + 1 < H) { load8(h + 4, x0, y0); step(h + 1, x1, y1); }
            if (h + 2 < H) { load8(h + 5, x1, y1); step(h + 2, x2, y2); }
            if (h + 3 < H) { load8(h + 6, x2, y2); step(h + 3, x3, y3); }
        }
        if (H > 0) flush();                                 // the slice's last species
        float2* out = recip + (size_t)s * nx * ny;
        const int my = ky0 + i;
        if (my > hy) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mx = kx0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (mx > hx) continue;
            store_quad_bin(out, nx, ny, mx, my, tA[r], tB[r], tC[r], tD[r], write_mx);
        }
    }
}

// The streaming kernel on the bf16 matrix instruction at fp32 accuracy (round 4).  The exact-f32 MFMA runs at the vector rate (64
// cycles per 32x32x2 step and SIMD) and the kernel above is bound by it (its waves are issue-stalled on the matrix pipe 65 % of their
// cycles, profiles/r04_c3_summary.json).  v_mfma_f32_32x32x16_bf16 does eight times the multiply-adds in half the cycles; a float
// is split in registers into three bf16 pieces, x = h + m + l (|m| <= 2^-8 |h|, |l| <= 2^-16 |h|), and a product keeps the six terms
// down to 2^-16: hh + hm + mh + hl + lh + mm -- relative error ~2^-24 per product, accumulation in f32 inside the matrix unit as
// before.  6 x 4 = 24 instructions of 32 cycles per 16 atoms instead of 32 of 64 cycles, plus ~180 vector instructions for the
// splits, which issue while the matrix pipe works.  A trip is 16 rows (atoms): lane (i, kk) holds rows 8 kk .. 8 kk + 7 of frequency i
// as the matrix operand; bins are padded to SF_ALIGN = 16 rows.  Everything else -- persistent waves, species flush with the weights
// f_Z, quadrant store -- is the kernel above.
typedef unsigned sf_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sf_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sf_pk_bf16(float a, float b) {
    const sf_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sf_bf16x2));          // v_cvt_pk_bf16_f32 (round to nearest even)
}
// (a, b) -> packed bf16 pairs of the three pieces
__device__ __forceinline__ void sf_split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = sf_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = sf_pk_bf16(ra, rb);
    l = sf_pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}
__global__ void __launch_bounds__(256, 1) structure_factor_stream_bf16_kernel(float2* __restrict__ recip, const float2* __restrict__ ex,
                                                                           const float2* __restrict__ ey, const float* __restrict__ ff,
                                                                           const int* __restrict__ start, int n_species, int nx, int ny,
                                                                           int tiles_y, int n_tiles, int n_rows, int write_mx, int px_pitch,
                                                                           int py_pitch, int n_slices) {
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wg_x = gridDim.x >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hx = nx / 2, hy = ny / 2;
    const int i = lane & 31, kk = lane >> 5;
    const int G = (n_tiles + 3) / 4, Mx = (n_slices + 7) / 8;
    for (int q = jw; q < Mx * G; q += wg_x) {
        const int m = q / G, g = q - m * G;
        const int s = xcd + 8 * m, tile = 4 * g + wave;
        if (s >= n_slices || tile >= n_tiles) continue;
        const int kx0 = (tile / tiles_y) * 32, ky0 = (tile % tiles_y) * 32;
        const int lx = min(kx0 + i, hx), ly = min(ky0 + i, hy);
        const int* st = start + s * n_species;
        const int S0 = __builtin_amdgcn_readfirstlane(st[0]), S1 = __builtin_amdgcn_readfirstlane(st[n_species]);
        const int H = (S1 - S0) >> 4;                   // whole trips of 16 rows: every bin is padded to a multiple of 16
        // the slice's rows of the two phase tables as raw buffers (zero past the table's end: no row clamp, no vector address
        // arithmetic); lane (i, kk) reads rows 8 kk + u of a trip at frequency lx / ly: one lane offset per table, row offsets scalar
        const unsigned rows_left = (unsigned)(n_rows - S0);
        const msl_i4v rx = make_raw_rsrc(ex + (size_t)S0 * px_pitch, (unsigned)((unsigned long long)rows_left * px_pitch * 8ull < 0xfffffff0ull ? (unsigned long long)rows_left * px_pitch * 8ull : 0xfffffff0ull));
        const msl_i4v ry = make_raw_rsrc(ey + (size_t)S0 * py_pitch, (unsigned)((unsigned long long)rows_left * py_pitch * 8ull < 0xfffffff0ull ? (unsigned long long)rows_left * py_pitch * 8ull : 0xfffffff0ull));
        const int vox = (8 * kk * px_pitch + lx) * 8, voy = (8 * kk * py_pitch + ly) * 8;
        f32x16 tA = {0}, tB = {0}, tC = {0}, tD = {0};
        f32x16 A = {0}, B = {0}, C = {0}, D = {0};
        int sp = -1, seg1 = S0, spn = -1, segn1 = S0;
        auto next_of = [&](int& k, int& e) { const int from = e; do { ++k; if (k >= n_species) return; e = st[k + 1]; } while (e <= from); };
        float wc[16], wn[16];
        auto load_w = [&](int spc, float (&w)[16]) {
            const float* f = ff + (size_t)spc * nx * ny;
#pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = f[(size_t)min(kx0 + (r & 3) + 8 * (r >> 2) + 4 * kk, hx) * ny + ly];
        };
#pragma unroll
        for (int r = 0; r < 16; ++r) { wc[r] = 0.f; wn[r] = 0.f; }
        next_of(sp, seg1);
        spn = sp; segn1 = seg1;
        if (sp < n_species) { load_w(sp, wc); next_of(spn, segn1); if (spn < n_species) load_w(spn, wn); }
        auto flush = [&]() {
            // (wait states by hand: the sums below are read by vector moves of accumulators that inline-assembly matrix instructions wrote --
            // the compiler's hazard recognizer does not know them to be such; 19 would do for a 16-pass instruction)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tA[r] = fmaf(wc[r], A[r], tA[r]); tB[r] = fmaf(wc[r], B[r], tB[r]);
                tC[r] = fmaf(wc[r], C[r], tC[r]); tD[r] = fmaf(wc[r], D[r], tD[r]);
                A[r] = 0.f; B[r] = 0.f; C[r] = 0.f; D[r] = 0.f;
                wc[r] = wn[r];
            }
            sp = spn; seg1 = segn1;
            if (sp < n_species) { next_of(spn, segn1); if (spn < n_species) load_w(spn, wn); }
        };
        auto load16 = [&](int h, float2 (&x)[8], float2 (&y)[8]) {
            const int r0 = __builtin_amdgcn_readfirstlane(16 * h);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const msl_f2v a = msl_raw_buffer_load_f2(rx, vox, (r0 + u) * px_pitch * 8, 0);
                const msl_f2v b = msl_raw_buffer_load_f2(ry, voy, (r0 + u) * py_pitch * 8, 0);
                x[u] = make_float2(a.x, a.y); y[u] = make_float2(b.x, b.y);
            }
        };
        auto mma = [&](f32x16& acc, const sf_u32x4& a, const sf_u32x4& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); };
        auto mma6 = [&](f32x16& acc, const sf_u32x4 (&p)[3], const sf_u32x4 (&q)[3]) {      // all six kept terms of one real product sum
            mma(acc, p[0], q[0]); mma(acc, p[0], q[1]); mma(acc, p[1], q[0]);
            mma(acc, p[0], q[2]); mma(acc, p[2], q[0]); mma(acc, p[1], q[1]);
        };
        // one plane (cos or sin along one axis) of a raw trip -> its three bf16 pieces, 8 rows packed in pairs: 44 vector instructions
        auto conv = [&](sf_u32x4 (&o)[3], const float2 (&v)[8], bool imag) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned p0, p1, p2;
                sf_split3(imag ? v[2 * u].y : v[2 * u].x, imag ? v[2 * u + 1].y : v[2 * u + 1].x, p0, p1, p2);
                o[0][u] = p0; o[1][u] = p1; o[2][u] = p2;
            }
        };
        // ONE set of split operands (48 registers) and two raw sets (64): a plane is converted for the next trip as soon as the last
        // matrix instruction that reads its current contents has issued -- cx after A and C, cy after D, sx and sy after B -- so the
        // 176 vector instructions of a trip's splits issue between its matrix instructions (the matrix pipe works meanwhile):
        //     A(cx,cy)  [sy, sx of THIS trip]   C(cx,sy)  [cx of the next]   D(sx,cy)  [cy of the next]   B(sx,sy)
        sf_u32x4 cx[3], sx[3], cy[3], sy[3];
        // (x, y): the raw set of this trip -- its sin planes are converted first, then it is free and takes the loads of trip h + 2,
        // one trip ahead of their first use; (xn, yn): the raw set of the next trip, whose cos planes are converted here
        auto trip = [&](int h, float2 (&x)[8], float2 (&y)[8], const float2 (&xn)[8], const float2 (&yn)[8]) {
            if (S0 + 16 * h == seg1) flush();               // the species ended with the previous trip
            mma6(A, cx, cy); conv(sy, y, true); conv(sx, x, true); __builtin_amdgcn_sched_barrier(0);
            load16(h + 2, x, y); __builtin_amdgcn_sched_barrier(0);
            mma6(C, cx, sy); conv(cx, xn, false); __builtin_amdgcn_sched_barrier(0);
            mma6(D, sx, cy); conv(cy, yn, false); __builtin_amdgcn_sched_barrier(0);
            mma6(B, sx, sy); __builtin_amdgcn_sched_barrier(0);
        };
        float2 x0[8], y0[8], x1[8], y1[8];
        if (H > 0) { load16(0, x0, y0); load16(1, x1, y1); conv(cx, x0, false); conv(cy, y0, false); }
        for (int h = 0; h < H; h += 2) {
            trip(h, x0, y0, x1, y1);
            if (h + 1 < H) trip(h + 1, x1, y1, x0, y0);
        }
        if (H > 0) flush();
        float2* out = recip + (size_t)s * nx * ny;
        const int my = ky0 + i;
        if (my > hy) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mx = kx0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (mx > hx) continue;
            store_quad_bin(out, nx, ny, mx, my, tA[r], tB[r], tC[r], tD[r], write_mx);
        }
    }
}

// ---- two tiles per wave ------------------------------------------------------------------------------------------------------------
// The kernel above is bound by its table reads out of the L2, not by the splits or the matrix pipe: every 32 x 32 tile reads 64 columns
// of every atom of its slice (2048^2: 56 GB per frame at the L2; with tables pre-split into bf16 pieces -- 1.5 x the bytes, no split
// instructions -- it ran 1.45 x slower, profiles/r04_sf_presplit_tables.txt).  Here a wave owns TWO tiles along kx (columns kx0 .. kx0 + 63
// of ex) that share the ey operand planes: 96 table columns per 2 tiles instead of 128, 48 matrix instructions per trip against 264
// split instructions (24 against 176), and the four waves of a workgroup take four neighbouring ky tiles of the same kx pair, so their
// ex reads coincide in the L1.  Species weights are fetched when a species starts (its accumulation hides the latency); the species
// totals of the eight accumulators (128 values per lane) live in the LDS -- 32 KB per wave, touched at the two or three species
// boundaries of a work item only -- because accumulators + totals + operands of two tiles do not fit the vector registers.
// n_pairs = ceil(tiles_x / 2) tiles_y work items per slice; the second tile of the last pair of an odd tile count recomputes clamped
// columns and stores nothing.
__global__ void __launch_bounds__(256, 1) structure_factor_stream_bf16x2_kernel(float2* __restrict__ recip, const float2* __restrict__ ex,
                                                                             const float2* __restrict__ ey, const float* __restrict__ ff,
                                                                             const int* __restrict__ start, int n_species, int nx, int ny,
                                                                             int tiles_y, int n_pairs, int n_rows, int write_mx, int px_pitch,
                                                                             int py_pitch, int n_slices) {
    extern __shared__ float sf_tot[];                               // 4 waves x 128 totals x 64 lanes
    const int xcd = blockIdx.x & 7, jw = blockIdx.x >> 3, wg_x = gridDim.x >> 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hx = nx / 2, hy = ny / 2;
    const int i = lane & 31, kk = lane >> 5;
    const int G = (n_pairs + 3) / 4, Mx = (n_slices + 7) / 8;
    for (int q = jw; q < Mx * G; q += wg_x) {
        const int m = q / G, g = q - m * G;
        const int s = xcd + 8 * m, pair = 4 * g + wave;
        if (s >= n_slices || pair >= n_pairs) continue;
        const int kx0 = (pair / tiles_y) * 64, ky0 = (pair % tiles_y) * 32;
        const int lx0 = min(kx0 + i, hx), lx1 = min(kx0 + 32 + i, hx), ly = min(ky0 + i, hy);
        const int* st = start + s * n_species;
        const int S0 = __builtin_amdgcn_readfirstlane(st[0]), S1 = __builtin_amdgcn_readfirstlane(st[n_species]);
        const int H = (S1 - S0) >> 4;                   // whole trips of 16 rows: every bin is padded to a multiple of 16
        const unsigned rows_left = (unsigned)(n_rows - S0);
        const msl_i4v rx = make_raw_rsrc(ex + (size_t)S0 * px_pitch, (unsigned)((unsigned long long)rows_left * px_pitch * 8ull < 0xfffffff0ull ? (unsigned long long)rows_left * px_pitch * 8ull : 0xfffffff0ull));
        const msl_i4v ry = make_raw_rsrc(ey + (size_t)S0 * py_pitch, (unsigned)((unsigned long long)rows_left * py_pitch * 8ull < 0xfffffff0ull ? (unsigned long long)rows_left * py_pitch * 8ull : 0xfffffff0ull));
        const int vox0 = (8 * kk * px_pitch + lx0) * 8, vox1 = (8 * kk * px_pitch + lx1) * 8, voy = (8 * kk * py_pitch + ly) * 8;
        // (accumulators: every definition is a matrix instruction, so that the eight tuples stay in the accumulation registers --
        // zeroed by a product of zeros; with ordinary assignments the allocator kept them in vector registers and spilled)
        f32x16 A0, B0, C0, D0, A1, B1, C1, D1;
        const sf_u32x4 z4 = {0u, 0u, 0u, 0u};
        // (wait states by hand around it: the compiler does not know that the statement is a matrix instruction, and without them a
        // second work item of a workgroup read NaNs -- the species flush in front of it reads the same registers with vector moves)
        auto zero_acc = [&](f32x16& acc) { asm volatile("s_nop 15\n\ts_nop 15\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %1, 0\n\ts_nop 15\n\ts_nop 15" : "=a"(acc) : "v"(z4)); };
        zero_acc(A0); zero_acc(B0); zero_acc(C0); zero_acc(D0); zero_acc(A1); zero_acc(B1); zero_acc(C1); zero_acc(D1);
        float* tot = sf_tot + (wave * 128) * 64 + lane;            // [plane 0..7][r][lane]: plane = 4 tile + (A, B, C, D)
#pragma unroll
        for (int r = 0; r < 128; ++r) tot[r * 64] = 0.f;
        int sp = -1, seg1 = S0;
        auto next_of = [&](int& k, int& e) { const int from = e; do { ++k; if (k >= n_species) return; e = st[k + 1]; } while (e <= from); };
        float w0[16], w1[16];
        auto load_w = [&](int spc) {
            const float* f = ff + (size_t)spc * nx * ny;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                w0[r] = f[(size_t)min(kx0 + row, hx) * ny + ly];
                w1[r] = f[(size_t)min(kx0 + 32 + row, hx) * ny + ly];
            }
        };
#pragma unroll
        for (int r = 0; r < 16; ++r) { w0[r] = 0.f; w1[r] = 0.f; }
        next_of(sp, seg1);
        if (sp < n_species) load_w(sp);
        auto flush = [&]() {
            // (the sums are read by vector moves the compiler places without knowing that inline assembly wrote them: let the last
            // matrix instructions finish first -- 16 passes of 4 cycles and their write-back)
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* t = tot + r * 64;
                t[0 * 1024] = fmaf(w0[r], A0[r], t[0 * 1024]); t[1 * 1024] = fmaf(w0[r], B0[r], t[1 * 1024]);
                t[2 * 1024] = fmaf(w0[r], C0[r], t[2 * 1024]); t[3 * 1024] = fmaf(w0[r], D0[r], t[3 * 1024]);
                t[4 * 1024] = fmaf(w1[r], A1[r], t[4 * 1024]); t[5 * 1024] = fmaf(w1[r], B1[r], t[5 * 1024]);
                t[6 * 1024] = fmaf(w1[r], C1[r], t[6 * 1024]); t[7 * 1024] = fmaf(w1[r], D1[r], t[7 * 1024]);
            }
            zero_acc(A0); zero_acc(B0); zero_acc(C0); zero_acc(D0); zero_acc(A1); zero_acc(B1); zero_acc(C1); zero_acc(D1);
            next_of(sp, seg1);
            if (sp < n_species) load_w(sp);
        };
        auto load16 = [&](int h, float2 (&xa)[8], float2 (&xb)[8], float2 (&y)[8]) {
            const int r0 = __builtin_amdgcn_readfirstlane(16 * h);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const msl_f2v a = msl_raw_buffer_load_f2(rx, vox0, (r0 + u) * px_pitch * 8, 0);
                const msl_f2v c = msl_raw_buffer_load_f2(rx, vox1, (r0 + u) * px_pitch * 8, 0);
                const msl_f2v b = msl_raw_buffer_load_f2(ry, voy, (r0 + u) * py_pitch * 8, 0);
                xa[u] = make_float2(a.x, a.y); xb[u] = make_float2(c.x, c.y); y[u] = make_float2(b.x, b.y);
            }
        };
        auto mma = [&](f32x16& acc, const sf_u32x4& a, const sf_u32x4& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); };
        auto mma6 = [&](f32x16& acc, const sf_u32x4 (&p)[3], const sf_u32x4 (&qq)[3]) {
            mma(acc, p[0], qq[0]); mma(acc, p[0], qq[1]); mma(acc, p[1], qq[0]);
            mma(acc, p[0], qq[2]); mma(acc, p[2], qq[0]); mma(acc, p[1], qq[1]);
        };
        auto conv = [&](sf_u32x4 (&o)[3], const float2 (&v)[8], bool imag) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned p0, p1, p2;
                sf_split3(imag ? v[2 * u].y : v[2 * u].x, imag ? v[2 * u + 1].y : v[2 * u + 1].x, p0, p1, p2);
                o[0][u] = p0; o[1][u] = p1; o[2][u] = p2;
            }
        };
        // one set of split operands (72 registers) and two raw sets (96): the schedule of the kernel above with both tiles side by side --
        //     A0 A1 (cx, cy)  [sy, sx0, sx1 of THIS trip]   C0 C1 (cx, sy)  [cx0, cx1 of the next]   D0 D1 (sx, cy)  [cy of the next]   B0 B1 (sx, sy)
        sf_u32x4 cx0[3], sx0[3], cx1[3], sx1[3], cy[3], sy[3];
        auto trip = [&](int h, float2 (&xa)[8], float2 (&xb)[8], float2 (&y)[8], const float2 (&xan)[8], const float2 (&xbn)[8], const float2 (&yn)[8]) {
            if (S0 + 16 * h == seg1) flush();               // the species ended with the previous trip
            mma6(A0, cx0, cy); mma6(A1, cx1, cy); conv(sy, y, true); conv(sx0, xa, true); conv(sx1, xb, true); __builtin_amdgcn_sched_barrier(0);
            load16(h + 2, xa, xb, y); __builtin_amdgcn_sched_barrier(0);
            mma6(C0, cx0, sy); mma6(C1, cx1, sy); conv(cx0, xan, false); conv(cx1, xbn, false); __builtin_amdgcn_sched_barrier(0);
            mma6(D0, sx0, cy); mma6(D1, sx1, cy); conv(cy, yn, false); __builtin_amdgcn_sched_barrier(0);
            mma6(B0, sx0, sy); mma6(B1, sx1, sy); __builtin_amdgcn_sched_barrier(0);
        };
        float2 xa0[8], xb0[8], y0[8], xa1[8], xb1[8], y1[8];
        if (H > 0) { load16(0, xa0, xb0, y0); load16(1, xa1, xb1, y1); conv(cx0, xa0, false); conv(cx1, xb0, false); conv(cy, y0, false); }
        for (int h = 0; h < H; h += 2) {
            trip(h, xa0, xb0, y0, xa1, xb1, y1);
            if (h + 1 < H) trip(h + 1, xa1, xb1, y1, xa0, xb0, y0);
        }
        if (H > 0) flush();
        float2* out = recip + (size_t)s * nx * ny;
        const int my = ky0 + i;
        if (my > hy) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mx = kx0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float* t = tot + r * 64;
            if (mx <= hx) store_quad_bin(out, nx, ny, mx, my, t[0 * 1024], t[1 * 1024], t[2 * 1024], t[3 * 1024], write_mx);
            if (mx + 32 <= hx) store_quad_bin(out, nx, ny, mx + 32, my, t[4 * 1024], t[5 * 1024], t[6 * 1024], t[7 * 1024], write_mx);
        }
    }
}

// (The tiled kernel for few atoms per bin keeps the exact-f32 instruction: with 4-5 trips per bin the splits are not hidden and trips of
// 16 rows waste a fifth of them -- a split-bf16 form of it measured 7 % slower on 512^2 / 501^2 / 256^2 single-probe runs.)

// The Nyquist row mx = nx/2 (edge_x) and / or column my = ny/2 (edge_y) of a slice by direct summation -- (nx/2 + ny/2 + 1) bins
// x the slice's atoms: along the row every thread shares the atom's x phase and reads consecutive y phases, along the column the
// other way round, so the table reads are broadcast + coalesced.  grid (ceil(bins / 128), n_slices), 128 threads.
__global__ void __launch_bounds__(128) structure_factor_edge_kernel(float2* __restrict__ recip, const float2* __restrict__ ex,
                                                                    const float2* __restrict__ ey, const float* __restrict__ ff,
                                                                    const int* __restrict__ start, int n_species, int nx, int ny,
                                                                    int edge_x, int edge_y, int write_mx, int px_pitch, int py_pitch) {
    const int s = blockIdx.y;
    const int hx = nx / 2, hy = ny / 2;
    const int n_row = edge_x ? hy + 1 : 0;                  // bins (hx, 0 .. hy)
    const int n_col = edge_y ? (edge_x ? hx : hx + 1) : 0;  // bins (0 .. hx [- 1: the corner belongs to the row], hy)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_row + n_col) return;
    const int mx = i < n_row ? hx : i - n_row, my = i < n_row ? i : hy;
    float tA = 0.f, tB = 0.f, tC = 0.f, tD = 0.f;
    for (int sp = 0; sp < n_species; ++sp) {
        const int a0 = start[s * n_species + sp], a1 = start[s * n_species + sp + 1];
        float A = 0.f, B = 0.f, C = 0.f, D = 0.f;
        int a = a0;
        for (; a + 8 <= a1; a += 8) {                           // eight atoms' loads in flight; summation order unchanged
            float2 x[8], y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { x[u] = ex[(size_t)(a + u) * px_pitch + mx]; y[u] = ey[(size_t)(a + u) * py_pitch + my]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                A = fmaf(x[u].x, y[u].x, A); B = fmaf(x[u].y, y[u].y, B); C = fmaf(x[u].x, y[u].y, C); D = fmaf(x[u].y, y[u].x, D);
            }
        }
        for (; a < a1; ++a) {
            const float2 x = ex[(size_t)a * px_pitch + mx], y = ey[(size_t)a * py_pitch + my];
            A = fmaf(x.x, y.x, A); B = fmaf(x.y, y.y, B); C = fmaf(x.x, y.y, C); D = fmaf(x.y, y.x, D);
        }
        const float w = ff[(size_t)sp * nx * ny + (size_t)mx * ny + my];
        tA = fmaf(w, A, tA); tB = fmaf(w, B, tB); tC = fmaf(w, C, tC); tD = fmaf(w, D, tD);
    }
    store_quad_bin(recip + (size_t)s * nx * ny, nx, ny, mx, my, tA, tB, tC, tD, write_mx);
}

// Reciprocal-space probes: psik[p][mx][my] = mask * exp(2 pi i (fx (hx/nx + px/Lx) + fy (hy/ny + py/Ly)))
// so that ifft2(psik)[p] == create_batched_probes(Probe(...))[p]  (multislice.py:116-124, 216-227).
// Plane wave (mrad == 0): psik = nx*ny at DC only (ones after the normalised inverse FFT).
__global__ void probe_kspace_kernel(float2* __restrict__ psik, const double* __restrict__ xy, int P, int nx, int ny,
                                    int pitch, double inv_lx, double inv_ly, double kfreq_x, double kfreq_y, double radius,
                                    int plane_wave) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long npix = (long long)nx * ny;
    if (i >= npix * P) return;
    int p = (int)(i / npix);
    long long q = i - (long long)p * npix;
    int mx = (int)(q / ny), my = (int)(q - (long long)mx * ny);
    const long long o = ((long long)p * nx + mx) * pitch + my;
    if (plane_wave) {
        psik[o] = (mx == 0 && my == 0) ? make_float2((float)npix, 0.f) : make_float2(0.f, 0.f);
        return;
    }
    int fx = signed_freq(mx, nx), fy = signed_freq(my, ny);
    double kx = fx * kfreq_x, ky = fy * kfreq_y;       // fftfreq value = index * (1/(n*d))
    bool inside = sqrt(kx * kx + ky * ky) < radius;
    if (!inside) { psik[o] = make_float2(0.f, 0.f); return; }
    double t = fx * ((double)(nx / 2) / nx + xy[2 * p] * inv_lx) + fy * ((double)(ny / 2) / ny + xy[2 * p + 1] * inv_ly);
    t -= rint(t);
    float sn, cs;
    sincospif((float)(2.0 * t), &sn, &cs);
    psik[o] = make_float2(cs, sn);
}

// psi0k[p][mx][my] = basek[mx][my] * exp(2 pi i (kx px + ky py))   (multislice.py:221-223)
__global__ void probe_ramp_kernel(float2* __restrict__ out, const float2* __restrict__ basek,
                                  const double* __restrict__ xy, int P, int nx, int ny, int pitch, double inv_lx, double inv_ly) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long npix = (long long)nx * ny;
    if (i >= npix * P) return;
    int p = (int)(i / npix);
    long long q = i - (long long)p * npix;
    int mx = (int)(q / ny), my = (int)(q - (long long)mx * ny);
    const long long o = ((long long)p * nx + mx) * pitch + my;
    double t = signed_freq(mx, nx) * (xy[2 * p] * inv_lx) + signed_freq(my, ny) * (xy[2 * p + 1] * inv_ly);
    t -= rint(t);
    float sn, cs;
    sincospif((float)(2.0 * t), &sn, &cs);
    float2 b = basek[q];
    out[o] = make_float2(b.x * cs - b.y * sn, b.x * sn + b.y * cs);
}

// t = exp(i sigma V) from an uploaded V (nz,nx,ny) float32
__global__ void transmission_kernel(float2* __restrict__ t, const float* __restrict__ V, long long n, float sigma) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    sincosf(sigma * V[i], &sn, &cs);
    t[i] = make_float2(cs, sn);
}

}  // namespace msl
