// Small device helpers shared by the kernel headers: complex products, the LDS ordering points, streaming (non-temporal) accesses.
#pragma once
#include <hip/hip_runtime.h>

namespace msl {

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmulf_conj(float2 a, float2 b) {      // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}

// LDS ordering inside one wave: DS instructions of a wave execute in order, so only the compiler
// must be kept from reordering the accesses of the transpose (LLVM AMDGPU memory model: LDS operations of one
// wavefront are issued and complete in program order -- "ds" instructions return in order and lgkmcnt counts them
// in order).  This holds only while a scratch area is touched by ONE wave: every transform below asserts that an
// R-lane group never straddles a wave (64 % R == 0).  A/B check: build with -DMSL_LDS_ORDER_ONLY=0.
#ifndef MSL_LDS_ORDER_ONLY
#define MSL_LDS_ORDER_ONLY 1
#endif
// Ordering point between LDS phases of ONE wave (write a scratch, read it back transposed, write the next part ...).
// The LDS executes the operations of a wave in issue order, so no wait is needed for correctness -- only the compiler
// must not reorder the accesses: a scheduling barrier and a compiler memory barrier.  (With real wavefront-scope fences
// every phase change costs a full s_waitcnt lgkmcnt(0); MSL_LDS_ORDER_ONLY=0 restores them.)
__device__ __forceinline__ void wave_lds_fence() {
#if MSL_LDS_ORDER_ONLY
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// psi is read once and written once per pass: non-temporal ("nt") accesses mark the lines for early eviction, so they do not
// push t_k, the tables and the other streams out of L2 / Infinity Cache (1024^2 x 64 probes: loads +0.3 %, stores +1.2 %,
// both +2.0 % in a same-box A/B: 1 041 -> 1 020 us per launch of 256 images; 512^2 +2.8 %, 2048^2 and the TACAW time
// transform unchanged, the chirp-z kernel for N <= 512 1.8 % slower: it keeps plain accesses).
typedef float msl_f2v __attribute__((ext_vector_type(2)));
typedef float msl_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float2 ld_stream(const float2* p) {
    const msl_f2v t = __builtin_nontemporal_load(reinterpret_cast<const msl_f2v*>(p));
    return make_float2(t.x, t.y);
}
__device__ __forceinline__ void st_stream(float2* p, float ax, float ay, float bx, float by) {     // two complex values, 16 bytes
    const msl_f4v t = {ax, ay, bx, by};
    __builtin_nontemporal_store(t, reinterpret_cast<msl_f4v*>(p));
}

// Pin values to this point of the program: an empty volatile asm that "modifies" them.  Volatile asm statements, scheduling barriers
// and memory operations keep their program order, but plain arithmetic does not -- instruction selection is free to sink the
// butterflies of one transform below the table reads of the next (the reads then wait in registers: +64 VGPRs and the kernel
// spills).  Pinning a transform's results where it ends confines every value to its own phase; it emits no instruction.
__device__ __forceinline__ void pin(float2& a) { asm volatile("" : "+v"(a.x), "+v"(a.y)); }
template <int N>
__device__ __forceinline__ void pin_all(float2 (&v)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) pin(v[j]);
}

// Raw buffer loads: a descriptor in 4 SGPRs, a byte offset in one more (SALU arithmetic) and ONE 32-bit lane offset register -- no
// 64-bit address pair per load in flight, no vector address arithmetic.  aux 2 = "nt" (the non-temporal hint of ld_stream).
// num_bytes: the buffer unit returns zero for offsets past it (-1: no bound; offsets are 32-bit either way).
typedef int msl_i4v __attribute__((ext_vector_type(4)));
__device__ msl_f2v msl_raw_buffer_load_f2(msl_i4v rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ __forceinline__ msl_i4v make_raw_rsrc(const void* base, unsigned num_bytes = 0xffffffffu) {          // stride 0
    const unsigned long long a = (unsigned long long)base;
    msl_i4v r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)num_bytes);
    r.w = 0x00020000;
    return r;
}

// workgroup barrier that drains LDS traffic only (global loads/stores stay in flight across it)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace msl
