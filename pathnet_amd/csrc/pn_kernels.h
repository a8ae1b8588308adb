// pn_kernels.h -- device helpers shared by the aggregator kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: D[i][j] += sum_{k<2} A[i][k] * B[k][j], exact fp32 FMA chain.
// Lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; accumulator register r
// of lane l is D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- fp32 products on the bf16 matrix pipe ------------------------------------------------------------------
// gfx950's fp32-input MFMA runs at 1/16 of the bf16 rate (64 vs 1024 flop/cycle/SIMD).  An fp32 value is the
// exact sum of three bf16 values (8 significant bits each, round-to-nearest residuals: x = x0 + x1 + x2 with
// |x1| <= 2^-9 |x|, |x2| <= 2^-18 |x|), so  a.b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + O(2^-24 |a||b|):
// six bf16 MFMAs with fp32 accumulation give an fp32-accurate product (the dropped terms are below one fp32 ulp of
// |a||b|; measured 2e-9 of sum|a||b| against 2e-7 for an fp32 FMA chain) at 16/6 of the fp32 MFMA rate.
// Range is bf16's = fp32's: no scaling, no overflow hazard.  (A value that rounds to +-inf in bf16 -- |x| > 3.39e38
// -- yields NaN instead of inf.)
// v_mfma_f32_32x32x16_bf16: lane l supplies A[i = l & 31][k = 8 (l >> 5) .. +7] and B[k = 8 (l >> 5) .. +7][j = l & 31]
// as 8 bf16 in 4 VGPRs; the accumulator layout is that of mfma32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}
// v_cvt_pk_bf16_f32: two fp32 -> two bf16 (round to nearest even), `lo` in bits 0-15
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// (a, b) -> three packed bf16 pairs with a = a0 + a1 + a2 exactly (same for b); the subtractions are exact
__device__ __forceinline__ void split3(float a, float b, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
    p0 = cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = cvt_pk_bf16(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(ra, rb);
}

// ---- fp32 products on the fp16 matrix pipe: two planes, three MFMAs ------------------------------------------------------
// a = a_hi + a_lo with a_hi = fp16(a) (round to nearest, 11 significant bits) and a_lo = fp16(a - a_hi) (the subtraction
// is exact): while a_lo is a normal fp16 number, |a - a_hi - a_lo| <= 2^-24 |a| -- the pair carries fp32's 24 bits.
//   a.b = a_hi b_hi + (a_hi b_lo + a_lo b_hi) + O(2^-24 |a||b|):  THREE fp16 MFMAs with fp32 accumulation per product
// instead of the six of the bf16 three-plane form above, and two planes instead of three in LDS and in the weight stream.
// What fp16 lacks is bf16's range.  Every operand tensor is therefore multiplied by a power of two (exact) that puts its
// largest magnitude into [2^14, 2^15) -- scale_exp() of a maximum the kernels find in device memory or compute for their
// own tile -- and the product of the two scales is divided out of the accumulator.  With the maximum at 2^14 an element
// needs |a| >= 2^-2, i.e. >= 2^-16 of the maximum, for a_lo to stay normal; below that a_lo loses bits and the element's
// absolute error stops shrinking at 2^-25 = 2^-39 of the tensor's maximum (an fp32 FMA chain: 2^-24 of the element).
// gfx950's fp16 MFMA does not flush denormal inputs.  Known-answer tests: tests/test_gpu_seqh.py.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// (a, b) -> packed fp16 pairs hi = {fp16(a), fp16(b)}, lo = {fp16(a - hi.a), fp16(b - hi.b)}   (v_cvt_pk_f16_f32 x 2,
// two v_cvt_f32_f16, one v_pk_add_f32)
__device__ __forceinline__ void split2h(float a, float b, uint32_t &hi, uint32_t &lo) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
    const f32x2 back = __builtin_convertvector(h, f32x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a - back[0], b - back[1]}, f16x2));
}
// e with vmax * 2^e in [2^14, 2^15), clamped to [-100, 100]; 0 for vmax = 0, negative or not finite
__device__ __forceinline__ int scale_exp(float vmax) {
    const int be = (int)((__float_as_uint(vmax) >> 23) & 255u);
    if (be == 255 || !(vmax > 0.0f)) return 0;
    const int e = 14 - (be - 127);          // (a subnormal maximum: be = 0, clamped below)
    return e > 100 ? 100 : e < -100 ? -100 : e;
}
// 2^e as a float: 0 below the normal range, 2^127 above
__device__ __forceinline__ float exp2i(int e) {
    if (e < -126) return 0.0f;
    return __uint_as_float((uint32_t)((e > 127 ? 127 : e) + 127) << 23);
}

// ---- the LSTM's four gate values of one (path step, unit) in 96 bits (fp16 kernels' saved-for-backward layout) --------------
// i, f, o in [0, 1] and g in [-1, 1] as 24-bit fixed point: steps of 2^-24 / 2^-23, i.e. the ABSOLUTE precision fp32 itself
// has for values near 1.  The BPTT multiplies them into gradients whose own fp32 rounding is of that size; what is lost is
// relative precision of gates below ~2^-8, whose products i (1 - i), ... are then small in absolute terms.  Three dwords
// instead of four per element: -0.11 GB each way at the headline shape.  (The forward's own recurrence uses the unrounded
// values; NaN packs as a finite value -- the forward's h and the loss still carry it.)
__device__ __forceinline__ uint3 pack_gates(float i, float f, float g, float o) {
    const uint32_t qi = (uint32_t)fminf(fmaf(i, 16777216.0f, 0.5f), 16777215.0f);
    const uint32_t qf = (uint32_t)fminf(fmaf(f, 16777216.0f, 0.5f), 16777215.0f);
    const uint32_t qo = (uint32_t)fminf(fmaf(o, 16777216.0f, 0.5f), 16777215.0f);
    const uint32_t qg = (uint32_t)fminf(fmaf(g + 1.0f, 8388608.0f, 0.5f), 16777215.0f);
    return make_uint3(qi | (qf << 24), (qf >> 8) | (qg << 16), (qg >> 16) | (qo << 8));
}
__device__ __forceinline__ void unpack_gates(uint3 w, float &i, float &f, float &g, float &o) {
    i = (float)(w.x & 0xffffffu) * (1.0f / 16777216.0f);
    f = (float)((w.x >> 24) | ((w.y & 0xffffu) << 8)) * (1.0f / 16777216.0f);
    g = (float)((w.y >> 16) | ((w.z & 0xffu) << 16)) * (1.0f / 8388608.0f) - 1.0f;
    o = (float)(w.z >> 8) * (1.0f / 16777216.0f);
}

// ---- weight-fragment loads the compiler must not re-schedule -------------------------------------------
// hipcc sinks ordinary loads of loop-invariant-addressable data next to their first use (it re-issues the
// load instead of carrying registers around the loop), which turns a software prefetch into a load -> wait ->
// MFMA chain with the full L2 latency exposed.  These two helpers keep the prefetch: an asm load the compiler
// cannot move, and a wait statement that names the destination registers so that every consumer is ordered
// behind it (cdna_hip_programming.md §5.7, form (ii)).  vmcnt is in-order: waiting for "all but the N youngest"
// also covers any older compiler-issued VMEM op, never fewer.
__device__ __forceinline__ void async_load_b128(f32x4 &dst, const void *ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void async_load_b32(float &dst, const void *ptr) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void async_load_b128(u32x4 &dst, const void *ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
// the same load with a scalar base: address = sbase (SGPR pair, wave-uniform) + voff (one VGPR) + IMM (< 4096).
// A stream of fragments then costs one offset register in total instead of a 64-bit pointer per fragment.
template <int IMM>
__device__ __forceinline__ void async_load_b128_s(u32x4 &dst, const void *sbase, uint32_t voff) {
    static_assert(IMM >= 0 && IMM < 4096, "13-bit signed instruction offset");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ void async_load_b128_s(f32x4 &dst, const void *sbase, uint32_t voff) {
    static_assert(IMM >= 0 && IMM < 4096, "13-bit signed instruction offset");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
}
// G (1, 2 or 4) consecutive 1 KB fragments starting at sbase
template <int G>
__device__ __forceinline__ void async_load_frags(u32x4 (&b)[G], const void *sbase, uint32_t voff) {
    static_assert(G == 1 || G == 2 || G == 4, "fragment groups of 1, 2 or 4");
    async_load_b128_s<0>(b[0], sbase, voff);
    if constexpr (G > 1) async_load_b128_s<1024>(b[1], sbase, voff);
    if constexpr (G > 2) {
        async_load_b128_s<2048>(b[2], sbase, voff);
        async_load_b128_s<3072>(b[3], sbase, voff);
    }
}
template <int N>
__device__ __forceinline__ void wait_vm(u32x4 &r0) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r0) : "i"(N));
}
template <int N>
__device__ __forceinline__ void wait_vm(u32x4 &r0, u32x4 &r1) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "i"(N));
}
template <int N>
__device__ __forceinline__ void wait_vm(u32x4 &r0, u32x4 &r1, u32x4 &r2, u32x4 &r3) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "i"(N));
}
template <int N, int G>
__device__ __forceinline__ void wait_frag(u32x4 (&b)[G]) {
    if constexpr (G == 1) wait_vm<N>(b[0]);
    else if constexpr (G == 2) wait_vm<N>(b[0], b[1]);
    else wait_vm<N>(b[0], b[1], b[2], b[3]);
}
// wait for every outstanding VMEM op; names three 8-register groups so their consumers stay below the wait
__device__ __forceinline__ void wait_vm_all(float (&a)[8], float (&b)[8], float (&c)[8]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                   "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
}
template <int N>
__device__ __forceinline__ void wait_vm(float &r0, float &r1, float &r2, float &r3) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "i"(N));
}
template <int N>
__device__ __forceinline__ void wait_vm(f32x4 &r0) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r0) : "i"(N));
}
template <int N>
__device__ __forceinline__ void wait_vm(f32x4 &r0, f32x4 &r1) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "i"(N));
}
template <int N>
__device__ __forceinline__ void wait_vm(f32x4 &r0, f32x4 &r1, f32x4 &r2, f32x4 &r3) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "i"(N));
}
template <int N>
__device__ __forceinline__ void wait_vm(f32x4 &r0, f32x4 &r1, f32x4 &r2, f32x4 &r3, f32x4 &r4, f32x4 &r5, f32x4 &r6,
                                        f32x4 &r7) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                 : "i"(N));
}
template <int N, int G>
__device__ __forceinline__ void wait_frag(f32x4 (&b)[G]) {
    if constexpr (G == 1) wait_vm<N>(b[0]);
    else if constexpr (G == 2) wait_vm<N>(b[0], b[1]);
    else wait_vm<N>(b[0], b[1], b[2], b[3]);
}

// element at `bytes` (32-bit, zero-extended) past a wave-uniform 64-bit base: the form hipcc turns into
// global_load/store with an SGPR base and ONE offset VGPR (a 64-bit index would cost an address pair per access)
template <class T>
__device__ __forceinline__ T &at_bytes(T *base, uint32_t bytes) {
    return *reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(base) + bytes);
}
template <class T>
__device__ __forceinline__ const T &at_bytes(const T *base, uint32_t bytes) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const unsigned char *>(base) + bytes);
}

// the lane id (0..63) recomputed from the hardware, never common-subexpression'd: per-step address arithmetic derived
// from it does not become a set of loop invariants living (and spilling) across the MFMA loops
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The wave's total as a wave-uniform value, on the DPP path: four row rotations leave every lane its row's sum, the four rows'
// sums are read with v_readlane -- ~12 VALU instead of six ds_bpermute round trips (~100 cycles each) of the butterfly above.
// Another order of additions than wave_sum (and one that differs from lane to lane before the read-out: only the returned
// value is meant to be used).
__device__ __forceinline__ float wave_sum_u(float v) {
#define PN_ROW_ROR(n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
    v += PN_ROW_ROR(8);
    v += PN_ROW_ROR(4);
    v += PN_ROW_ROR(2);
    v += PN_ROW_ROR(1);
#undef PN_ROW_ROR
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Gate activations.  PN_FAST_ACT (default): v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE division and
// the ocml expf/tanhf call sequences -- ~6 VALU per activation instead of ~40; abs error < 2e-7, far inside the
// 1e-5 output contract (tests compare against the oracle's torch.sigmoid / torch.tanh).
#ifndef PN_FAST_ACT
#define PN_FAST_ACT 1
#endif
#if PN_FAST_ACT
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    // tanh(x) = 1 - 2 / (1 + e^{2x});  e^{2x} = 2^{x * 2 log2 e}.  Saturates correctly for |x| large.
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}
#else
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }
#endif

// Philox4x32-10 as a stateless hash: four 32-bit words for (seed, counter).  Used for the dropout
// masks so that forward and backward regenerate the same mask without storing it.
__device__ __forceinline__ uint4 philox4(uint64_t seed, uint64_t ctr, uint32_t stream) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream, c3 = 0x50415448u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)m1;
        const uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// keep-mask scale for 4 consecutive elements starting at element index 4*vec (keep prob 1-p)
__device__ __forceinline__ float4 dropout4(uint64_t seed, uint64_t vec, uint32_t stream, float p) {
    const uint4 r = philox4(seed, vec, stream);
    const float scale = 1.0f / (1.0f - p);
    const uint32_t cut = (uint32_t)fminf(p * 4294967296.0f, 4294967040.0f);
    return make_float4(r.x >= cut ? scale : 0.0f, r.y >= cut ? scale : 0.0f, r.z >= cut ? scale : 0.0f,
                       r.w >= cut ? scale : 0.0f);
}

// the same mask, one element at a time (element index e of the [.., H] tensor the mask covers)
__device__ __forceinline__ float dropout1(uint64_t seed, uint64_t e, uint32_t stream, float p) {
    const float4 m = dropout4(seed, e >> 2, stream, p);
    const int k = (int)(e & 3);
    return k == 0 ? m.x : k == 1 ? m.y : k == 2 ? m.z : m.w;
}

}  // namespace pn
