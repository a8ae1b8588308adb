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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

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
