// pn_gemm.h -- the dense GEMM launchers of the aggregator's node-level layers (pn_gemm.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "pn_internal.h"

namespace pn {

// ---- gemm_kernel: C[m][n] (op)= act( sum_k A(m,k) * B(n,k) + bias[n] ) on the fp32-input MFMA, 64 x 64 tiles ------------------
//   A(m,k) = A[m*sAm + k*sAk] (optionally multiplied by [gateA(m,k) > 0]), B(n,k) = B[n*sBn + k*sBk]; exactly one stride of
//   each operand is 1.
constexpr int GEMM_BM = 64, GEMM_BN = 64, GEMM_KT = 32;
enum { GEMM_STORE = 0, GEMM_ADD = 1, GEMM_ATOMIC = 2, GEMM_PARTIAL = 3 };   // PARTIAL: raw sums of K chunk z to C[z][M][ldc]
// Compact rows (the distance bank over the (node, code) rows a batch touches): `list` holds the node of every compact row, the
// launch covers rows [seg[0], seg[1]) of it -- counts that exist in device memory only
enum { GEMM_IND_NONE = 0,
       GEMM_IND_A_ROWS = 1,     // A row m = list[b + m] (rows of Xh by node), C row m = b + m (compact rows)
       GEMM_IND_C_ROWS = 2,     // A row m = b + m (compact rows), C row m = list[b + m]
       GEMM_IND_K = 3 };        // reduction index k: A column k = b + k (compact rows), B column k = list[b + k]

// ksplit > 1 (ATOMIC / PARTIAL modes): the reduction is cut into that many chunks over blockIdx.z.  rowsum: optional [M] +=
// sum_k A(m,k) after gating (the bias gradient that goes with a dW GEMM).  absmax / clear_word: the fp16 recurrence's operand
// range taken where the values are produced (GemmParams in pn_gemm.hip).
int launch_gemm(hipStream_t stream, const float *A, int64_t sAm, int64_t sAk, const float *gateA, const float *B, int64_t sBn,
                int64_t sBk, float *C, int64_t ldc, const float *bias, int M, int N, int K, int relu, int mode, int ksplit,
                float *rowsum = nullptr, int ind = GEMM_IND_NONE, const int32_t *seg = nullptr, const int32_t *list = nullptr,
                uint32_t *absmax = nullptr, uint32_t *clear_word = nullptr);
// K chunks a split node-level GEMM is cut into (1 = not split): aim at ~2 workgroups per CU, at least two K tiles each
constexpr int GEMM_MAX_SPLIT = 8;
int gemm_split_count(int M, int N, int K);
// STORE / ADD GEMM with the reduction split over gemm_split_count chunks: chunk sums to `partial` ([nz][M][N]), then one fixed-order
// finish launch (bias / ReLU / accumulate)
int launch_gemm_split(hipStream_t stream, const float *A, int64_t sAm, int64_t sAk, const float *gateA, const float *B, int64_t sBn,
                      int64_t sBk, float *C, int64_t ldc, const float *bias, int M, int N, int K, int relu, int mode, float *partial,
                      uint32_t *clear_word = nullptr);
// deterministic weight-gradient GEMM: C += A . B^T and rowsum += row sums of A, chunk sums stored and added in chunk order
constexpr int DET_MAX_SPLIT = 32;
inline size_t det_gemm_floats(size_t M, size_t N) { return (size_t)DET_MAX_SPLIT * (M * N + M); }
int launch_gemm_det(hipStream_t stream, const float *A, int64_t sAm, int64_t sAk, const float *gateA, const float *B, int64_t sBn,
                    int64_t sBk, float *C, int64_t ldc, int M, int N, int K, int ksplit, float *rowsum, float *partial,
                    int ind = GEMM_IND_NONE, const int32_t *seg = nullptr, const int32_t *list = nullptr);
// out[n] (+)= sum_m A[m*ld + n] * [gate[m*ld+n] > 0]  (bias gradients); det: one workgroup per 64 columns, a single add per output
int launch_colsum(hipStream_t stream, const float *A, const float *gate, int64_t ld, int M, int N, float *out, bool det = false);
int launch_transpose(hipStream_t stream, const float *in, int R, int C, float *out);      // out [C, R] = in [R, C]^T

// ---- gemm3_kernel: C[m][n] = sum_k A[m*lda + k] * B[n*ldb + k] (+ bias[n]) with fp32 results from the bf16 matrix pipe (three
//      planes, six MFMAs per product), 128 x 128 tiles; both operands K-contiguous, K a multiple of 32 ------------------------------
constexpr int G3_BM = 128, G3_BN = 128, G3_KT = 32;
int launch_gemm3(hipStream_t stream, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias,
                 int M, int N, int K, int relu = 0, int add = 0, const float *gate = nullptr, int ind = GEMM_IND_NONE,
                 const int32_t *seg = nullptr, const int32_t *list = nullptr);
// the bf16 x 3 GEMM pays once its 128 x 128 tiles fill the GPU; `which`: bit of the context knob PN_NODE_GEMM3
enum { G3_FC0 = 1, G3_BANK = 2, G3_BANK_DX = 4 };
bool gemm3_pays(const pn_context *ctx, int64_t M, int64_t N, int K, int which);

}  // namespace pn
