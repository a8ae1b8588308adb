// pn_gemm.hip -- the dense GEMMs of the aggregator's node-level layers (fc0, the distance bank, their backward, the classifier's
// weight gradient; /root/reference/PathNet_run.py:175, :185-192 / :242-257, :210 / :277 and autograd's backward of them):
//   gemm_kernel    fp32-input MFMA, 64 x 64 tiles, any operand layout, ReLU gate, row indirection (compact bank rows), K split
//   gemm3_kernel   fp32 results from the bf16 matrix pipe (three planes, six MFMAs per product), 128 x 128 tiles: large graphs
//   gemm_finish_kernel / colsum_kernel / transpose_kernel and the launchers (pn_gemm.h).  gfx950 only.
// Split out of pn_pagg.hip in round 6; the kernels are unchanged.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "pn_gemm.h"
#include "pn_internal.h"
#include "pn_kernels.h"

namespace pn {

// ================================================================================================
// generic fp32 GEMM on MFMA:  C[m][n] (op)= act( sum_k A(m,k) * B(n,k) + bias[n] )
//   A(m,k) = A[m*sAm + k*sAk] (optionally multiplied by [gateA(m,k) > 0]), B(n,k) = B[n*sBn + k*sBk];
//   exactly one stride of each operand is 1.  64x64 block tile, 4 waves of 32x32, K tile 32, both
//   operand tiles K-major in LDS (pitch 65) so every MFMA operand fetch is a conflict-free
//   ds_read_b32 of 32 consecutive floats per half-wave.
// ================================================================================================
constexpr int GEMM_PITCH = 65;

struct GemmParams {
    const float *A;
    int64_t sAm, sAk;
    const float *gateA;
    const float *B;
    int64_t sBn, sBk;
    float *C;
    int64_t ldc;
    const float *bias;
    int M, N, K;
    int relu, mode;
    int kchunk;  // K range per blockIdx.z
    float *rowsum;  // optional [M]: += sum_k A(m,k) (after gating) -- the bias gradient that goes with a dW GEMM
                    // (PARTIAL: [nz][M], chunk z stores its own sums)
    // Compact rows (the distance bank over the (node, code) rows a batch touches, GEMM_IND_*): `list` holds the node of every
    // compact row, the launch covers rows [seg[0], seg[1]) of it -- counts that exist in device memory only; M (or K) given
    // on the host is their upper bound (it sizes the grid), workgroups past the real count leave at once.
    const int32_t *seg, *list;
    int ind;
    // operand range of the fp16 recurrence taken where the values are produced (run_tables; SeqRange of pn_seq.h, whose first
    // three words these point to): absmax[0] = atomicMax of the bit patterns of |C| over what this launch stores (after bias /
    // ReLU); absmax[1] += the biased exponents, absmax[2] += the count of the non-zero 32 x 32 sub-tile maxima of every eighth
    // workgroup (the spread statistic); clear_word[0..2] are set to 0 by one thread of the launch -- the GEMM in front of the
    // one that takes the maximum, on the same stream
    uint32_t *absmax, *clear_word;
};
// (IND is a template parameter: as run-time branches in front of the 24 loads of a K tile the row indirection halved the
//  speed of every GEMM, indirect or not)
template <bool A_KCONTIG, bool B_KCONTIG, bool GATE, int IND>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    __shared__ float As[GEMM_KT * GEMM_PITCH];
    __shared__ float Bs[GEMM_KT * GEMM_PITCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GEMM_BM, n0 = blockIdx.x * GEMM_BN;
    int pM = p.M, pK = p.K, ib = 0;
    if (IND != GEMM_IND_NONE) {       // (block-uniform)
        ib = p.seg[0];
        const int cnt = p.seg[1] - ib;
        if (IND == GEMM_IND_K) pK = min(pK, cnt); else pM = min(pM, cnt);
        // (a chunk past the real K leaves at once -- except in PARTIAL mode, where the finish kernel adds up every chunk:
        //  it stores zeros)
        if (m0 >= pM || ((int)blockIdx.z * p.kchunk >= pK && p.mode != GEMM_PARTIAL)) return;
    }
    if (p.clear_word && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && tid == 0) p.clear_word[0] = p.clear_word[1] = p.clear_word[2] = 0u;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = max(kbeg, min(pK, kbeg + p.kchunk));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;

    // Software pipeline: the next K tile's 8+8(+8) global loads are issued (branch-free, clamped addresses;
    // asm-pinned so hipcc cannot sink them) before the MFMAs of the current tile; out-of-range elements and
    // gated-off elements become zeros when the tile is written to LDS.
    float ra[8], rb[8], rg[8];
    float rsum = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) rg[i] = 1.0f;
    auto issue = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = tid + 256 * i;
            const int am = A_KCONTIG ? (idx >> 5) : (idx & 63), ak = A_KCONTIG ? (idx & 31) : (idx >> 6);
            int arow = min(m0 + am, pM - 1), acol = min(k0 + ak, kend - 1);
            if (IND == GEMM_IND_A_ROWS) arow = p.list[ib + arow];
            else if (IND == GEMM_IND_C_ROWS) arow += ib;
            else if (IND == GEMM_IND_K) acol += ib;
            const int64_t at = (int64_t)arow * p.sAm + (int64_t)acol * p.sAk;
            async_load_b32(ra[i], p.A + at);
            if (GATE) async_load_b32(rg[i], p.gateA + at);
            const int bn = B_KCONTIG ? (idx >> 5) : (idx & 63), bk = B_KCONTIG ? (idx & 31) : (idx >> 6);
            int bcol = min(k0 + bk, kend - 1);
            if (IND == GEMM_IND_K) bcol = p.list[ib + bcol];
            async_load_b32(rb[i], p.B + (int64_t)min(n0 + bn, p.N - 1) * p.sBn + (int64_t)bcol * p.sBk);
        }
    };
    if (kbeg < kend) issue(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GEMM_KT) {
        wait_vm_all(ra, rb, rg);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = tid + 256 * i;
            const int am = A_KCONTIG ? (idx >> 5) : (idx & 63), ak = A_KCONTIG ? (idx & 31) : (idx >> 6);
            const int bn = B_KCONTIG ? (idx >> 5) : (idx & 63), bk = B_KCONTIG ? (idx & 31) : (idx >> 6);
            const bool a_ok = (m0 + am < pM) && (k0 + ak < kend) && (!GATE || rg[i] > 0.0f);
            const bool b_ok = (n0 + bn < p.N) && (k0 + bk < kend);
            As[ak * GEMM_PITCH + am] = a_ok ? ra[i] : 0.0f;
            Bs[bk * GEMM_PITCH + bn] = b_ok ? rb[i] : 0.0f;
        }
        __syncthreads();
        issue(min(k0 + GEMM_KT, kend - 1));   // last trip: harmless re-load, drained below
        if (p.rowsum && blockIdx.x == 0 && tid < GEMM_BM) {
#pragma unroll
            for (int k = 0; k < GEMM_KT; k++) rsum += As[k * GEMM_PITCH + tid];
        }
#pragma unroll
        for (int kk = 0; kk < GEMM_KT / 2; kk++) {
            const float a = As[(2 * kk + hk) * GEMM_PITCH + wm * 32 + li];
            const float b = Bs[(2 * kk + hk) * GEMM_PITCH + wn * 32 + li];
            acc = mfma32(a, b, acc);
        }
        __syncthreads();
    }
    if (kbeg < kend) wait_vm_all(ra, rb, rg);
    if (p.rowsum && blockIdx.x == 0 && tid < GEMM_BM && m0 + tid < pM) {
        if (p.mode == GEMM_PARTIAL)
            p.rowsum[(int64_t)blockIdx.z * p.M + m0 + tid] = rsum;       // [nz][M] chunk sums (gemm_finish_kernel)
        else
            atomicAdd(&p.rowsum[m0 + tid], rsum);
    }
    const int col = n0 + wn * 32 + li;
    float vmax = 0.0f;
    if (p.absmax) {             // (only GEMM_STORE launches ask for it: what is stored is the final value)
        const float bias_m = (p.bias && col < p.N) ? p.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + acc_row(r, lane);
            float v = acc[r] + bias_m;
            if (p.relu) v = fmaxf(v, 0.0f);
            if (row < pM && col < p.N) vmax = fmaxf(vmax, fabsf(v));
        }
        vmax = wave_max(vmax);
        // one atomic per wave at most, and none once a larger value is in (a plain read first: stale is fine, it only grows)
        if (lane == 0 && vmax > 0.0f && __float_as_uint(vmax) > *reinterpret_cast<volatile uint32_t *>(p.absmax))
            atomicMax(p.absmax, __float_as_uint(vmax));
        if (lane == 0 && wave == 0 && vmax > 0.0f && ((blockIdx.x + blockIdx.y) & 7u) == 0u) {
            atomicAdd(reinterpret_cast<int *>(p.absmax + 1), (int)((__float_as_uint(vmax) >> 23) & 0xffu));
            atomicAdd(p.absmax + 2, 1u);
        }
    }
    if (col >= p.N) return;
    const float bias = (p.bias && blockIdx.z == 0 && p.mode != GEMM_PARTIAL) ? p.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 32 + acc_row(r, lane);
        if (row >= pM) continue;
        float v = acc[r] + bias;
        if (p.mode == GEMM_PARTIAL) {
            p.C[((int64_t)blockIdx.z * p.M + row) * p.ldc + col] = v;
            continue;
        }
        if (p.relu) v = fmaxf(v, 0.0f);
        const int crow = IND == GEMM_IND_A_ROWS ? ib + row : IND == GEMM_IND_C_ROWS ? p.list[ib + row] : row;
        float *dst = p.C + (int64_t)crow * p.ldc + col;
        if (p.mode == GEMM_STORE)
            *dst = v;
        else if (p.mode == GEMM_ADD)
            *dst += v;
        else
            atomicAdd(dst, v);
    }
}

// ---- the same GEMM for a reduction of at most 128 (four K tiles), both operands K-contiguous, no gate, no row list, one
// K chunk, stored: the distance bank (K = hid), every forward.  gemm_kernel keeps ONE K tile of loads in flight; with four
// tiles in all that is four dependent round trips to operands the launch in front has just written (21 us for 0.35 GFLOP at the
// headline shape).  Here the loads of all four tiles leave at once (64 registers) and each tile is consumed under a counted
// wait; the LDS layout, the MFMA sequence and the epilogue are gemm_kernel's, so the values are bit for bit the same.
template <int N>
__device__ __forceinline__ void wait_vm_tile(float (&a)[8], float (&b)[8]) {
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                 : "i"(N));
}
__global__ __launch_bounds__(256) void gemm_k128_kernel(GemmParams p) {
    __shared__ float As[GEMM_KT * GEMM_PITCH];
    __shared__ float Bs[GEMM_KT * GEMM_PITCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GEMM_BM, n0 = blockIdx.x * GEMM_BN;
    if (p.clear_word && (blockIdx.x | blockIdx.y) == 0 && tid == 0) p.clear_word[0] = p.clear_word[1] = p.clear_word[2] = 0u;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    float ra[4][8], rb[4][8];
#pragma unroll
    for (int t = 0; t < 4; t++)        // (tiles past K: clamped re-loads, never used)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = tid + 256 * i, r = idx >> 5, k = min(GEMM_KT * t + (idx & 31), p.K - 1);
            async_load_b32(ra[t][i], p.A + (int64_t)min(m0 + r, p.M - 1) * p.sAm + k);
            async_load_b32(rb[t][i], p.B + (int64_t)min(n0 + r, p.N - 1) * p.sBn + k);
        }
    auto tile = [&](float (&a8)[8], float (&b8)[8], int k0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = tid + 256 * i, r = idx >> 5, k = idx & 31;
            As[k * GEMM_PITCH + r] = (m0 + r < p.M && k0 + k < p.K) ? a8[i] : 0.0f;
            Bs[k * GEMM_PITCH + r] = (n0 + r < p.N && k0 + k < p.K) ? b8[i] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GEMM_KT / 2; kk++) {
            const float a = As[(2 * kk + hk) * GEMM_PITCH + wm * 32 + li];
            const float b = Bs[(2 * kk + hk) * GEMM_PITCH + wn * 32 + li];
            acc = mfma32(a, b, acc);
        }
        __syncthreads();
    };
    wait_vm_tile<48>(ra[0], rb[0]);
    tile(ra[0], rb[0], 0);
    wait_vm_tile<32>(ra[1], rb[1]);
    if (p.K > GEMM_KT) tile(ra[1], rb[1], GEMM_KT);
    wait_vm_tile<16>(ra[2], rb[2]);
    if (p.K > 2 * GEMM_KT) tile(ra[2], rb[2], 2 * GEMM_KT);
    wait_vm_tile<0>(ra[3], rb[3]);
    if (p.K > 3 * GEMM_KT) tile(ra[3], rb[3], 3 * GEMM_KT);
    // ---- gemm_kernel's epilogue for a stored result
    const int col = n0 + wn * 32 + li;
    float vmax = 0.0f;
    if (p.absmax) {
        const float bias_m = (p.bias && col < p.N) ? p.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + acc_row(r, lane);
            float v = acc[r] + bias_m;
            if (p.relu) v = fmaxf(v, 0.0f);
            if (row < p.M && col < p.N) vmax = fmaxf(vmax, fabsf(v));
        }
        vmax = wave_max(vmax);
        // ONE atomic per workgroup (the tile buffer is free: every wave is past the last barrier): all workgroups of this launch
        // are resident at once and reach this point together -- with an atomic per wave ~1 400 of them queued up on one address
        if (lane == 0 && wave == 0 && vmax > 0.0f && ((blockIdx.x + blockIdx.y) & 7u) == 0u) {        // (the spread statistic: as gemm_kernel)
            atomicAdd(reinterpret_cast<int *>(p.absmax + 1), (int)((__float_as_uint(vmax) >> 23) & 0xffu));
            atomicAdd(p.absmax + 2, 1u);
        }
        if (lane == 0) As[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            const float wmax = fmaxf(fmaxf(As[0], As[1]), fmaxf(As[2], As[3]));
            if (wmax > 0.0f && __float_as_uint(wmax) > *reinterpret_cast<volatile uint32_t *>(p.absmax))
                atomicMax(p.absmax, __float_as_uint(wmax));
        }
    }
    if (col >= p.N) return;
    const float bias = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * 32 + acc_row(r, lane);
        if (row >= p.M) continue;
        float v = acc[r] + bias;
        if (p.relu) v = fmaxf(v, 0.0f);
        p.C[(int64_t)row * p.ldc + col] = v;
    }
}

template <bool GATE, int IND>
void launch_gemm_layout(hipStream_t stream, dim3 grid, bool ak, bool bk, const GemmParams &p) {
    if (ak && bk)
        hipLaunchKernelGGL((gemm_kernel<true, true, GATE, IND>), grid, dim3(256), 0, stream, p);
    else if (ak && !bk)
        hipLaunchKernelGGL((gemm_kernel<true, false, GATE, IND>), grid, dim3(256), 0, stream, p);
    else if (!ak && bk)
        hipLaunchKernelGGL((gemm_kernel<false, true, GATE, IND>), grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_kernel<false, false, GATE, IND>), grid, dim3(256), 0, stream, p);
}
template <bool GATE>
void launch_gemm_variant(hipStream_t stream, dim3 grid, bool ak, bool bk, const GemmParams &p) {
    switch (p.ind) {
        case GEMM_IND_A_ROWS: return launch_gemm_layout<GATE, GEMM_IND_A_ROWS>(stream, grid, ak, bk, p);
        case GEMM_IND_C_ROWS: return launch_gemm_layout<GATE, GEMM_IND_C_ROWS>(stream, grid, ak, bk, p);
        case GEMM_IND_K: return launch_gemm_layout<GATE, GEMM_IND_K>(stream, grid, ak, bk, p);
        default: return launch_gemm_layout<GATE, GEMM_IND_NONE>(stream, grid, ak, bk, p);
    }
}

int launch_gemm(hipStream_t stream, const float *A, int64_t sAm, int64_t sAk, const float *gateA, const float *B,
                int64_t sBn, int64_t sBk, float *C, int64_t ldc, const float *bias, int M, int N, int K, int relu,
                int mode, int ksplit, float *rowsum, int ind, const int32_t *seg, const int32_t *list, uint32_t *absmax,
                uint32_t *clear_word) {
    if (M <= 0 || N <= 0) return PN_OK;
    GemmParams p{A, sAm, sAk, gateA, B, sBn, sBk, C, ldc, bias, M, N, K, relu, mode, 0, rowsum, seg, list, ind, absmax, clear_word};
    if (absmax && (mode != GEMM_STORE || ksplit > 1)) PN_FAIL(PN_ERR_ARG, "internal: gemm absmax needs a storing launch");
    if (ksplit < 1) ksplit = 1;
    if (mode != GEMM_ATOMIC && mode != GEMM_PARTIAL) ksplit = 1;
    int kchunk = (K + ksplit - 1) / ksplit;
    kchunk = ((kchunk + GEMM_KT - 1) / GEMM_KT) * GEMM_KT;
    if (kchunk < GEMM_KT) kchunk = GEMM_KT;
    p.kchunk = kchunk;
    const int nz = K > 0 ? (K + kchunk - 1) / kchunk : 1;
    if (K <= 0 || M <= 0 || N <= 0) PN_FAIL(PN_ERR_ARG, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
    const bool ak = (sAk == 1), bk = (sBk == 1);
    dim3 grid((N + GEMM_BN - 1) / GEMM_BN, (M + GEMM_BM - 1) / GEMM_BM, nz);
    // a short reduction over K-contiguous operands, stored in one piece: every K tile's loads in flight at once (gemm_k128_kernel)
    if (ak && bk && !gateA && ind == GEMM_IND_NONE && mode == GEMM_STORE && nz == 1 && K <= 4 * GEMM_KT && !rowsum)
        hipLaunchKernelGGL(gemm_k128_kernel, grid, dim3(256), 0, stream, p);
    else if (gateA)
        launch_gemm_variant<true>(stream, grid, ak, bk, p);
    else
        launch_gemm_variant<false>(stream, grid, ak, bk, p);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

// ================================================================================================
// gemm3_kernel: C[m][n] = sum_k A[m*lda + k] * B[n*ldb + k] (+ bias[n]) with fp32 results from the bf16 matrix pipe
// (pn_kernels.h: three planes, six MFMAs per product) -- the step GEMMs of the generic recurrence (hid > 256), where
// the fp32-input MFMA of gemm_kernel is the bound.  Both operands K-contiguous, K a multiple of 32.
//   128 x 128 block tile, 4 waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32 x 16), K tile 32.  A thread fetches four
//   float4 of each operand per K tile (asm loads, one K tile ahead), splits them into the three planes on the way to
//   LDS; plane rows are 64 B of bf16 + 16 B of padding: conflict-free ds_read_b128 fragments.  60 KB of LDS: two
//   workgroups per CU.
// ================================================================================================
constexpr int G3_PITCH = 80, G3_PLANE = 128 * G3_PITCH;
struct Gemm3Params {
    const float *A;
    int64_t lda;
    const float *B;
    int64_t ldb;
    float *C;
    int64_t ldc;
    const float *bias;
    int M, N, K;
    // node-level GEMMs over row lists (the compact distance bank, GEMM_IND_A_ROWS / GEMM_IND_C_ROWS as in gemm_kernel: the
    // row count lives in device memory, M bounds it), a ReLU gate on A (element kept where gate > 0, same indexing as A),
    // ReLU on the result, C += instead of C =
    const float *gate;
    const int32_t *seg, *list;
    int relu, add;
};

template <int IND, bool GATE>
__global__ __launch_bounds__(256, 2) void gemm3_kernel(Gemm3Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * G3_PLANE];     // A planes 0..2 | B planes 0..2
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * G3_BM, n0 = blockIdx.x * G3_BN;
    int pM = p.M, ib = 0;
    if (IND != GEMM_IND_NONE) {         // (block-uniform)
        ib = p.seg[0];
        pM = min(pM, p.seg[1] - ib);
        if (m0 >= pM) return;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
    // staging: thread -> rows lr + 32 i (i < 4) of the tile, floats lk .. lk + 3 of the K tile (rows past M / N: clamped,
    // their results are never stored)
    const int lr = tid >> 3, lk = 4 * (tid & 7);
    const float *ap[4], *bp[4], *gp[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int arow = min(m0 + lr + 32 * i, pM - 1);
        if (IND == GEMM_IND_A_ROWS) arow = p.list[ib + arow];
        else if (IND == GEMM_IND_C_ROWS) arow += ib;
        ap[i] = p.A + (int64_t)arow * p.lda + lk;
        gp[i] = GATE ? p.gate + (int64_t)arow * p.lda + lk : nullptr;
        bp[i] = p.B + (int64_t)min(n0 + lr + 32 * i, p.N - 1) * p.ldb + lk;
    }
    f32x4 ra[4], rb[4], rgt[GATE ? 4 : 1];
    auto issue = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            async_load_b128(ra[i], ap[i] + k0);
            if constexpr (GATE) async_load_b128(rgt[i], gp[i] + k0);
            async_load_b128(rb[i], bp[i] + k0);
        }
    };
    auto commit = [&]() {
        wait_vm<0>(ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]);
        if constexpr (GATE) {       // (a second wait names the gate registers: nothing may read them before it)
            wait_vm<0>(rgt[0], rgt[1], rgt[2], rgt[3]);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) ra[i][e] = rgt[i][e] > 0.0f ? ra[i][e] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned char *da = lds + (lr + 32 * i) * G3_PITCH + 2 * lk, *db = da + 3 * G3_PLANE;
            uint32_t x0, x1, x2, y0, y1, y2;
            split3(ra[i][0], ra[i][1], x0, x1, x2);
            split3(ra[i][2], ra[i][3], y0, y1, y2);
            *reinterpret_cast<uint2 *>(da) = make_uint2(x0, y0);
            *reinterpret_cast<uint2 *>(da + G3_PLANE) = make_uint2(x1, y1);
            *reinterpret_cast<uint2 *>(da + 2 * G3_PLANE) = make_uint2(x2, y2);
            split3(rb[i][0], rb[i][1], x0, x1, x2);
            split3(rb[i][2], rb[i][3], y0, y1, y2);
            *reinterpret_cast<uint2 *>(db) = make_uint2(x0, y0);
            *reinterpret_cast<uint2 *>(db + G3_PLANE) = make_uint2(x1, y1);
            *reinterpret_cast<uint2 *>(db + 2 * G3_PLANE) = make_uint2(x2, y2);
        }
    };
    const unsigned char *fa = lds + (wm * 64 + li) * G3_PITCH + 16 * hk;
    const unsigned char *fb = lds + 3 * G3_PLANE + (wn * 64 + li) * G3_PITCH + 16 * hk;
    issue(0);
    for (int k0 = 0; k0 < p.K; k0 += G3_KT) {
        commit();
        __syncthreads();
        issue(min(k0 + G3_KT, p.K - G3_KT));        // last trip: harmless re-load, drained below (no branch before the wait)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            u32x4 a[2][3], b[2][3];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    a[t][pl] = *reinterpret_cast<const u32x4 *>(fa + pl * G3_PLANE + t * 32 * G3_PITCH + 32 * ks);
                    b[t][pl] = *reinterpret_cast<const u32x4 *>(fb + pl * G3_PLANE + t * 32 * G3_PITCH + 32 * ks);
                }
            // a2.b0 a1.b0 a0.b0 | a1.b1 a0.b1 | a0.b2, each over the four accumulators
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const int pa = q == 0 ? 2 : (q == 1 || q == 3) ? 1 : 0, pb = q < 3 ? 0 : q < 5 ? 1 : 2;
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = mfma_bf16(a[i][pa], b[j][pb], acc[i][j]);
            }
        }
        __syncthreads();
    }
    wait_vm<0>(ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]);
    if constexpr (GATE) wait_vm<0>(rgt[0], rgt[1], rgt[2], rgt[3]);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int col = n0 + wn * 64 + j * 32 + li;
        if (col >= p.N) continue;
        const float bias = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (row >= pM) continue;
                const int crow = IND == GEMM_IND_A_ROWS ? ib + row : IND == GEMM_IND_C_ROWS ? p.list[ib + row] : row;
                float v = acc[i][j][r] + bias;
                if (p.relu) v = fmaxf(v, 0.0f);
                float *dst = p.C + (int64_t)crow * p.ldc + col;
                if (p.add)
                    *dst += v;
                else
                    *dst = v;
            }
    }
}

int launch_gemm3(hipStream_t stream, const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                 const float *bias, int M, int N, int K, int relu, int add, const float *gate, int ind, const int32_t *seg,
                 const int32_t *list) {
    if (M <= 0 || N <= 0) return PN_OK;
    if (K < G3_KT || K % G3_KT != 0) PN_FAIL(PN_ERR_ARG, "gemm3: K=%d is not a multiple of %d", K, G3_KT);
    Gemm3Params p{A, lda, B, ldb, C, ldc, bias, M, N, K, gate, seg, list, relu, add};
    const dim3 grid((N + G3_BN - 1) / G3_BN, (M + G3_BM - 1) / G3_BM);
    if (ind == GEMM_IND_A_ROWS && !gate)
        hipLaunchKernelGGL((gemm3_kernel<GEMM_IND_A_ROWS, false>), grid, dim3(256), 0, stream, p);
    else if (ind == GEMM_IND_C_ROWS && gate)
        hipLaunchKernelGGL((gemm3_kernel<GEMM_IND_C_ROWS, true>), grid, dim3(256), 0, stream, p);
    else if (ind == GEMM_IND_C_ROWS)
        hipLaunchKernelGGL((gemm3_kernel<GEMM_IND_C_ROWS, false>), grid, dim3(256), 0, stream, p);
    else if (ind == GEMM_IND_NONE && gate)
        hipLaunchKernelGGL((gemm3_kernel<GEMM_IND_NONE, true>), grid, dim3(256), 0, stream, p);
    else if (ind == GEMM_IND_NONE)
        hipLaunchKernelGGL((gemm3_kernel<GEMM_IND_NONE, false>), grid, dim3(256), 0, stream, p);
    else
        PN_FAIL(PN_ERR_ARG, "gemm3: unsupported indirection %d", ind);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

// weights [R][C] -> [C][R] (the dX GEMMs of the node-level backward want the reduction index contiguous)
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int R, int C, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * C) return;
    const int r = (int)(i / C), c = (int)(i - (int64_t)r * C);
    out[(int64_t)c * R + r] = in[i];
}

bool gemm3_pays(const pn_context *ctx, int64_t M, int64_t N, int K, int which) {
    if (!(knobs_of(ctx).node_gemm3 & which)) return false;      // bit mask of G3_*: A/B runs and tests (default: all)
    return K >= G3_KT && K % G3_KT == 0 && ((M + G3_BM - 1) / G3_BM) * ((N + G3_BN - 1) / G3_BN) >= 384;
}

// ---- deterministic split-K for the STORE / ADD GEMMs -------------------------------------------------------------
// The node-level GEMMs (fc0: 2708 x 128 x 1433, the bank backward) are 86 workgroups of 64 x 64 -- a third of the
// CUs, each walking all of K alone, one wave per SIMD.  With K cut into nz chunks there are nz times as many
// workgroups; the chunk sums go to a [nz][M][N] buffer and one small kernel adds them up in a fixed order and applies
// bias / ReLU / accumulate.
__global__ __launch_bounds__(256) void gemm_finish_kernel(const float *__restrict__ part, int nz, int M, int N,
                                                          const float *__restrict__ bias, int relu, int mode,
                                                          float *__restrict__ C, int64_t ldc,
                                                          const float *__restrict__ rs_part = nullptr,
                                                          float *__restrict__ rowsum = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rowsum && i < M) {          // rowsum[m] += the chunk sums, in chunk order
        float v = 0.0f;
        for (int z = 0; z < nz; z++) v += rs_part[(int64_t)z * M + i];
        rowsum[i] += v;
    }
    if (i >= (int64_t)M * N) return;
    const int row = (int)(i / N), col = (int)(i - (int64_t)row * N);
    float v = bias ? bias[col] : 0.0f;
    // the chunk sums eight at a time, all eight loads in flight (asm loads: as a plain loop hipcc waits for each partial before
    // it asks for the next -- nz dependent round trips, the whole 6 us of this kernel at the headline shape); added in chunk order
    const float *pz = part + i;
    const int64_t mn = (int64_t)M * N;
    for (int z0 = 0; z0 < nz; z0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) async_load_b32(t[u], pz + (int64_t)min(z0 + u, nz - 1) * mn);
        wait_vm<0>(t[0], t[1], t[2], t[3]);
        wait_vm<0>(t[4], t[5], t[6], t[7]);
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (z0 + u < nz) v += t[u];
    }
    if (relu) v = fmaxf(v, 0.0f);
    float *dst = C + (int64_t)row * ldc + col;
    if (mode == GEMM_ADD)
        *dst += v;
    else
        *dst = v;
}

// out[n] (+)= sum_m A[m*ld + n] * [gate[m*ld+n] > 0]      (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ A, const float *__restrict__ gate,
                                                     int64_t ld, int M, int N, int rows_per_block,
                                                     float *__restrict__ out) {
    __shared__ float part[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.0f;
    if (n < N)
        for (int m = r0 + wave; m < r1; m += 4) {
            const int64_t at = (int64_t)m * ld + n;
            float v = A[at];
            if (gate && !(gate[at] > 0.0f)) v = 0.0f;
            s += v;
        }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) atomicAdd(out + n, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}


int launch_colsum(hipStream_t stream, const float *A, const float *gate, int64_t ld, int M, int N, float *out, bool det) {
    if (M <= 0 || N <= 0) return PN_OK;
    int ysplit = (M + 511) / 512;
    if (ysplit > 1024) ysplit = 1024;
    if (det) ysplit = 1;        // one workgroup per 64 columns walks every row: a single add per output
    const int rows_per_block = (M + ysplit - 1) / ysplit;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, ysplit), dim3(256), 0, stream, A, gate, ld, M, N,
                       rows_per_block, out);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

// K chunks a split node-level GEMM is cut into (1 = not split): aim at ~2 workgroups per CU, at least two K tiles each
int gemm_split_count(int M, int N, int K) {
    const int64_t tiles = (int64_t)((M + GEMM_BM - 1) / GEMM_BM) * ((N + GEMM_BN - 1) / GEMM_BN);
    int64_t nz = tiles > 0 ? (512 + tiles - 1) / tiles : 1;
    if (nz > GEMM_MAX_SPLIT) nz = GEMM_MAX_SPLIT;
    if (nz > K / (2 * GEMM_KT)) nz = K / (2 * GEMM_KT);
    return nz < 1 ? 1 : (int)nz;
}

int launch_gemm_split(hipStream_t stream, const float *A, int64_t sAm, int64_t sAk, const float *gateA, const float *B,
                      int64_t sBn, int64_t sBk, float *C, int64_t ldc, const float *bias, int M, int N, int K, int relu,
                      int mode, float *partial, uint32_t *clear_word) {
    int nz = gemm_split_count(M, N, K);
    if (nz <= 1 || !partial || M <= 0 || N <= 0)
        return launch_gemm(stream, A, sAm, sAk, gateA, B, sBn, sBk, C, ldc, bias, M, N, K, relu, mode, 1, nullptr, GEMM_IND_NONE,
                           nullptr, nullptr, nullptr, clear_word);
    int kchunk = (K + nz - 1) / nz;
    kchunk = (kchunk + GEMM_KT - 1) / GEMM_KT * GEMM_KT;
    nz = (K + kchunk - 1) / kchunk;
    if (int rc = launch_gemm(stream, A, sAm, sAk, gateA, B, sBn, sBk, partial, N, nullptr, M, N, K, 0, GEMM_PARTIAL, nz, nullptr,
                             GEMM_IND_NONE, nullptr, nullptr, nullptr, clear_word))
        return rc;
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(gemm_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, nz, M, N,
                       bias, relu, mode, C, ldc);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

// ---- deterministic weight-gradient GEMM (pn_pagg_shape.deterministic): C += A . B^T and rowsum += row sums of A with the
// reduction cut into at most DET_MAX_SPLIT chunks whose sums are stored ([nz][M][N], [nz][M]) and added up in chunk order
// -- where the default path lets the chunks race with atomics.  C and rowsum accumulate (micro-batches).
int launch_gemm_det(hipStream_t stream, const float *A, int64_t sAm, int64_t sAk, const float *gateA, const float *B,
                    int64_t sBn, int64_t sBk, float *C, int64_t ldc, int M, int N, int K, int ksplit, float *rowsum,
                    float *partial, int ind, const int32_t *seg, const int32_t *list) {
    if (M <= 0 || N <= 0 || K <= 0) return PN_OK;
    int nz = std::max(1, std::min(ksplit, DET_MAX_SPLIT));
    int kchunk = (K + nz - 1) / nz;
    kchunk = std::max(GEMM_KT, (kchunk + GEMM_KT - 1) / GEMM_KT * GEMM_KT);
    nz = (K + kchunk - 1) / kchunk;             // (= the grid's z extent launch_gemm derives from the same numbers)
    float *rs_part = rowsum ? partial + (size_t)nz * M * N : nullptr;
    if (int rc = launch_gemm(stream, A, sAm, sAk, gateA, B, sBn, sBk, partial, N, nullptr, M, N, K, 0, GEMM_PARTIAL, nz,
                             rs_part, ind, seg, list))
        return rc;
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(gemm_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, nz, M, N,
                       (const float *)nullptr, 0, (int)GEMM_ADD, C, ldc, (const float *)rs_part, rowsum);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int launch_transpose(hipStream_t stream, const float *in, int R, int C, float *out) {
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(((int64_t)R * C + 255) / 256)), dim3(256), 0, stream, in, R, C, out);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

}  // namespace pn
