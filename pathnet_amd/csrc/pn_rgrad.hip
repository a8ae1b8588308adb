// pn_rgrad.hip -- the node-level weight gradients of large graphs: C [M, N] += A^T . B with the reduction over the rows of
// the graph (g_bank_w = dZ'^T . Xh, g_fc0_w = dXh'^T . X; PathNet_run.py:175,185-191 through autograd), M and N a few
// hundred, the reduction millions.  wgrad4_kernel's structure (pn_seq4.hip) cut for a 128 x 128 output tile: both operands
// have the reduction dimension outermost, a thread fetches a 4-row x 4-column fp32 block, splits it into the three bf16
// planes (pn_kernels.h: six bf16 MFMAs = one fp32-accurate product) and writes 8-byte k-octet halves; K tiles of 32 rows,
// two LDS stages, the commit of tile i + 1 between the MFMA groups of tile i, one barrier per tile ahead of the last
// group, under which the next tile's first fragments are fetched.  The kernel is bound by the HBM stream of its operands
// (1 KB per reduction row, 1.5 KB with a ReLU gate).
//   GATE  A is gated element-wise by gate > 0 (the ReLU of the homo class: gate = Z or Xh, same indexing as A)
//   LIST  compact rows (touched-row bank): A row = seg[0] + k, B row = list[seg[0] + k], k < seg[1] - seg[0] (device memory)
// Every workgroup adds its tile to C (and its column sums of the gated A to rowsum) with atomics at the end: a few hundred
// workgroups, 16 384 adds each.  Not used in deterministic mode (pn_pagg.hip keeps the chunk-sum GEMM there).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "pn_internal.h"
#include "pn_kernels.h"
#include "pn_seq.h"

using namespace pn;

namespace {

constexpr int R_BM = 128, R_BN = 128, R_KT = 32, R_THREADS = 512;
constexpr int R_BLK = 4 * 36;                               // 16-byte slots per (plane, operand, k-octet) block of 128 columns
constexpr int R_PLANE = 2 * 4 * R_BLK;                      // slots per plane: 2 operands x 4 k-octets
constexpr int R_STAGE = 3 * R_PLANE;                        // slots per stage (55 296 bytes)
constexpr int R_LDS_BYTES = 2 * R_STAGE * 16;               // 110 592

template <bool GATE, bool LIST>
__global__ __launch_bounds__(R_THREADS, 2) void rgrad_kernel(RgradParams p) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, hk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;        // wave tile: 32 rows of M x 64 columns of N
    const int m0 = blockIdx.y * R_BM, n0 = blockIdx.x * R_BN;
    int64_t R = p.R;
    int64_t ib = 0;
    if (LIST) {                                     // (block-uniform)
        ib = p.seg[0];
        R = std::min<int64_t>(R, (int64_t)p.seg[1] - ib);
    }
    // split z takes the K tiles z, z + nz, z + 2 nz, ...
    const int64_t ntiles = (R + R_KT - 1) / R_KT;
    const int64_t nz = gridDim.z;
    const int64_t my_tiles = (int64_t)blockIdx.z < ntiles ? (ntiles - blockIdx.z + nz - 1) / nz : 0;
    if (my_tiles == 0) return;                      // block-uniform
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.0f;

    // staging task of this thread: operand op, rows 4 rq .. 4 rq + 3 of the K tile, columns 4 cq .. 4 cq + 3
    const int op = tid >> 8, rq = (tid >> 5) & 7, cq = tid & 31;
    const float *src = op == 0 ? p.A : p.B;
    const int64_t ld = op == 0 ? p.lda : p.ldb;
    const int c0 = (op == 0 ? m0 : n0) + 4 * cq;
    const bool c_ok = c0 < (op == 0 ? p.M : p.N);
    const float *srcc = src + (c_ok ? c0 : 0);
    const float *gatec = GATE ? p.gate + (c_ok ? c0 : 0) : nullptr;
    const bool gated = GATE && op == 0;             // (wave-uniform: waves 0..3 stage A)
    f32x4 rgA[4], rgB[4], ggA[GATE ? 4 : 1], ggB[GATE ? 4 : 1];
    float lidx[LIST ? 4 : 1];                       // B rows of the tile after next (bit patterns of int32)
    auto row0_of = [&](int64_t i) { return (blockIdx.z + std::min(i, my_tiles - 1) * nz) * R_KT; };     // (clamped: harmless re-load)
    auto issue_list = [&](int64_t i) {              // operand-1 threads: the node of each of their four rows
        if constexpr (LIST) {
            const int64_t k0 = row0_of(i);
#pragma unroll
            for (int e = 0; e < 4; e++) async_load_b32(lidx[e], p.list + ib + std::min(k0 + 4 * rq + e, R - 1));
        }
    };
    auto issue = [&](f32x4 (&rg)[4], f32x4 (&gg)[GATE ? 4 : 1], int64_t i) {
        const int64_t k0 = row0_of(i);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            int64_t row = ib + std::min(k0 + 4 * rq + e, R - 1);
            if constexpr (LIST)
                if (op == 1) row = (int64_t)__float_as_int(lidx[e]);     // (the list entries have landed: wait_vm<0> before)
            async_load_b128(rg[e], srcc + row * ld);
            // (the B threads have no gate: a dummy load of one hot line keeps the instruction stream free of branches)
            if constexpr (GATE) async_load_b128(gg[e], gated ? gatec + row * ld : srcc);
        }
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};             // column sums of the gated A over this thread's rows (bias gradient)
    // operand op, k-octet rq >> 1, column 4 cq + j at slot j * 36 + cq; this thread's rows are half (rq & 1) of the octet
    unsigned char *stage_wr = reinterpret_cast<unsigned char *>(lds4 + (op * 4 + (rq >> 1)) * R_BLK + cq) + (rq & 1) * 8;
    auto piece = [&](f32x4 (&rg)[4], f32x4 (&gg)[GATE ? 4 : 1], int64_t i, int buf, int j) {
        const int64_t k0 = row0_of(i);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            bool keep = c_ok && i < my_tiles && k0 + 4 * rq + e < R;
            if constexpr (GATE) keep = keep && (!gated || gg[e][j] > 0.0f);
            v[e] = keep ? rg[e][j] : 0.0f;
        }
        unsigned char *w = stage_wr + (size_t)buf * (R_STAGE * 16);
        uint32_t x0, x1, x2, y0, y1, y2;
        split3(v[0], v[1], x0, x1, x2);
        split3(v[2], v[3], y0, y1, y2);
        bs[j] += (v[0] + v[1]) + (v[2] + v[3]);
        *reinterpret_cast<uint2 *>(w + j * 36 * 16) = make_uint2(x0, y0);
        *reinterpret_cast<uint2 *>(w + (R_PLANE + j * 36) * 16) = make_uint2(x1, y1);
        *reinterpret_cast<uint2 *>(w + (2 * R_PLANE + j * 36) * 16) = make_uint2(x2, y2);
    };
    // fragment slots: operand 0 columns 32 wm + li, operand 1 columns 64 wn + 32 j + li; k-octet 2 kk + hk of the tile
    const int sa = hk * R_BLK + (li & 3) * 36 + (li >> 2) + wm * 8;
    const int sb = (4 + hk) * R_BLK + (li & 3) * 36 + (li >> 2) + wn * 16;
#define R_FENCE() __builtin_amdgcn_sched_barrier(0)
    // fragments of k-step kk of the tile in a stage: a[plane], b[plane][j]
    u32x4 a0[3], b0[3][2], a1[3], b1[3][2];
    auto frags = [&](int buf, int kk, u32x4 (&a)[3], u32x4 (&b)[3][2]) {
        const u32x4 *fa = lds4 + buf * R_STAGE + sa + 2 * kk * R_BLK, *fb = lds4 + buf * R_STAGE + sb + 2 * kk * R_BLK;
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            a[pl] = fa[pl * R_PLANE];
            b[pl][0] = fb[pl * R_PLANE];
            b[pl][1] = fb[pl * R_PLANE + 8];
        }
    };
    // the six products of a k-step, a0.b0 a0.b1 a1.b0 | a1.b1 a0.b2 a2.b0, each over the wave's two output tiles
    auto prods = [&](const u32x4 (&a)[3], const u32x4 (&b)[3][2], int from, int to) {
        R_FENCE();
#pragma unroll
        for (int q = from; q < to; q++) {
            const int pa = q == 2 || q == 3 ? 1 : q == 5 ? 2 : 0, pb = q == 1 || q == 3 ? 1 : q == 4 ? 2 : 0;
#pragma unroll
            for (int j = 0; j < 2; j++) acc[j] = mfma_bf16(a[pa], b[pb][j], acc[j]);
        }
        R_FENCE();
    };
    // One step: k-step 0 (fragments already in a0 / b0), half of the commit, k-step 1's fragments, its first three products,
    // the other half of the commit, the barrier, the last three products with the next tile's k-step-0 fragments fetched
    // under them.
    auto step = [&](int buf, f32x4 (&rg)[4], f32x4 (&gg)[GATE ? 4 : 1], int64_t inext) {
        frags(buf, 1, a1, b1);
        prods(a0, b0, 0, 6);
        piece(rg, gg, inext, buf ^ 1, 0);
        piece(rg, gg, inext, buf ^ 1, 1);
        prods(a1, b1, 0, 3);
        piece(rg, gg, inext, buf ^ 1, 2);
        piece(rg, gg, inext, buf ^ 1, 3);
        R_FENCE();
        __syncthreads();        // stage buf ^ 1 is complete; every read of stage buf has been issued and has landed
        frags(buf ^ 1, 0, a0, b0);
        prods(a1, b1, 3, 6);
    };

    issue_list(0);
    if constexpr (LIST) wait_vm<0>(lidx[0], lidx[1], lidx[2], lidx[3]);
    issue(rgA, ggA, 0);
    issue_list(1);
    wait_vm<0>(rgA[0], rgA[1], rgA[2], rgA[3]);
    if constexpr (GATE) wait_vm<0>(ggA[0], ggA[1], ggA[2], ggA[3]);
    if constexpr (LIST) wait_vm<0>(lidx[0], lidx[1], lidx[2], lidx[3]);
#pragma unroll
    for (int j = 0; j < 4; j++) piece(rgA, ggA, 0, 0, j);
    issue(rgB, ggB, 1);
    issue_list(2);
    __syncthreads();
    frags(0, 0, a0, b0);
    // two tiles per trip (register sets and stages alternate); an odd count runs one tile of zeros.  Everything issued in
    // the previous half trip is waited for at the top of the next one (vmcnt(0): the rows of the next tile, the list entries
    // of the one after), then the loads of the tile after next go out.
#pragma unroll 1
    for (int64_t i = 0; i < my_tiles; i += 2) {
        wait_vm<0>(rgB[0], rgB[1], rgB[2], rgB[3]);
        if constexpr (GATE) wait_vm<0>(ggB[0], ggB[1], ggB[2], ggB[3]);
        if constexpr (LIST) wait_vm<0>(lidx[0], lidx[1], lidx[2], lidx[3]);
        issue(rgA, ggA, i + 2);
        issue_list(i + 3);
        step(0, rgB, ggB, i + 1);
        wait_vm<0>(rgA[0], rgA[1], rgA[2], rgA[3]);
        if constexpr (GATE) wait_vm<0>(ggA[0], ggA[1], ggA[2], ggA[3]);
        if constexpr (LIST) wait_vm<0>(lidx[0], lidx[1], lidx[2], lidx[3]);
        issue(rgB, ggB, i + 3);
        issue_list(i + 4);
        step(1, rgA, ggA, i + 2);
    }
    wait_vm<0>(rgA[0], rgA[1], rgA[2], rgA[3]);     // drain the trailing (clamped) loads
    wait_vm<0>(rgB[0], rgB[1], rgB[2], rgB[3]);
    if constexpr (GATE) {
        wait_vm<0>(ggA[0], ggA[1], ggA[2], ggA[3]);
        wait_vm<0>(ggB[0], ggB[1], ggB[2], ggB[3]);
    }
    if constexpr (LIST) wait_vm<0>(lidx[0], lidx[1], lidx[2], lidx[3]);
    __syncthreads();        // (the bias sums below reuse the stages)
#undef R_FENCE
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n = n0 + wn * 64 + j * 32 + li;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + wm * 32 + acc_row(r, lane);
            if (m < p.M) atomicAdd(p.C + (int64_t)m * p.ldc + n, acc[j][r]);
        }
    }
    // bias gradient: the eight row-quad owners of a column add up through LDS
    if (!p.rowsum || blockIdx.x != 0) return;       // block-uniform
    float *fl = reinterpret_cast<float *>(lds4);
    if (op == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) fl[rq * R_BM + 4 * cq + j] = bs[j];
    }
    __syncthreads();
    if (tid < R_BM && m0 + tid < p.M) {
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; q++) s += fl[q * R_BM + tid];
        atomicAdd(p.rowsum + m0 + tid, s);
    }
}

}  // namespace

namespace pn {

bool rgrad_pays(const pn_context *ctx, int64_t R, int M, int N) {
    if (knobs_of(ctx).node_rgrad == 0) return false;    // 0: never (A/B runs, tests)
    if (M % 4 != 0 || N % 4 != 0) return false;
    // Round 3 chose it from ~49 000 rows up, stand-alone.  In a training step these GEMMs run on the eighth of the CUs the
    // recurrent weight-gradient GEMM leaves free, where the fp32-input MFMA's rate (1 / 16 of the bf16 pipe's) is what
    // bounds them: from a few thousand rows on the bf16 x 3 kernel is the shorter one there (round 6, PN_NODE_RGRAD 3 / 1:
    // Pubmed step 5.610 -> 5.566 ms, Cora 0.9252 -> 0.9221; profiles/r06_glue.txt section 8).  3: the round-3 threshold.
    return R >= (knobs_of(ctx).node_rgrad >= 3 ? 49152 : 2048) || knobs_of(ctx).node_rgrad == 2;
}

int launch_rgrad(pn_context *ctx, void *stream_, const RgradParams &p) {
    hipStream_t stream = (hipStream_t)stream_;
    if (p.R <= 0 || p.M <= 0 || p.N <= 0) return PN_OK;
    // (N itself may fall short of a multiple of 4 when the rows of B are padded that far: the kernel reads whole quads of
    //  columns and stores the first N)
    if (p.M % 4 || p.lda % 4 || p.ldb % 4 || (p.N + 3) / 4 * 4 > p.ldb || p.M > p.lda)
        PN_FAIL(PN_ERR_ARG, "rgrad: M and the row pitches must be multiples of 4, the pitches cover M / N rounded up to 4");
    if ((p.seg == nullptr) != (p.list == nullptr)) PN_FAIL(PN_ERR_ARG, "rgrad: seg and list go together");
    const int tiles = ((p.M + R_BM - 1) / R_BM) * ((p.N + R_BN - 1) / R_BN);
    const int64_t ntiles = (p.R + R_KT - 1) / R_KT;
    int64_t nz = std::max<int64_t>(1, (512 + tiles - 1) / tiles);       // ~two workgroups per CU over all output tiles
    nz = std::min(nz, std::max<int64_t>(1, ntiles / 8));                // ... each walking at least eight K tiles
    const dim3 grid((p.N + R_BN - 1) / R_BN, (p.M + R_BM - 1) / R_BM, (unsigned)nz);
    const void *kern = p.gate ? (p.list ? (const void *)rgrad_kernel<true, true> : (const void *)rgrad_kernel<true, false>)
                              : (p.list ? (const void *)rgrad_kernel<false, true> : (const void *)rgrad_kernel<false, false>);
    if (int rc = ensure_dynamic_lds(ctx, kern, R_LDS_BYTES)) return rc;
    if (p.gate && p.list)
        hipLaunchKernelGGL((rgrad_kernel<true, true>), grid, dim3(R_THREADS), R_LDS_BYTES, stream, p);
    else if (p.gate)
        hipLaunchKernelGGL((rgrad_kernel<true, false>), grid, dim3(R_THREADS), R_LDS_BYTES, stream, p);
    else if (p.list)
        hipLaunchKernelGGL((rgrad_kernel<false, true>), grid, dim3(R_THREADS), R_LDS_BYTES, stream, p);
    else
        hipLaunchKernelGGL((rgrad_kernel<false, false>), grid, dim3(R_THREADS), R_LDS_BYTES, stream, p);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

}  // namespace pn
