// pn_seq4.hip -- the weight-gradient GEMM of the bf16 x 3 arithmetic (pn_pagg_shape.seq_math = bf16x3) at the headline shape
// (hidden size 128, four gate slots: nn.LSTM of PathNet / PathNet_homo, /root/reference/PathNet_run.py:164,195,265, and the GRU
// ablation): K tiles of 16 rows through two LDS stages.  The file also held a 128-path forward and a 64 / 128-path BPTT with BOTH
// operands of every k-step in LDS; they measured slower than the fused kernels in rounds 3 and 4 and were removed in round 6
// (the experiments and their numbers: profiles/HISTORY_r1_r4.md).
//
// Arithmetic (pn_kernels.h): fp32 products as six bf16 MFMAs over exact three-plane splits.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "pn_kernels.h"
#include "pn_seq.h"

using namespace pn;

namespace {

constexpr int H4 = 128, G4 = 4;      // hidden size and gate slots this file is built for

// =====================================================================================================================
// weight gradient:  [g_W_ih | g_W_hh] [G*H, 2H] = dG^T [G*H, R] . XH [R, 2H]   (R = P*L rows; colsum(dG) = bias gradient)
//   Same decomposition as wgrad3_kernel (pn_pagg.hip): 256 x 256 output tile per workgroup (8 waves, 64 x 128 each),
//   the R rows split over blockIdx.z in strided K tiles, partial tiles to part_w / part_b for wgrad_reduce_kernel.
//   What changes is the pipeline: K tiles of 16 rows, TWO LDS stages, and the loads of tile i+2 issued behind the
//   products of tile i, in flight across the barrier -- wgrad3 alternates "stage 32 rows" and "96 MFMAs per wave" in
//   lockstep with one buffer and had the matrix pipe busy 48 % of the time.
//   Staging task of a thread per tile: operand op, rows 4 rq .. +3 of the 16, columns 4 cq .. +3: four coalesced 16-byte
//   loads; per column the four rows are split into their bf16 planes (two packed pairs = 8 bytes per plane) and written
//   to the half (rq & 1) of the column's 16-byte k-octet slot -- the transposition to "8 consecutive k per lane" is free.
//   LDS image per (plane, operand, k-octet): 256 columns, column c at slot (c & 3) * 68 + (c >> 2)  (as wgrad3).
// =====================================================================================================================
constexpr int W4_BM = 256, W4_BN = 256, W4_KT = 16, W4_THREADS = 512;
constexpr int W4_BLK = 4 * 68;                              // 16-byte slots per (plane, operand, k-octet) block
constexpr int W4_PLANE = 2 * 2 * W4_BLK;                    // slots per plane: 2 operands x 2 k-octets
constexpr int W4_STAGE = 3 * W4_PLANE;                      // slots per stage (52 224 bytes)
constexpr int W4_LDS_BYTES = 2 * W4_STAGE * 16;             // 104 448

__global__ __launch_bounds__(W4_THREADS, 2) void wgrad4_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, hk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * W4_BM, n0 = blockIdx.x * W4_BN;
    // split z takes the K tiles z, z + nz, z + 2 nz, ...: every workgroup starts on the low rows, which the (reversed)
    // BPTT wrote last and which are still in the Infinity Cache
    const int64_t ntiles = (p.R + W4_KT - 1) / W4_KT;
    const int64_t nz = gridDim.z;
    const int64_t my_tiles = blockIdx.z < ntiles ? (ntiles - blockIdx.z + nz - 1) / nz : 0;
    if (my_tiles == 0) return;      // block-uniform
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int op = tid >> 8, rq = (tid >> 6) & 3, cq = tid & 63;
    const float *src = op == 0 ? p.dG : p.xh;
    const int ld = op == 0 ? p.GH : p.H2;
    const int c0 = (op == 0 ? m0 : n0) + 4 * cq;
    const bool c_ok = c0 < ld;
    const float *srcc = src + (c_ok ? c0 : 0);
    // two register sets: the rows of tile i + 1 are split and written to LDS BETWEEN the MFMA groups of tile i (the
    // matrix pipe works on a group for 256 cycles, the wave's VALU / LDS instructions issue in its shadow), while the
    // loads of tile i + 2 are in flight into the other set.  (A finer cut -- one register set, one (row pair, column)
    // between half groups of four MFMAs -- measured the same: 0.281-0.283 vs 0.280-0.286 ms.)
    f32x4 rgA[4], rgB[4];
    auto row0_of = [&](int64_t i) { return (blockIdx.z + min(i, my_tiles - 1) * nz) * W4_KT; };     // (clamped: harmless re-load)
    auto issue = [&](f32x4 (&rg)[4], int64_t i) {
        const int64_t k0 = row0_of(i);
#pragma unroll
        for (int e = 0; e < 4; e++)
            async_load_b128(rg[e], srcc + min(k0 + 4 * rq + e, p.R - 1) * ld);
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};     // column sums of dG over this thread's rows (bias gradient)
    // operand op, k-octet rq >> 1, column 4 cq + j at slot j * 68 + cq; this thread's rows are half (rq & 1) of the octet
    unsigned char *stage_wr = reinterpret_cast<unsigned char *>(lds4 + (op * 2 + (rq >> 1)) * W4_BLK + cq) + (rq & 1) * 8;
    // column j of the thread's 4 x 4 block of tile i -> the three planes of stage buf (tiles past the end: zeros)
    auto piece = [&](f32x4 (&rg)[4], int64_t i, int buf, int j) {
        const int64_t k0 = row0_of(i);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = (c_ok && i < my_tiles && k0 + 4 * rq + e < p.R) ? rg[e][j] : 0.0f;
        unsigned char *w = stage_wr + (size_t)buf * (W4_STAGE * 16);
        uint32_t x0, x1, x2, y0, y1, y2;
        split3(v[0], v[1], x0, x1, x2);
        split3(v[2], v[3], y0, y1, y2);
        bs[j] += (v[0] + v[1]) + (v[2] + v[3]);
        *reinterpret_cast<uint2 *>(w + j * 68 * 16) = make_uint2(x0, y0);
        *reinterpret_cast<uint2 *>(w + (W4_PLANE + j * 68) * 16) = make_uint2(x1, y1);
        *reinterpret_cast<uint2 *>(w + (2 * W4_PLANE + j * 68) * 16) = make_uint2(x2, y2);
    };
    const int sa = hk * W4_BLK + (li & 3) * 68 + (li >> 2) + wm * 16;                     // operand 0 (dG^T), k-octet hk
    const int sb = (2 + hk) * W4_BLK + (li & 3) * 68 + (li >> 2) + wn * 32;               // operand 1 ([x|h])
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
    // One step = the products of the tile in stage buf, the commit of tile inext (registers rg) to the other stage in the
    // gaps after the first four MFMA groups, ONE barrier, then the last two groups -- under which the first fragments of the
    // next tile are already fetched from the stage just completed, so that the next step starts on full registers instead
    // of an empty matrix pipe behind the barrier.  Program order is pinned by the fences; every LDS fragment read sits a
    // group ahead of its first use.  On entry A0 / B0 hold (or are receiving) the plane-0 fragments of this tile and B1 its
    // plane-1 b fragments; on exit the same holds for the next tile with the roles of B0 and B1 exchanged.
    u32x4 fA0[2], fA1[2], fB0[4], fB1[4];
    auto group = [&](const u32x4 (&a)[2], const u32x4 (&b)[4]) {
        W4_FENCE();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = mfma_bf16(a[i], b[j], acc[i][j]);
        W4_FENCE();
    };
    auto step = [&](int buf, u32x4 (&A0)[2], u32x4 (&A1)[2], u32x4 (&B0)[4], u32x4 (&B1)[4], f32x4 (&rg)[4], int64_t inext) {
        [[maybe_unused]] const int ts = ((int)inext - 1 - 8) * 5;       // (tuning builds stamp tiles 8 .. 8 + 101)
        const u32x4 *fa = lds4 + buf * W4_STAGE + sa, *fb = lds4 + buf * W4_STAGE + sb;
        const u32x4 *na = lds4 + (buf ^ 1) * W4_STAGE + sa, *nb = lds4 + (buf ^ 1) * W4_STAGE + sb;
        group(A0, B0);                                  // a0.b0
        piece(rg, inext, buf ^ 1, 0);
#pragma unroll
        for (int i = 0; i < 2; i++) A1[i] = fa[W4_PLANE + i * 8];
        group(A0, B1);                                  // a0.b1
        piece(rg, inext, buf ^ 1, 1);
        group(A1, B1);                                  // a1.b1
#pragma unroll
        for (int j = 0; j < 4; j++) B1[j] = fb[2 * W4_PLANE + j * 8];
        piece(rg, inext, buf ^ 1, 2);
        group(A1, B0);                                  // a1.b0
#pragma unroll
        for (int i = 0; i < 2; i++) A1[i] = fa[2 * W4_PLANE + i * 8];
        piece(rg, inext, buf ^ 1, 3);
        W4_FENCE();
        __syncthreads();        // stage buf ^ 1 is complete; every read of stage buf has been issued and has landed
        group(A0, B1);                                  // a0.b2
#pragma unroll
        for (int i = 0; i < 2; i++) A0[i] = na[i * 8];                          // next tile: a plane 0
#pragma unroll
        for (int j = 0; j < 4; j++) B1[j] = nb[j * 8];                          //            b plane 0 (the next step's B0)
        group(A1, B0);                                  // a2.b0
#pragma unroll
        for (int j = 0; j < 4; j++) B0[j] = nb[W4_PLANE + j * 8];               //            b plane 1 (the next step's B1)
    };

    issue(rgA, 0);
    wait_vm<0>(rgA[0], rgA[1], rgA[2], rgA[3]);
#pragma unroll
    for (int j = 0; j < 4; j++) piece(rgA, 0, 0, j);
    issue(rgB, 1);
    __syncthreads();
    {
        const u32x4 *fa = lds4 + sa, *fb = lds4 + sb;
#pragma unroll
        for (int i = 0; i < 2; i++) fA0[i] = fa[i * 8];
#pragma unroll
        for (int j = 0; j < 4; j++) fB0[j] = fb[j * 8];
#pragma unroll
        for (int j = 0; j < 4; j++) fB1[j] = fb[W4_PLANE + j * 8];
    }
    // two tiles per trip (register sets, stages and the b fragment arrays alternate); an odd count runs one tile of zeros
#pragma unroll 1
    for (int64_t i = 0; i < my_tiles; i += 2) {
        issue(rgA, i + 2);
        wait_vm<4>(rgB[0], rgB[1], rgB[2], rgB[3]);         // tile i + 1 has arrived (the four loads just issued may be out)
        step(0, fA0, fA1, fB0, fB1, rgB, i + 1);
        issue(rgB, i + 3);
        wait_vm<4>(rgA[0], rgA[1], rgA[2], rgA[3]);
        step(1, fA0, fA1, fB1, fB0, rgA, i + 2);
    }
    wait_vm<0>(rgA[0], rgA[1], rgA[2], rgA[3]);     // drain the trailing (clamped) loads
    wait_vm<0>(rgB[0], rgB[1], rgB[2], rgB[3]);
    __syncthreads();        // (the bias sums below reuse the stages)
#undef W4_FENCE
    float *pw = p.part_w + (int64_t)blockIdx.z * p.GH * p.H2;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + wn * 128 + j * 32 + li;
            if (n >= p.H2) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (m < p.GH) pw[(int64_t)m * p.H2 + n] = acc[i][j][r];
            }
        }
    // bias gradient: the four row-quad owners of a column add up through LDS
    if (blockIdx.x != 0) return;   // block-uniform
    float *fl = reinterpret_cast<float *>(lds4);
    if (op == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) fl[rq * W4_BM + 4 * cq + j] = bs[j];
    }
    __syncthreads();
    if (tid < W4_BM && m0 + tid < p.GH)
        p.part_b[(int64_t)blockIdx.z * p.GH + m0 + tid] =
            (fl[tid] + fl[W4_BM + tid]) + (fl[2 * W4_BM + tid] + fl[3 * W4_BM + tid]);
}

}  // namespace

namespace pn {

int seq4_select(const pn_context *ctx, int H, int G, int L) {
    if (H != H4 || G != G4 || L < 1 || L > 8) return 0;      // (LDS: the index arrays of a 128-path tile)
    // default: the weight-gradient GEMM of this file (commit of the next K tile between the MFMA groups of the current
    // one: 0.278 vs 0.297 ms, A/B in one session); its forward and BPTT measured slower than the fused kernels and stay
    // opt-in (PN_SEQ4 = bit mask, 0 = every fused kernel)
    return knobs_of(ctx).seq4 & SEQ4_WGRAD;
}

int launch_wgrad4(pn_context *ctx, void *stream, const WgradParams &wp, int nsplit) {
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(wgrad4_kernel), W4_LDS_BYTES)) return rc;
    hipLaunchKernelGGL(wgrad4_kernel, dim3((wp.H2 + W4_BN - 1) / W4_BN, (wp.GH + W4_BM - 1) / W4_BM, nsplit), dim3(W4_THREADS),
                       W4_LDS_BYTES, (hipStream_t)stream, wp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

}  // namespace pn
