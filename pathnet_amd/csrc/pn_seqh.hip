// pn_seqh.hip -- the three recurrent kernels of the aggregator on gfx950's fp16 matrix pipe: nn.LSTM / nn.RNN of
// /root/reference/PathNet_run.py:164,195,265 and baseline/GPRGNN/src/copy.py:308,349 (forward), the autograd backward
// of PathNet_run.py:351 through them (BPTT and weight gradient), for every hidden size up to 256.
//
// Same decomposition as the bf16 kernels of pn_pagg.hip (32-path tile per workgroup, wave w owns hidden units 32w..32w+31
// of all gates, [x_t | h_{t-1}] in LDS, weight fragments streamed L2 -> VGPR by asm loads) with the arithmetic of
// pn_kernels.h "two planes, three MFMAs": half the matrix instructions per product, 2/3 of the weight stream and of the
// LDS tile.  What that buys structurally:
//   forward  33 KB tile (52): same three workgroups per CU, 2/3 of the fragment stream and LDS traffic per step;
//   BPTT     the gate gradients of ALL four gates fit one tile (66 KB, two workgroups per CU): one k loop and two
//            barriers per step where the bf16 kernel runs two passes over gate pairs with four barriers;
//   wgrad    three MFMA groups per K tile instead of six, 70 KB of stages instead of 104.
// Operand scaling (fp16 has 5 exponent bits): pn_kernels.h.  The scales are powers of two derived in the kernels from
// maxima in device memory (SeqRange) -- weights: range_part_kernel + pack_fb_kernel; gathered rows: range_rows_kernel over the bank's output;
// the BPTT scales every tile of gate gradients by that tile's own maximum (computed in registers, exchanged through LDS
// at the barrier the step needs anyway) and leaves the launch's maximum for the weight-gradient GEMM.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "pn_kernels.h"
#include "pn_seq.h"

using namespace pn;

#ifndef PN_FWDH_WAVES
#define PN_FWDH_WAVES 3     // waves per SIMD the forward is compiled for at H <= 128 (= workgroups per CU at H = 128)
#endif
#ifndef PN_BWDH_WAVES
#define PN_BWDH_WAVES 2
#endif
// Settled by measurement in rounds 4-5 and no longer build options (the alternatives and their numbers: profiles/HISTORY_r1_r4.md,
// profiles/r05_tune_*.txt): one row block of 32 paths per wave in the forward (two: 0.253 vs 0.240 ms); the LSTM's saved values as
// ONE 16-byte quad {packed gates (3 dwords), c_{t-1}} per path step and unit; c carried in registers across the BPTT's steps; the next
// step's saved values requested ahead of the scatter; descending tile order in the BPTT; no atomic for elements dropout zeroed;
// gate gradients as [R][4][H] planes; no LDS-DMA "touch" prefetch, no early x gather, no XCD pairing of the weight gradient's
// column blocks (each measured neutral or slower).
#ifndef PN_TRACE_H
#define PN_TRACE_H 0        // 1: tuning builds only -- wave 0 of every workgroup stamps the cycle counter at phase boundaries
#endif
#if PN_TRACE_H
__device__ long long *g_trace_h = nullptr;      // [blocks][64] stamps, set with pn_debug_set_trace_h
#define HSTAMP(slot)                                                                                      \
    do {                                                                                                  \
        if (g_trace_h && threadIdx.x == 0 && (slot) < 64)                                                 \
            g_trace_h[(size_t)blockIdx.x * 64 + (slot)] = (long long)__builtin_readcyclecounter();        \
    } while (0)
// (the cycle counters are per CU and not aligned with each other: the dispatch timeline of a launch -- when workgroups
//  start and end relative to each other -- is taken from the 100 MHz device-wide counter, slots 62 / 63)
#define HSTAMP_RT(slot)                                                                                   \
    do {                                                                                                  \
        if (g_trace_h && threadIdx.x == 0)                                                                \
            g_trace_h[(size_t)blockIdx.x * 64 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime();    \
    } while (0)
#else
#define HSTAMP(slot) do { } while (0)
#define HSTAMP_RT(slot) do { } while (0)
#endif

namespace {

__device__ __forceinline__ uint32_t fbits_abs(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// ---- tile geometry ------------------------------------------------------------------------------------------------
// A launch of T = ceil(P / 32) tiles on S resident workgroup slots runs as ceil(T / S) synchronised rounds (every tile costs
// the same): at the headline shape 1624 tiles on 768 (forward) / 512 (BPTT) slots, and the last 20 % / 14 % of the launch
// keeps 143 / 78 workgroups in flight (profiles/r04_tail_experiments.txt).  A lone workgroup's step is bound by its memory
// phases and by the weight-fragment stream, not by the matrix pipe, so the remainder round is cut into SMALLER tiles --
// small_rows = 8, 16 or 24 paths, the accumulator registers of the missing rows are whole-wave dead (acc_row: rows < 8k are
// registers r < 4k) and their stores / atomics never issue -- one per CU instead of a 32-path tile on a third of the CUs.
// Tile j < n_big * (1 - small_first):  paths [32 j, 32 j + 32);  the small ones follow (forward) or come first (BPTT,
// whose dispatch order is descending: its remainder round is the START of the path range).
// MEASURED (round 5, profiles/r05_tune_scatter_tiling.txt): no gain -- forward 0.236 = 0.236 ms, BPTT 0.371 / 0.368 against
// 0.375 / 0.364 with 16-path tiles, slower with 8 (0.252 / 0.387): a lone workgroup's step is its latency chain and the
// 512 KB weight stream, neither of which shrinks with the rows.  Kept behind the context knob PN_SEQH_TAIL (default 0) with
// its parity tests (tests/test_gpu_seqh.py runs under forced sizes), as the evidence that a 16-path MFMA instantiation for
// the remainder round would not pay either.
__device__ __forceinline__ SeqTile seq_tile_of(const SeqTiling tg, int j, int MT, int P) {
    SeqTile t;
    if (tg.small_rows == 0) {
        t.q0 = j * MT;
        t.rows = min(MT, P - t.q0);
    } else if (tg.small_first) {
        if (j < tg.n_small) {
            t.q0 = j * tg.small_rows;
            t.rows = min(tg.small_rows, P - t.q0);
        } else {
            t.q0 = tg.n_small * tg.small_rows + (j - tg.n_small) * MT;
            t.rows = min(MT, P - t.q0);
        }
    } else {
        if (j < tg.n_big) {
            t.q0 = j * MT;
            t.rows = MT;
        } else {
            t.q0 = tg.n_big * MT + (j - tg.n_big) * tg.small_rows;
            t.rows = min(tg.small_rows, P - t.q0);
        }
    }
    t.q0 = __builtin_amdgcn_readfirstlane(t.q0);
    t.rows = __builtin_amdgcn_readfirstlane(t.rows);
    return t;
}

// host side: the split of P paths into n_big tiles of MT and n_small of small_rows for a kernel with `slots` resident
// workgroups on `cus` compute units.  Pure arithmetic (tests/test_abi.py drives it through pn_debug_seq_tiling).
void seq_tiling_for(int64_t P, int MT, int slots, int cus, int mode, bool small_first, SeqTiling *tg, int *blocks) {
    *tg = SeqTiling{0, 0, 0, small_first ? 1 : 0};
    const int64_t T = (P + MT - 1) / MT;
    *blocks = (int)T;
    if (mode == 0 || MT != 32 || slots <= 0 || cus <= 0 || P <= 0) return;
    const int64_t full = T / slots * slots;             // tiles of the full rounds
    const int64_t rem_paths = P - full * MT;            // what the remainder round holds
    if (rem_paths <= 0) return;
    int rows = mode >= 8 ? mode : (int)((rem_paths + (int64_t)cus * 8 - 1) / ((int64_t)cus * 8)) * 8;      // one tile per CU
    if (rows >= MT) return;                             // the remainder round fills the CUs as it is
    const int64_t n_small = (rem_paths + rows - 1) / rows;
    if (small_first) {
        const int64_t rest = P - n_small * rows;
        tg->n_small = (int)n_small;
        tg->n_big = (int)(rest > 0 ? (rest + MT - 1) / MT : 0);
    } else {
        tg->n_big = (int)full;
        tg->n_small = (int)n_small;
    }
    tg->small_rows = rows;
    *blocks = tg->n_big + tg->n_small;
}

int plan_tiling(pn_context *ctx, const void *kernel, int threads, size_t lds_bytes, int64_t P, int MT, bool small_first,
                SeqTiling *tg, int *blocks) {
    int slots = 0, cus = 0;
    const int mode = knobs_of(ctx).seqh_tail;
    if (mode != 0)
        if (int rc = resident_slots(ctx, kernel, threads, lds_bytes, &slots, &cus)) return rc;
    seq_tiling_for(P, MT, slots, cus, mode, small_first, tg, blocks);
    return PN_OK;
}

// ---- operand ranges ---------------------------------------------------------------------------------------------------
// max |W_ih|, max |W_hh| in two steps without an atomic or a slot to clear: RANGE_PARTS workgroups leave the maxima of their
// slices in part[0 .. PARTS) / part[PARTS .. 2 PARTS) (the workspace's `wmax`), the packing launch that follows on the same
// stream takes the maximum of those in every workgroup (64 L2-resident words) and its first workgroup stores range->w_ih /
// w_hh for the recurrent kernels.  (Until round 6 ONE workgroup read both tensors: 10 us of one CU's bandwidth on the
// chain the recurrence waits for.)  The first workgroup also clears the slots the atomicMax launches of this call add to --
// x unless the call keeps the bank of an earlier one, and dg -- so that a step issues no memset launch of its own.
constexpr int RANGE_PARTS = 32;
__global__ __launch_bounds__(256) void range_part_kernel(const float *__restrict__ w_ih, const float *__restrict__ w_hh, int64_t n4,
                                                         int clear_x, SeqRange *__restrict__ range, uint32_t *__restrict__ part) {
    __shared__ float red[2][4];
    const int64_t per = (n4 + RANGE_PARTS - 1) / RANGE_PARTS, lo = (int64_t)blockIdx.x * per, hi = min(n4, lo + per);
    float m0 = 0.0f, m1 = 0.0f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float4 a = reinterpret_cast<const float4 *>(w_ih)[i], b = reinterpret_cast<const float4 *>(w_hh)[i];
        m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
    }
    m0 = wave_max(m0);
    m1 = wave_max(m1);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = m0, red[1][threadIdx.x >> 6] = m1;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = __float_as_uint(fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])));
        part[RANGE_PARTS + blockIdx.x] = __float_as_uint(fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])));
        if (blockIdx.x == 0) {
            range->dg = 0u;
            if (clear_x) {
                range->x = 0u;
                range->x_esum = 0;
                range->x_cnt = 0u;
            }
        }
    }
}
// the two maxima from the partials, wave-uniform (every wave of the packing launch)
__device__ __forceinline__ void range_from_parts(const uint32_t *__restrict__ part, float &m_ih, float &m_hh) {
    const int lane = threadIdx.x & 63;
    const float v = __uint_as_float(part[lane]);        // (RANGE_PARTS = 32: lanes 0..31 hold w_ih's, 32..63 w_hh's partial maxima)
    static_assert(2 * RANGE_PARTS == 64, "one partial per lane");
    float a = lane < RANGE_PARTS ? v : 0.0f, b = lane < RANGE_PARTS ? 0.0f : v;
    m_ih = wave_max(a);
    m_hh = wave_max(b);
}


__global__ __launch_bounds__(256) void range_rows_kernel(const float *__restrict__ rows, int64_t nrows, int H4,
                                                         const int32_t *__restrict__ count, SeqRange *__restrict__ range) {
    const int64_t n = (count ? min((int64_t)*count, nrows) : nrows) * H4;
    const int lane = threadIdx.x & 63;
    const bool sampled = (blockIdx.x & 7u) == 0u;       // every eighth workgroup contributes to the spread statistic
    float m = 0.0f;
    int esum = 0, ecnt = 0;
    // a wave covers 256 consecutive values per trip (a tile of the statistic); the loop bound is wave-uniform
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); i0 < n; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + lane;
        float t = 0.0f;
        if (i < n) {
            const float4 a = reinterpret_cast<const float4 *>(rows)[i];
            t = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
        }
        m = fmaxf(m, t);
        if (sampled) {
            const float tm = wave_max(t);
            if (tm > 0.0f) {
                esum += (int)((__float_as_uint(tm) >> 23) & 0xffu);
                ecnt += 1;
            }
        }
    }
    // one atomic per WORKGROUP: thousands of atomicMax on one address serialise (2704 of them took 33 us, 676 took 11)
    __shared__ float red[4];
    __shared__ int reds[8];
    m = wave_max(m);
    if (lane == 0) {
        red[threadIdx.x >> 6] = m;
        reds[threadIdx.x >> 6] = esum;
        reds[4 + (threadIdx.x >> 6)] = ecnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.0f) atomicMax(&range->x, __float_as_uint(m));
        const int c = reds[4] + reds[5] + reds[6] + reds[7];
        if (sampled && c > 0) {
            atomicAdd(&range->x_esum, reds[0] + reds[1] + reds[2] + reds[3]);
            atomicAdd(&range->x_cnt, (uint32_t)c);
        }
    }
}

// scales of the forward GEMM  acc = S (b + x . W_ih^T + h . W_hh^T):  x s_x in fp16 times W_ih 2^e_ih, h s_h times W_hh 2^e_hh
// with s_x 2^e_ih = s_h 2^e_hh = S.  The weights sit at the top of the range (their scales are fixed when they are
// packed); of the two products the one that would overflow first decides S, the other side's activations are scaled lower.
struct FwdScales {
    float s_x, s_h, S, inv_S;
};
__device__ __forceinline__ FwdScales fwd_scales(const SeqRange *rg, float xmul) {
    const int e_ih = scale_exp(__uint_as_float(rg->w_ih)), e_hh = scale_exp(__uint_as_float(rg->w_hh));
    const int e_x = scale_exp(__uint_as_float(rg->x) * xmul), e_h = 14;       // |h| <= 1
    int ES = min(e_x + e_ih, e_h + e_hh);
    ES = ES > 120 ? 120 : ES < -120 ? -120 : ES;
    return FwdScales{exp2i(ES - e_ih), exp2i(ES - e_hh), exp2i(ES), exp2i(-ES)};
}


__device__ __forceinline__ int gru_weight_row(int slot, int j, int H) { return (slot < 2 ? slot : 2) * H + j; }

// =====================================================================================================================
// forward recurrence
//   Weights: pack_fwdh_body (pack_fb_kernel), B fragments of v_mfma_f32_32x32x16_f16,
//     Wp[(((w*KS + s)*2 + plane)*G + g)*64 + lane] (16 bytes) =
//         plane of 2^e Wcat[g*H + 32w + (lane & 31)][16 s + 8 (lane >> 5) .. +7],   KS = 2H/16 k-steps, e = e_ih | e_hh
//   A operand: LDS holds the two planes of the tile [32][x_t | h_{t-1}], row pitch 4H + 16 bytes.
//   GRU on the LSTM's four gate slots as in pn_pagg.hip (pack_fwd3_kernel).
// =====================================================================================================================
__device__ __forceinline__ void pack_fwdh_body(int idx, const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                               const float *__restrict__ b_ih, const float *__restrict__ b_hh, int H, int G, int gru,
                                               float m_ih, float m_hh, u32x4 *__restrict__ Wp, float *__restrict__ biasc) {
    if (idx < G * H) {
        if (!gru) {
            biasc[idx] = b_ih[idx] + b_hh[idx];
        } else {
            const int slot = idx / H, j = idx - slot * H, wr = gru_weight_row(slot, j, H);
            biasc[idx] = slot < 2 ? b_ih[wr] + b_hh[wr] : slot == 2 ? b_ih[wr] : b_hh[wr];
        }
    }
    const int KS = H / 8, NW = H / 32;
    if (idx >= NW * KS * G * 64) return;
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int g = rest % G;
    rest /= G;
    const int s = rest % KS, w = rest / KS;
    const int j = 32 * w + (lane & 31), k = 16 * s + 8 * (lane >> 5);
    const int row = gru ? gru_weight_row(g, j, H) : g * H + j;
    const float *src = k < H ? w_ih + (int64_t)row * H + k : w_hh + (int64_t)row * H + (k - H);
    float4 v0 = reinterpret_cast<const float4 *>(src)[0], v1 = reinterpret_cast<const float4 *>(src)[1];
    if (gru && ((g == 2 && k >= H) || (g == 3 && k < H))) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float sc = exp2i(scale_exp(k < H ? m_ih : m_hh));
    u32x4 q0, q1;
    uint32_t x0, x1;
    split2h(v0.x * sc, v0.y * sc, x0, x1); q0[0] = x0; q1[0] = x1;
    split2h(v0.z * sc, v0.w * sc, x0, x1); q0[1] = x0; q1[1] = x1;
    split2h(v1.x * sc, v1.y * sc, x0, x1); q0[2] = x0; q1[2] = x1;
    split2h(v1.z * sc, v1.w * sc, x0, x1); q0[3] = x0; q1[3] = x1;
    u32x4 *dst = Wp + ((int64_t)(w * KS + s) * 2 * G + g) * 64 + lane;
    dst[0] = q0;
    dst[G * 64] = q1;
}

// RB: row blocks of 32 paths per wave.  RB = 2: a wave keeps two accumulator sets and every weight fragment feeds two MFMAs --
// the fragment stream per path, which bounds the k loop (35 B/clk/CU through the vector-memory return path: 14.7 k cycles
// per step where the matrix pipe needs 6.1 k, profiles/r04_phase_trace.txt), halves.  With two fp16 planes the 64-row tile
// is 68 KB: two workgroups per CU (the bf16 form of this needed 101 KB, one workgroup per CU, and was slower).
template <int H, int RB>
constexpr int fwdh_waves() { return H == 32 && RB == 1 ? 1 : (H > 128 || RB > 1) ? 2 : PN_FWDH_WAVES; }

// GC: 4 = LSTM, 1 = tanh RNN, 3 = GRU
template <int H, int GC, int RB>
__global__ __launch_bounds__(H / 32 * 64, (fwdh_waves<H, RB>())) void seq_fwdh_kernel(SeqFwdParams p) {
    constexpr int G = GC == 3 ? 4 : GC;
    constexpr bool GRU = GC == 3;
    constexpr int MT = 32 * RB;
    constexpr bool PACKED = GC == 4;          // LSTM: {gates as 3 dwords (pack_gates), c_{t-1}}: 4 H dwords per path step
    constexpr int NW = H / 32, NT = NW * 64, SV = PACKED ? 4 : (G == 4 ? 5 : 1);
    constexpr int KS = H / 8, KX = KS / 2;    // k-steps of 16 over [x | h]; the first KX walk x
    constexpr int PB = 4 * H + 16;            // row pitch of a plane of the tile [x | h], bytes: conflict-free ds_read_b128
    constexpr int PLANE = MT * PB;
    // three workgroups per CU: no register room for the x_{t+1} rows or a second hi-plane fragment set, the third
    // workgroup covers those latencies instead (as in seq_fwd3_kernel)
    constexpr bool PREFETCH_X = RB == 1 && fwdh_waves<H, RB>() < 3, PING_PONG = fwdh_waves<H, RB>() < 3 && RB == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    int *s_rowidx = reinterpret_cast<int *>(ldsb + 2 * PLANE);  // [MT][L] gather rows of this tile
    int *s_slotof = s_rowidx + MT * p.L;                        // [MT]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31;
    // tile geometry (SeqTiling): the first n_big workgroups take MT paths each, the rest -- the launch's remainder round --
    // small_rows each, so that the last round runs on every CU instead of a third of them
    const SeqTile tl = seq_tile_of(p.tiling, (int)blockIdx.x, MT, p.P);
    const int q0 = tl.q0, rows_here = tl.rows;
    const int col = 32 * wave + li;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);    // the wave's column slice as a scalar (weight stream base)

    for (int i = tid; i < MT * p.L; i += NT) s_rowidx[i] = i / p.L < rows_here ? p.rowidx[(int64_t)q0 * p.L + i] : 0;
    for (int i = tid; i < MT; i += NT) s_slotof[i] = i < rows_here ? p.slotof[q0 + i] : 0;

    const FwdScales sc = fwd_scales(p.range, p.xmul);
    f32x16 cst[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) cst[rb][r] = 0.0f;
    const float keep_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (1.0f - p.p_drop))));
    const bool builtin_drop = !p.mask && p.p_drop > 0.0f;
    const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
    const size_t tile_row = (size_t)q0 * (size_t)p.L;                       // first [P, L] row of this tile
    uint8_t *keep_t = p.keep ? p.keep + tile_row * (H / 4) : nullptr;
    float4 *xh4_t = p.xh ? reinterpret_cast<float4 *>(p.xh) + tile_row * (2 * H / 4) : nullptr;
    float *xh_t = p.xh ? p.xh + tile_row * (2 * H) : nullptr;
    float *saved_t = p.saved ? p.saved + tile_row * (SV * H) : nullptr;
    float *hn_t = p.hn + (size_t)q0 * H;
    __syncthreads();

    // ---- coalesced row gather of x_{t+1} (H*4 bytes per row), dropout keep bits drawn behind the loads, mask and
    //      scale applied when the rows are committed to LDS
    constexpr int NLD = 4 * RB;   // float4 per thread = MT * (H/4) / NT
    f32x4 xr[NLD];
    uint32_t keepbits = 0;        // 4 bits per row of this thread
    int tid_g = tid;              // re-derived per step (fresh_lane): offsets derived from it must not live across the MFMA loop
    auto gather_issue = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = tid_g + NT * i;
            const int row = idx / (H / 4), c4 = idx - row * (H / 4);
            async_load_b128(xr[i], p.Z + ((size_t)(uint32_t)s_rowidx[row * p.L + t] * (H / 4) + c4) * 4);
        }
        uint32_t bits = 0;
        if (builtin_drop) {
#pragma unroll
            for (int i = 0; i < NLD; i++) {
                const int idx = tid_g + NT * i;
                const int row = idx / (H / 4), c4 = idx - row * (H / 4);
                const float4 m = dropout4(seed, ((uint64_t)t * p.Pmask + s_slotof[row]) * (H / 4) + c4, 1u, p.p_drop);
                bits |= ((m.x != 0.f ? 1u : 0u) | (m.y != 0.f ? 2u : 0u) | (m.z != 0.f ? 4u : 0u) |
                         (m.w != 0.f ? 8u : 0u)) << (4 * i);
            }
        }
        asm volatile("" : "+v"(bits));      // drawn here, not sunk to the commit
        keepbits = bits;
    };
    auto gather_commit = [&](int t) {
        if constexpr (RB == 1)
            wait_vm<0>(xr[0], xr[1], xr[2], xr[3]);
        else
            wait_vm<0>(xr[0], xr[1], xr[2], xr[3], xr[4 % NLD], xr[5 % NLD], xr[6 % NLD], xr[7 % NLD]);
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = tid_g + NT * i;
            const int row = idx / (H / 4), c4 = idx - row * (H / 4);
            const bool live = row < rows_here;
            float4 v = make_float4(xr[i][0], xr[i][1], xr[i][2], xr[i][3]);
            if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.mask) {
                if (live) {
                    const float4 m = reinterpret_cast<const float4 *>(
                        p.mask)[((int64_t)t * p.Pmask + s_slotof[row]) * (H / 4) + c4];
                    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
                }
            } else if (builtin_drop) {
                const uint32_t b = keepbits >> (4 * i);
                v.x = b & 1u ? v.x * keep_scale : 0.0f;
                v.y = b & 2u ? v.y * keep_scale : 0.0f;
                v.z = b & 4u ? v.z * keep_scale : 0.0f;
                v.w = b & 8u ? v.w * keep_scale : 0.0f;
                if (keep_t && live) keep_t[((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)(H / 4) + c4] = (uint8_t)(b & 15u);
            }
            uint32_t a0, a1, b0, b1;
            split2h(v.x * sc.s_x, v.y * sc.s_x, a0, a1);
            split2h(v.z * sc.s_x, v.w * sc.s_x, b0, b1);
            unsigned char *d = ldsb + row * PB + 8 * c4;
            *reinterpret_cast<uint2 *>(d) = make_uint2(a0, b0);
            *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(a1, b1);
            if (xh4_t && live) {
                float4 *xo = &at_bytes(xh4_t, (((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)(2 * H / 4) + c4) * 16u);
                xo[0] = v;
                if (t == 0) xo[H / 4] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    gather_issue(0);
    gather_commit(0);
    __syncthreads();

    for (int t = 0; t < p.L; t++) {
        if (t == 0) HSTAMP_RT(62);
        HSTAMP(4 * t + 0);
        tid_g = wave_u * 64 + fresh_lane();
        if (PREFETCH_X && t + 1 < p.L) gather_issue(t + 1);

        f32x16 acc[RB][G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float bias = p.biasc[g * H + col] * sc.S;
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[rb][g][r] = bias;
        }

        // ---- [x_t ; h_{t-1}] x [W_ih ; W_hh]^T.  Per k-step the products run  a_lo.B_hi, a_hi.B_hi | a_hi.B_lo : the lo
        //      plane of the A tile is dead after the first G MFMAs and re-read for k-step s+1 right there, the hi plane
        //      at the end; the weight planes stream L2 -> VGPR one k-step (hi, ping-pong) / two thirds of one (lo) ahead.
        //      Step 0 has h_{-1} = 0: it stops after the x half of K.
        {
            const int nsteps = t == 0 ? KX : KS;
            const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.Wp) + (size_t)wave_u * (KS * 2 * G * 1024);
            const int lane_k = fresh_lane();
            const uint32_t voff = lane_k * 16;
            const unsigned char *arow = ldsb + (lane_k & 31) * PB + 16 * (lane_k >> 5);
            u32x4 Bha[G], Bhb[G], Bl[G];
            auto load = [&](u32x4 (&B)[G], int s, int pl) {
                async_load_frags<G>(B, wb + (size_t)(s * 2 + pl) * (G * 1024), voff);
            };
            u32x4 a[RB][2];
            auto areads = [&](int s, int pl) {
#pragma unroll
                for (int rb = 0; rb < RB; rb++)
                    a[rb][pl] = *reinterpret_cast<const u32x4 *>(arow + rb * 32 * PB + 32 * s + pl * PLANE);
            };
            // all RB row blocks of a product before the next product: every weight fragment feeds RB MFMAs
            auto prod = [&](int pa, u32x4 (&B)[G]) {
#pragma unroll
                for (int rb = 0; rb < RB; rb++)
#pragma unroll
                    for (int g = 0; g < G; g++) acc[rb][g] = mfma_f16(a[rb][pa], B[g], acc[rb][g]);
            };
            // vmcnt (in order) at the top of k-step s: PING_PONG [Bh(s) Bl(s)] + the Bh(s+1) just issued; else [Bh(s) Bl(s)]
            auto kstep = [&](int s, u32x4 (&Bh)[G], u32x4 (&Bhnext)[G]) {
                const int sn = min(s + 1, nsteps - 1);
                if (PING_PONG) load(Bhnext, sn, 0);
                wait_frag<(PING_PONG ? 2 : 1) * G, G>(Bh);
                prod(1, Bh);
                areads(sn, 1);
                prod(0, Bh);
                if (!PING_PONG) load(Bh, sn, 0);
                wait_frag<G, G>(Bl);
                prod(0, Bl);
                areads(sn, 0);
                load(Bl, sn, 1);
            };
            areads(0, 0);
            areads(0, 1);
            load(Bha, 0, 0);
            load(Bl, 0, 1);
#pragma unroll 1
            for (int s = 0; s < nsteps; s += 2) {
                if (PING_PONG) {
                    kstep(s, Bha, Bhb);
                    kstep(s + 1, Bhb, Bha);
                } else {
                    kstep(s, Bha, Bha);
                    kstep(s + 1, Bha, Bha);
                }
            }
            wait_frag<0, G>(Bha);                         // drain (harmless re-loads of the last k-step)
            wait_frag<0, G>(Bl);
        }
        HSTAMP(4 * t + 1);
        __syncthreads();  // every wave is done reading x_t / h_{t-1}
        HSTAMP(4 * t + 2);

        // ---- cell update in registers; h_t goes back to LDS (scaled, split) for the next step ----------------------
        const int lane_o = fresh_lane();    // row offsets are re-derived in every step: hoisted out of the t loop they spill
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = 32 * rb + acc_row(r, lane_o);
            const bool live = row < rows_here;
            float h;
            if (GRU) {
                // saved: r, z, n, the pre-activation W_hn h + b_hn, h_{t-1}
                const float rg = sigmoidf_(acc[rb][0][r] * sc.inv_S);
                const float zg = sigmoidf_(acc[rb][G > 1 ? 1 : 0][r] * sc.inv_S);
                const float nh = acc[rb][G > 3 ? 3 : 0][r] * sc.inv_S;
                const float ng = tanhf_(acc[rb][G > 2 ? 2 : 0][r] * sc.inv_S + rg * nh);
                const float hp = cst[rb][r];
                h = (1.0f - zg) * ng + zg * hp;
                cst[rb][r] = h;
                if (saved_t && live) {
                    float *sv = &at_bytes(saved_t, (((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)(SV * H) + col) * 4u);
                    sv[0] = rg; sv[H] = zg; sv[2 * H] = ng; sv[3 * H] = nh; sv[4 * H] = hp;
                }
            } else if (G == 4) {
                const float ig = sigmoidf_(acc[rb][0][r] * sc.inv_S);
                const float fg = sigmoidf_(acc[rb][G > 1 ? 1 : 0][r] * sc.inv_S);
                const float gg = tanhf_(acc[rb][G > 2 ? 2 : 0][r] * sc.inv_S);
                const float og = sigmoidf_(acc[rb][G > 3 ? 3 : 0][r] * sc.inv_S);
                const float cprev = cst[rb][r];
                const float c = fg * cprev + ig * gg;
                cst[rb][r] = c;
                h = og * tanhf_(c);
                if (saved_t && live) {
                    const uint32_t base = ((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)(SV * H);
                    // what the BPTT needs of this path step and unit as ONE 16-byte quad: {packed gates (3 dwords), c_{t-1}}
                    const uint3 pk = pack_gates(ig, fg, gg, og);
                    *reinterpret_cast<uint4 *>(&at_bytes(saved_t, (base + 4u * col) * 4u)) =
                        make_uint4(pk.x, pk.y, pk.z, __float_as_uint(cprev));
                }
            } else {
                h = tanhf_(acc[rb][0][r] * sc.inv_S);
                if (saved_t && live) at_bytes(saved_t, (((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)H + col) * 4u) = h;
            }
            hv[r] = h;
            if (live) {
                if (t == p.L - 1)
                    at_bytes(hn_t, ((uint32_t)row * (uint32_t)H + col) * 4u) = h;
                else if (xh_t)
                    at_bytes(xh_t, (((uint32_t)row * (uint32_t)p.L + t + 1) * (uint32_t)(2 * H) + H + col) * 4u) = h;
            }
        }
        if (t + 1 < p.L) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {       // accumulator registers r, r+1 are tile rows row, row+1
                uint32_t h0, h1;
                split2h(hv[r] * sc.s_h, hv[r + 1] * sc.s_h, h0, h1);
                unsigned char *d = ldsb + (32 * rb + acc_row(r, lane_o)) * PB + 2 * (H + col);
                *reinterpret_cast<uint16_t *>(d) = (uint16_t)h0;
                *reinterpret_cast<uint16_t *>(d + PB) = (uint16_t)(h0 >> 16);
                *reinterpret_cast<uint16_t *>(d + PLANE) = (uint16_t)h1;
                *reinterpret_cast<uint16_t *>(d + PLANE + PB) = (uint16_t)(h1 >> 16);
            }
        }
        }
        if (t + 1 < p.L) {
            tid_g = wave_u * 64 + fresh_lane();
            if (!PREFETCH_X) gather_issue(t + 1);
            gather_commit(t + 1);     // (every wave is past its reads of x_t)
            __syncthreads();
        }
        HSTAMP(4 * t + 3);
    }
    HSTAMP_RT(63);
}

// =====================================================================================================================
// Inference forward with the input half of the gates applied BEFORE the gather.  Without dropout x_t is the gathered bank
// row itself, so W_ih x_t + b depends on (node, code) only -- the argument that put the distance bank before the gather
//  applies once more: ZW[row] = Z[row] . W_ih^T + b is ONE GEMM over the rows of the bank (the caller:
// pn_pagg.hip, run_tables), a path step gathers its G*H pre-activations straight into the accumulator layout, and the
// recurrence keeps only the W_hh half of its products -- half the MFMAs and half the weight-fragment stream per step, no x
// tile in LDS, no step-0 products at all.  The validation and test forwards of an epoch (PathNet_run.py:359-362, :378) are
// two of its three forwards.
//   acc = S ZW[row] + (s_h h) . (2^e_hh W_hh)^T   with S = s_h 2^e_hh as in seq_fwdh_kernel; the weights are the same
//   packed planes, of which only the k-steps KX .. KS-1 (the h half) are read.
// =====================================================================================================================
template <int H, int GC>
__global__ __launch_bounds__(H / 32 * 64, (fwdh_waves<H, 1>())) void seq_fwdzw_kernel(SeqFwdParams p) {
    constexpr int G = GC == 3 ? 4 : GC;
    constexpr bool GRU = GC == 3;
    constexpr int MT = 32, NW = H / 32, NT = NW * 64, GH = G * H;
    constexpr int KS = H / 8, KX = KS / 2, KH = KS - KX;      // k-steps of 16 over [x | h]: the last KH walk h
    constexpr int PB = 2 * H + 16, PLANE = MT * PB;           // the tile holds h_{t-1} only
    constexpr bool PING_PONG = fwdh_waves<H, 1>() < 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    int *s_rowidx = reinterpret_cast<int *>(ldsb + 2 * PLANE);  // [MT][L] gather rows of this tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31;
    const int q0 = blockIdx.x * MT;
    const int col = 32 * wave + li;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int i = tid; i < MT * p.L; i += NT) s_rowidx[i] = q0 + i / p.L < p.P ? p.rowidx[(int64_t)q0 * p.L + i] : 0;
    const FwdScales sc = fwd_scales(p.range, 1.0f);
    f32x16 cst;
#pragma unroll
    for (int r = 0; r < 16; r++) cst[r] = 0.0f;
    float *hn_t = p.hn + (size_t)q0 * H;
    __syncthreads();

    for (int t = 0; t < p.L; t++) {
        const int lane_k = fresh_lane();
        // ---- the gathered pre-activations, in accumulator layout (rows past P read the table's row 0)
        f32x16 acc[G];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float *zw = p.ZW + (size_t)(uint32_t)s_rowidx[acc_row(r, lane_k) * p.L + t] * GH + col;
#pragma unroll
            for (int g = 0; g < G; g++) acc[g][r] = zw[g * H] * sc.S;
        }
        if (t > 0) {
            const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.Wp) + (size_t)wave_u * (KS * 2 * G * 1024) +
                                      (size_t)KX * 2 * (G * 1024);
            const uint32_t voff = lane_k * 16;
            const unsigned char *arow = ldsb + (lane_k & 31) * PB + 16 * (lane_k >> 5);
            u32x4 Bha[G], Bhb[G], Bl[G];
            auto load = [&](u32x4 (&B)[G], int s, int pl) {
                async_load_frags<G>(B, wb + (size_t)(s * 2 + pl) * (G * 1024), voff);
            };
            u32x4 a[2];
            auto aread = [&](int s, int pl) { return *reinterpret_cast<const u32x4 *>(arow + 32 * s + pl * PLANE); };
            auto prod = [&](int pa, u32x4 (&B)[G]) {
#pragma unroll
                for (int g = 0; g < G; g++) acc[g] = mfma_f16(a[pa], B[g], acc[g]);
            };
            auto kstep = [&](int s, u32x4 (&Bh)[G], u32x4 (&Bhnext)[G]) {       // (the pipeline of seq_fwdh_kernel)
                const int sn = min(s + 1, KH - 1);
                if (PING_PONG) load(Bhnext, sn, 0);
                wait_frag<(PING_PONG ? 2 : 1) * G, G>(Bh);
                prod(1, Bh);
                a[1] = aread(sn, 1);
                prod(0, Bh);
                if (!PING_PONG) load(Bh, sn, 0);
                wait_frag<G, G>(Bl);
                prod(0, Bl);
                a[0] = aread(sn, 0);
                load(Bl, sn, 1);
            };
            a[0] = aread(0, 0);
            a[1] = aread(0, 1);
            load(Bha, 0, 0);
            load(Bl, 0, 1);
#pragma unroll 1
            for (int s = 0; s < KH; s += 2) {
                if (PING_PONG) {
                    kstep(s, Bha, Bhb);
                    if (KH % 2 == 0 || s + 1 < KH) kstep(s + 1, Bhb, Bha);
                } else {
                    kstep(s, Bha, Bha);
                    if (KH % 2 == 0 || s + 1 < KH) kstep(s + 1, Bha, Bha);
                }
            }
            wait_frag<0, G>(Bha);
            if (PING_PONG) wait_frag<0, G>(Bhb);
            wait_frag<0, G>(Bl);
            __syncthreads();  // every wave is done reading h_{t-1}
        }
        const int lane_o = fresh_lane();
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float h;
            if (GRU) {
                const float rg = sigmoidf_(acc[0][r] * sc.inv_S);
                const float zg = sigmoidf_(acc[G > 1 ? 1 : 0][r] * sc.inv_S);
                const float nh = acc[G > 3 ? 3 : 0][r] * sc.inv_S;
                const float ng = tanhf_(acc[G > 2 ? 2 : 0][r] * sc.inv_S + rg * nh);
                h = (1.0f - zg) * ng + zg * cst[r];
                cst[r] = h;
            } else if (G == 4) {
                const float ig = sigmoidf_(acc[0][r] * sc.inv_S);
                const float fg = sigmoidf_(acc[G > 1 ? 1 : 0][r] * sc.inv_S);
                const float gg = tanhf_(acc[G > 2 ? 2 : 0][r] * sc.inv_S);
                const float og = sigmoidf_(acc[G > 3 ? 3 : 0][r] * sc.inv_S);
                const float c = fg * cst[r] + ig * gg;
                cst[r] = c;
                h = og * tanhf_(c);
            } else {
                h = tanhf_(acc[0][r] * sc.inv_S);
            }
            hv[r] = h;
            const int row = acc_row(r, lane_o);
            if (t == p.L - 1 && q0 + row < p.P) at_bytes(hn_t, ((uint32_t)row * (uint32_t)H + col) * 4u) = h;
        }
        if (t + 1 < p.L) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                uint32_t h0, h1;
                split2h(hv[r] * sc.s_h, hv[r + 1] * sc.s_h, h0, h1);
                unsigned char *d = ldsb + acc_row(r, lane_o) * PB + 2 * col;
                *reinterpret_cast<uint16_t *>(d) = (uint16_t)h0;
                *reinterpret_cast<uint16_t *>(d + PB) = (uint16_t)(h0 >> 16);
                *reinterpret_cast<uint16_t *>(d + PLANE) = (uint16_t)h1;
                *reinterpret_cast<uint16_t *>(d + PLANE + PB) = (uint16_t)(h1 >> 16);
            }
            __syncthreads();
        }
    }
}

// =====================================================================================================================
// BPTT:  [dx_t | dh_{t-1}] = dG_t [32, G*H] . [W_ih | W_hh],  K = G*H gate columns in ONE pass (all gates resident)
//   Weights: pack_bwdh_body (pack_fb_kernel), B fragments in units of two k-steps (kk) x two output halves (nt),
//     WpT[(((w*NU + u)*2 + plane)*4 + kk*2 + nt)*64 + lane] (16 bytes) =
//         plane of 2^e Wcat[k = 32u + 16kk + 8(lane >> 5) .. +7][n = nt*H + 32w + (lane & 31)],   NU = G*H/32, e = e_ih | e_hh by nt
//   A operand: the two fp16 planes of s_g dG_t, s_g = the power of two that puts THIS tile's largest |dG| into [2^14, 2^15).
//   Per step: cell backward into registers + wave maxima to LDS | barrier | scale, split, planes to LDS | barrier | k loop |
//   scatter -- the next step's cell backward touches no LDS the k loop reads, so there is no third barrier.
// =====================================================================================================================
__device__ __forceinline__ void pack_bwdh_body(int idx, const float *__restrict__ w_ih, const float *__restrict__ w_hh, int H, int G,
                                               int gru, float m_ih, float m_hh, u32x4 *__restrict__ WpT) {
    const int GH = G * H, NU = GH / 32, NW = H / 32;
    if (idx >= NW * NU * 4 * 64) return;
    const int lane = idx & 63, f = (idx >> 6) & 3;
    int rest = idx >> 8;
    const int u = rest % NU, w = rest / NU;
    const int kk = f >> 1, nt = f & 1;
    const int k = 32 * u + 16 * kk + 8 * (lane >> 5), n = 32 * w + (lane & 31);
    float v[8];
    if (!gru) {
        const float *src = (nt == 0 ? w_ih : w_hh) + (int64_t)k * H + n;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = src[(int64_t)e * H];
    } else {        // k .. k+7 lie inside one gate slot (8 | H)
        const int slot = k / H, j = k - slot * H;
        const bool zero = (slot == 2 && nt == 1) || (slot == 3 && nt == 0);
        const float *src = (nt == 0 ? w_ih : w_hh) + (int64_t)gru_weight_row(slot, j, H) * H + n;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = zero ? 0.0f : src[(int64_t)e * H];
    }
    const float sc = exp2i(scale_exp(nt == 0 ? m_ih : m_hh));
    u32x4 q0, q1;
#pragma unroll
    for (int h = 0; h < 4; h++) {
        uint32_t x0, x1;
        split2h(v[2 * h] * sc, v[2 * h + 1] * sc, x0, x1);
        q0[h] = x0; q1[h] = x1;
    }
    u32x4 *dst = WpT + ((int64_t)(w * NU + u) * 2 * 4 + f) * 64 + lane;
    dst[0] = q0;
    dst[4 * 64] = q1;
}

// Both packings in ONE launch (round 6; two until then): workgroups [0, nb_fwd) the forward's fragments and the summed biases,
// the rest the BPTT's (none for an inference forward).  Every wave takes the operand maxima from range_part_kernel's partials;
// the first thread also leaves them in range->w_ih / w_hh for the recurrent kernels.
__global__ __launch_bounds__(256) void pack_fb_kernel(const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                                      const float *__restrict__ b_ih, const float *__restrict__ b_hh, int H, int G, int gru,
                                                      const uint32_t *__restrict__ part, SeqRange *__restrict__ range, int nb_fwd,
                                                      u32x4 *__restrict__ Wp, float *__restrict__ biasc, u32x4 *__restrict__ WpT) {
    float m_ih, m_hh;
    range_from_parts(part, m_ih, m_hh);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        range->w_ih = __float_as_uint(m_ih);
        range->w_hh = __float_as_uint(m_hh);
    }
    if ((int)blockIdx.x < nb_fwd)
        pack_fwdh_body(blockIdx.x * 256 + threadIdx.x, w_ih, w_hh, b_ih, b_hh, H, G, gru, m_ih, m_hh, Wp, biasc);
    else
        pack_bwdh_body(((int)blockIdx.x - nb_fwd) * 256 + threadIdx.x, w_ih, w_hh, H, G, gru, m_ih, m_hh, WpT);
}

template <int H, int GC>
__global__ __launch_bounds__(H / 32 * 64, H == 32 ? 1 : PN_BWDH_WAVES) void seq_bwdh_kernel(SeqBwdParams p) {
    constexpr int G = GC == 3 ? 4 : GC;         // GC: 4 = LSTM, 1 = tanh RNN, 3 = GRU on the LSTM's four gate slots
    constexpr bool GRU = GC == 3;
    constexpr int MT = 32, NW = H / 32;
    constexpr bool PACKED = GC == 4;                            // (the forward's layout: pack_gates quads)
    constexpr int NT = NW * 64, GH = G * H, SV = PACKED ? 4 : (G == 4 ? 5 : 1);
    constexpr int PB = 2 * GH + 16, PLANE = MT * PB;            // plane row pitch / plane size, bytes
    constexpr int NU = GH / 32;                                 // units of two k-steps
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    int *s_rowidx = reinterpret_cast<int *>(ldsb + 2 * PLANE);   // [MT][L] gather rows of this tile
    int *s_slotof = s_rowidx + MT * p.L;                         // [MT]
    float *s_max = reinterpret_cast<float *>(s_slotof + MT);     // [8] wave maxima of |dG_t|
    uint8_t *s_keep = reinterpret_cast<uint8_t *>(s_max + 8);    // [2][MT][H/4] dropout keep bits of step t (t & 1)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31;
    // (descending path order: the BPTT starts on the tiles the forward wrote last; its small tiles -- SeqTiling, the
    //  remainder round -- are therefore the ones at the START of the path range, dispatched last)
    const SeqTile tl = seq_tile_of(p.tiling, (int)(gridDim.x - 1 - blockIdx.x), MT, p.P);
    const int q0 = tl.q0, rows_here = tl.rows;          // rows_here >= 1
    const int col = 32 * wave + li;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    for (int i = tid; i < MT * p.L; i += NT) s_rowidx[i] = i / p.L < rows_here ? p.rowidx[(int64_t)q0 * p.L + i] : 0;
    for (int i = tid; i < MT; i += NT) s_slotof[i] = i < rows_here ? p.slotof[q0 + i] : 0;

    const float keep_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (1.0f - p.p_drop))));
    const int e_ih = scale_exp(__uint_as_float(p.range->w_ih)), e_hh = scale_exp(__uint_as_float(p.range->w_hh));
    const size_t tile_row = (size_t)q0 * (size_t)p.L;
    const float *saved_t = p.saved + tile_row * (SV * H);
    float *dG_t = p.dG + tile_row * GH;
    const uint8_t *keep_t = p.keep ? p.keep + tile_row * (H / 4) : nullptr;
    const float *dhn_t = p.dhn + (size_t)q0 * H;
    f32x16 dh, dc;
    [[maybe_unused]] f32x16 cnext;      // LSTM: c_t of the step processed next (= c_{t-1} now): loaded once, carried in registers
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = acc_row(r, lane);
        const int rc = min(row, rows_here - 1);
        const float dh0 = at_bytes(dhn_t, ((uint32_t)rc * (uint32_t)H + col) * 4u);     // unconditional load, select afterwards
        dh[r] = row < rows_here ? dh0 : 0.0f;
        dc[r] = 0.0f;
    }
    float launch_max = 0.0f;        // largest |dG| this workgroup has seen (wave-uniform after each step)

    // ---- the saved values of one step, as loaded (those of step t - 1 are requested right after step t's k
    //      loop, ahead of its scatter -- ordinary loads, which the compiler cannot sink below the scatter's atomics -- so their
    //      latency runs under the atomics instead of in front of the next cell backward).  All loads of a step are issued
    //      together, unconditionally (padded rows read a clamped row and are zeroed afterwards).
    constexpr int NRAW = PACKED ? 4 : GRU ? 5 : 1;
    uint32_t raw[NRAW][16];
    auto load_saved = [&](int t, int lane_x) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rc = min(acc_row(r, lane_x), rows_here - 1);
            const uint32_t base = ((uint32_t)rc * (uint32_t)p.L + t) * (uint32_t)(SV * H);
            if constexpr (PACKED) {       // one 16-byte quad per (path step, unit): {packed gates (3 dwords), c_{t-1}}
                const uint4 qd = *reinterpret_cast<const uint4 *>(&at_bytes(saved_t, (base + 4u * col) * 4u));
                raw[0][r] = qd.x; raw[1][r] = qd.y; raw[2][r] = qd.z; raw[NRAW > 3 ? 3 : 0][r] = qd.w;
            } else {
                const float *sv = &at_bytes(saved_t, (base + col) * 4u);
                if constexpr (GRU) {
#pragma unroll
                    for (int k = 0; k < 5; k++) raw[k][r] = __float_as_uint(sv[k * H]);    // r, z, n, W_hn h + b_hn, h_{t-1}
                } else {
                    raw[0][r] = __float_as_uint(sv[0]);                                    // h_t
                }
            }
        }
    };
    load_saved(p.L - 1, lane);

    for (int t = p.L - 1; t >= 0; t--) {
        // (row numbers are re-derived from an opaque copy of the lane id in every step: as loop invariants the
        //  per-row offsets would occupy ~40 registers across the MFMA loop and spill)
        const int lane_t = fresh_lane();
        if (t == p.L - 1) HSTAMP_RT(62);
        HSTAMP(6 * (p.L - 1 - t) + 0);
        if (p.keep) {      // this step's keep bytes (MT rows x H/4) -> LDS, read by the scatter phase below
            const int tid_t = wave_u * 64 + lane_t;
            for (int i = tid_t; i < MT * (H / 16); i += NT) {
                const int row = i / (H / 16), w = i - row * (H / 16);
                const uint32_t rc = (uint32_t)min(row, rows_here - 1);
                reinterpret_cast<uint32_t *>(s_keep + (t & 1) * MT * (H / 4))[i] =
                    at_bytes(reinterpret_cast<const uint32_t *>(keep_t), (rc * (uint32_t)p.L + t) * (uint32_t)(H / 4) + 4u * w);
            }
        }
        // ---- cell backward into registers
        float dgv[G][16];
        float vmax = 0.0f;
        {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                // (unpacked element by element: a second set of arrays beside `raw` would not fit the register file)
                float vi_, vf_ = 0.f, vg_ = 0.f, vo_ = 0.f, vc_ = 0.f, vn_ = 0.f;
                if constexpr (PACKED) {
                    unpack_gates(make_uint3(raw[0][r], raw[NRAW > 1 ? 1 : 0][r], raw[NRAW > 2 ? 2 : 0][r]), vi_, vf_, vg_, vo_);
                    vc_ = t > 0 ? __uint_as_float(raw[NRAW > 3 ? 3 : 0][r]) : 0.0f;
                    vn_ = cnext[r];
                    // (the last step's c is not stored -- it is f c_{t-1} + i g of the values just unpacked, which differs from
                    //  the forward's by the gates' 24-bit rounding, 2^-24 of a value that enters through tanh)
                    if (t == p.L - 1) vn_ = vf_ * vc_ + vi_ * vg_;
                } else if (GRU) {
                    vi_ = __uint_as_float(raw[0][r]); vf_ = __uint_as_float(raw[NRAW > 1 ? 1 : 0][r]); vg_ = __uint_as_float(raw[NRAW > 2 ? 2 : 0][r]);
                    vo_ = __uint_as_float(raw[NRAW > 3 ? 3 : 0][r]);
                    vc_ = __uint_as_float(raw[NRAW > 4 ? 4 : 0][r]);
                } else {
                    vi_ = __uint_as_float(raw[0][r]);
                }
                const int row = acc_row(r, lane_t);
                const bool ok = row < rows_here;
                float *d = &at_bytes(dG_t, (((uint32_t)min(row, rows_here - 1) * (uint32_t)p.L + t) * (uint32_t)GH + col) * 4u);
                if (GRU) {
                    // h = (1 - z) n + z h_prev,  n = tanh(nx + r nh):  gradients of the four slots r, z, nx, nh; the direct
                    // path d h_t / d h_{t-1} = z is carried in dc[] across the GEMM and added to its dh output
                    const float rg = vi_, zg = vf_, ng = vg_, nh = vo_, hp = vc_;
                    const float dhv = dh[r];
                    const float dnp = dhv * (1.0f - zg) * (1.0f - ng * ng);
                    float a_r = dnp * nh * rg * (1.0f - rg);
                    float a_z = dhv * (hp - ng) * zg * (1.0f - zg);
                    float a_nx = dnp;
                    float a_nh = dnp * rg;
                    if (!ok) a_r = a_z = a_nx = a_nh = 0.0f;
                    dc[r] = ok ? dhv * zg : 0.0f;
                    dgv[0][r] = a_r; dgv[G > 1 ? 1 : 0][r] = a_z; dgv[G > 2 ? 2 : 0][r] = a_nx; dgv[G > 3 ? 3 : 0][r] = a_nh;
                    if (ok) d[0] = a_r, d[H] = a_z, d[2 * (G > 1 ? H : 0)] = a_nx, d[3 * (G > 1 ? H : 0)] = a_nh;
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(a_r), fabsf(a_z))), fmaxf(fabsf(a_nx), fabsf(a_nh)));
                } else if (G == 4) {
                    const float ig = vi_, fg = vf_, gg = vg_, og = vo_, cprev = vc_;
                    const float tc = tanhf_(vn_);
                    const float dhv = dh[r];
                    const float d_o = dhv * tc;
                    const float dct = dc[r] + dhv * og * (1.0f - tc * tc);
                    float a_i = dct * gg * ig * (1.0f - ig);
                    float a_f = dct * cprev * fg * (1.0f - fg);
                    float a_g = dct * ig * (1.0f - gg * gg);
                    float a_o = d_o * og * (1.0f - og);
                    if (!ok) a_i = a_f = a_g = a_o = 0.0f;
                    dc[r] = dct * fg;
                    cnext[r] = cprev;
                    dgv[0][r] = a_i; dgv[G > 1 ? 1 : 0][r] = a_f; dgv[G > 2 ? 2 : 0][r] = a_g; dgv[G > 3 ? 3 : 0][r] = a_o;
                    if (ok) d[0] = a_i, d[H] = a_f, d[2 * (G > 1 ? H : 0)] = a_g, d[3 * (G > 1 ? H : 0)] = a_o;
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(a_i), fabsf(a_f))), fmaxf(fabsf(a_g), fabsf(a_o)));
                } else {
                    const float h = vi_;
                    const float a = ok ? dh[r] * (1.0f - h * h) : 0.0f;
                    dgv[0][r] = a;
                    if (ok) d[0] = a;
                    vmax = fmaxf(vmax, fabsf(a));
                }
            }
        }
        vmax = wave_max(vmax);
        if (lane_t == 0) s_max[wave_u] = vmax;
        HSTAMP(6 * (p.L - 1 - t) + 1);
        __syncthreads();        // the wave maxima are in place; every wave is past the previous step's k loop
        HSTAMP(6 * (p.L - 1 - t) + 2);
        float tmax = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; w++) tmax = fmaxf(tmax, s_max[w]);
        tmax = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(tmax)));
        launch_max = fmaxf(launch_max, tmax);
        const int e_g = scale_exp(tmax);
        const float s_g = exp2i(e_g);
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                uint32_t x0, x1;
                split2h(dgv[g][r] * s_g, dgv[g][r + 1] * s_g, x0, x1);
                unsigned char *d = ldsb + acc_row(r, lane_t) * PB + 2 * (g * H + col);
                *reinterpret_cast<uint16_t *>(d) = (uint16_t)x0;
                *reinterpret_cast<uint16_t *>(d + PB) = (uint16_t)(x0 >> 16);
                *reinterpret_cast<uint16_t *>(d + PLANE) = (uint16_t)x1;
                *reinterpret_cast<uint16_t *>(d + PLANE + PB) = (uint16_t)(x1 >> 16);
            }
        __syncthreads();
        HSTAMP(6 * (p.L - 1 - t) + 3);

        // ---- [dx_t ; dh_{t-1}] = dG_t . [W_ih | W_hh]; the dh half is not needed at t = 0 ------------------------
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = 0.0f;
        const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.WpT) + (size_t)wave_u * (NU * 8 * 1024);
        const uint32_t voff = lane_t * 16;
        const unsigned char *arow = ldsb + (lane_t & 31) * PB + 16 * (lane_t >> 5);
        // vmcnt (in order) at the top of unit u: [Bh(u) Bl(u)] + the Bh(u+1) just issued
        auto mfma_phase = [&](auto ntn_tag) {
            constexpr int NTN = decltype(ntn_tag)::value, NF = 2 * NTN;       // fragments per plane and unit
            u32x4 Bha[NF], Bhb[NF], Bl[NF];
            auto load = [&](u32x4 (&B)[NF], int u, int pl) {      // fragment kk*2 + nt of the unit's plane
                const unsigned char *sb = wb + (size_t)(u * 2 + pl) * 4096;
                if constexpr (NTN == 2) {
                    async_load_frags<4>(B, sb, voff);
                } else {
                    async_load_b128_s<0>(B[0], sb, voff);
                    async_load_b128_s<2048>(B[1], sb, voff);
                }
            };
            u32x4 a[2][2];      // [kk][plane]
            auto aread = [&](int u, int pl) {
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
                    a[kk][pl] = *reinterpret_cast<const u32x4 *>(arow + pl * PLANE + 64 * u + 32 * kk);
            };
            auto group = [&](const u32x4 (&B)[NF], int pl) {
#pragma unroll
                for (int f = 0; f < NF; f++) acc[f % NTN] = mfma_f16(a[f / NTN][pl], B[f], acc[f % NTN]);
            };
            auto unit = [&](int u, u32x4 (&Bh)[NF], u32x4 (&Bhnext)[NF]) {
                const int un = min(u + 1, NU - 1);
                load(Bhnext, un, 0);
                wait_frag<2 * NF, NF>(Bh);
                group(Bh, 1);
                aread(un, 1);
                group(Bh, 0);
                wait_frag<NF, NF>(Bl);
                group(Bl, 0);
                aread(un, 0);
                load(Bl, un, 1);
            };
            load(Bha, 0, 0);
            load(Bl, 0, 1);
            aread(0, 0);
            aread(0, 1);
#pragma unroll 1
            for (int u = 0; u < NU; u += 2) {
                unit(u, Bha, Bhb);
                if (NU % 2 == 0 || u + 1 < NU) unit(u + 1, Bhb, Bha);      // (an odd count -- the RNN at H = 32, 96, ... -- ends on the first)
            }
            wait_frag<0, NF>(Bha);       // drain (harmless re-loads of the last unit)
            wait_frag<0, NF>(Bhb);
            wait_frag<0, NF>(Bl);
        };
        if (t > 0)
            mfma_phase(std::integral_constant<int, 2>{});
        else
            mfma_phase(std::integral_constant<int, 1>{});
        HSTAMP(6 * (p.L - 1 - t) + 4);
        const float inv_x = exp2i(-(e_g + e_ih)), inv_h = exp2i(-(e_g + e_hh));
        load_saved(max(t - 1, 0), lane_t);        // (dgv is dead: these take its registers; t = 0: a harmless re-load)

        // ---- gather backward: dZ[row(q, t)] += mask * dx.  Step 0 is the last one of the kernel and its rows are the
        //      paths' own start nodes: the wave parks its 32 x 32 block in the (then dead) plane region and each half-wave
        //      adds up runs of equal table rows, one atomic per run and column (as seq_bwd3_kernel)
        constexpr bool MERGE_STEP0 = (H & (H - 1)) == 0;
        if (MERGE_STEP0 && t == 0 && p.merge0) {
            static_assert(2 * PLANE >= NW * 32 * 33 * 4, "the scatter scratch fits the plane region");
            __syncthreads();            // every wave is done with the planes
            const int lane_s = fresh_lane(), li_s = lane_s & 31;
            float *scr = reinterpret_cast<float *>(ldsb) + wave_u * (32 * 33);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rl = acc_row(r, lane_s), row = rl;
                float dx = acc[0][r] * inv_x;
                if (row < rows_here) {
                    if (p.mask)
                        dx *= p.mask[((uint64_t)t * p.Pmask + s_slotof[row]) * H + col];
                    else if (p.keep)
                        dx = (s_keep[((t & 1) * MT + row) * (H / 4) + (col >> 2)] >> (col & 3)) & 1 ? dx * keep_scale : 0.0f;
                }
                scr[rl * 33 + li_s] = dx;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int hk = lane_s >> 5;
            int cur = -1;
            float run = 0.0f;
#pragma unroll 1
            for (int i = 0; i < 16; i++) {
                const int rl = 16 * hk + i, row = rl;
                const int rid = row < rows_here ? s_rowidx[row * p.L] : -1;       // (uniform over a half-wave)
                if (rid != cur) {
                    if (cur >= 0) atomicAdd(p.dZ + ((size_t)(uint32_t)cur * (uint32_t)H + col), run);
                    cur = rid;
                    run = 0.0f;
                }
                run += scr[rl * 33 + li_s];
            }
            if (cur >= 0) atomicAdd(p.dZ + ((size_t)(uint32_t)cur * (uint32_t)H + col), run);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = acc_row(r, lane_t);
                if (row < rows_here) {
                    float dx = acc[0][r] * inv_x;
                    if (p.mask)
                        dx *= p.mask[((uint64_t)t * p.Pmask + s_slotof[row]) * H + col];
                    else if (p.keep)
                        dx = (s_keep[((t & 1) * MT + row) * (H / 4) + (col >> 2)] >> (col & 3)) & 1 ? dx * keep_scale : 0.0f;
                    float *dst = p.dZ + ((size_t)(uint32_t)s_rowidx[row * p.L + t] * (uint32_t)H + col);
                    if (p.store_dx)
                        *dst = dx;          // deterministic mode: a row of its own per path step (det_scatter_kernel adds them up)
                    else if (dx != 0.0f)    // (no atomic for an element dropout zeroed: 70 % of them at p = 0.7; BPTT 0.375 -> 0.361 ms)
                        atomicAdd(dst, dx);
                }
                dh[r] = GRU ? acc[1][r] * inv_h + dc[r] : acc[1][r] * inv_h;
            }
        }
        HSTAMP(6 * (p.L - 1 - t) + 5);
    }
    HSTAMP_RT(63);
    if (tid == 0 && launch_max > 0.0f) atomicMax(&p.range->dg, __float_as_uint(launch_max));
}

// =====================================================================================================================
// weight gradient:  [g_W_ih | g_W_hh] [G*H, 2H] = dG^T [G*H, R] . XH [R, 2H]   (R = P*L rows; colsum(dG) = bias gradient)
//   wgrad4_kernel's decomposition (pn_seq4.hip): 256 x 256 output tile per workgroup (8 waves, 64 x 128 each), the R rows
//   split over blockIdx.z in strided K tiles of 16 rows, two LDS stages, the rows of tile i + 1 split and written to LDS
//   between the MFMA groups of tile i, the loads of tile i + 2 in flight.  Two planes: three groups of eight MFMAs per tile
//   and wave (a_hi b_hi, a_hi b_lo, a_lo b_hi), one barrier ahead of the last group, under which the first fragments of the
//   next tile are fetched.  Scales: dG by the launch-wide power of two for max |dG| (range->dg, left by the BPTT), the x
//   columns of [x | h] by the one for max |Z| * xmul, the h columns by 2^14; divided out when the partial tile is stored.
// =====================================================================================================================
constexpr int WH_BM = 256, WH_BN = 256, WH_KT = 16, WH_THREADS = 512;
constexpr int WH_BLK = 4 * 68;                              // 16-byte slots per (plane, operand, k-octet) block
constexpr int WH_PLANE = 2 * 2 * WH_BLK;                    // slots per plane: 2 operands x 2 k-octets
constexpr int WH_STAGE = 2 * WH_PLANE;                      // slots per stage (34 816 bytes)
constexpr int WH_LDS_BYTES = 2 * WH_STAGE * 16;             // 69 632

__global__ __launch_bounds__(WH_THREADS, 2) void wgradh_kernel(WgradParams p, int H) {
    extern __shared__ __attribute__((aligned(16))) u32x4 ldsw[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, hk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned by = blockIdx.y, bz = blockIdx.z;
    const int m0 = by * WH_BM, n0 = blockIdx.x * WH_BN;
    const int64_t ntiles = (p.R + WH_KT - 1) / WH_KT;
    const int64_t nz = gridDim.z;
    const int64_t my_tiles = bz < ntiles ? (ntiles - bz + nz - 1) / nz : 0;
    if (my_tiles == 0) return;      // block-uniform
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int e_g = scale_exp(__uint_as_float(p.range->dg)), e_x = scale_exp(__uint_as_float(p.range->x) * p.xmul), e_h = 14;
    const int op = tid >> 8, rq = (tid >> 6) & 3, cq = tid & 63;
    const float *src = op == 0 ? p.dG : p.xh;
    const int ld = op == 0 ? p.GH : p.H2;
    const int c0 = (op == 0 ? m0 : n0) + 4 * cq;
    const bool c_ok = c0 < ld;
    const float *srcc = src + (c_ok ? c0 : 0);
    const float sc_op = exp2i(op == 0 ? e_g : c0 < H ? e_x : e_h);       // (4 | H: a thread's four columns share a half)
    f32x4 rgA[4], rgB[4];
    auto row0_of = [&](int64_t i) { return (bz + min(i, my_tiles - 1) * nz) * WH_KT; };     // (clamped: harmless re-load)
    auto issue = [&](f32x4 (&rg)[4], int64_t i) {
        const int64_t k0 = row0_of(i);
#pragma unroll
        for (int e = 0; e < 4; e++) async_load_b128(rg[e], srcc + min(k0 + 4 * rq + e, p.R - 1) * ld);
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};     // column sums of dG over this thread's rows (bias gradient)
    // operand op, k-octet rq >> 1, column 4 cq + j at slot j * 68 + cq; this thread's rows are half (rq & 1) of the octet
    unsigned char *stage_wr = reinterpret_cast<unsigned char *>(ldsw + (op * 2 + (rq >> 1)) * WH_BLK + cq) + (rq & 1) * 8;
    // column j of the thread's 4 x 4 block of tile i -> the two planes of stage buf (tiles past the end: zeros)
    auto piece = [&](f32x4 (&rg)[4], int64_t i, int buf, int j) {
        const int64_t k0 = row0_of(i);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = (c_ok && i < my_tiles && k0 + 4 * rq + e < p.R) ? rg[e][j] : 0.0f;
        unsigned char *w = stage_wr + (size_t)buf * (WH_STAGE * 16);
        uint32_t x0, x1, y0, y1;
        split2h(v[0] * sc_op, v[1] * sc_op, x0, x1);
        split2h(v[2] * sc_op, v[3] * sc_op, y0, y1);
        bs[j] += (v[0] + v[1]) + (v[2] + v[3]);
        *reinterpret_cast<uint2 *>(w + j * 68 * 16) = make_uint2(x0, y0);
        *reinterpret_cast<uint2 *>(w + (WH_PLANE + j * 68) * 16) = make_uint2(x1, y1);
    };
    const int sa = hk * WH_BLK + (li & 3) * 68 + (li >> 2) + wm * 16;                     // operand 0 (dG^T), k-octet hk
    const int sb = (2 + hk) * WH_BLK + (li & 3) * 68 + (li >> 2) + wn * 32;               // operand 1 ([x|h])
#define WH_FENCE() __builtin_amdgcn_sched_barrier(0)
    // One step = the products of the tile in stage buf, the commit of tile inext (registers rg) to the other stage in the
    // gaps after the first two MFMA groups, ONE barrier, then the last group -- under which the first fragments of the next
    // tile are already fetched from the stage just completed.  On entry A0 / B0 hold (or are receiving) the hi fragments
    // of this tile and B1 its lo b fragments; on exit the same holds for the next tile with the roles of B0 and B1 exchanged.
    u32x4 fA0[2], fA1[2], fB0[4], fB1[4];
    auto group = [&](const u32x4 (&a)[2], const u32x4 (&b)[4]) {
        WH_FENCE();
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = mfma_f16(a[i], b[j], acc[i][j]);
        WH_FENCE();
    };
    auto step = [&](int buf, u32x4 (&A0)[2], u32x4 (&A1)[2], u32x4 (&B0)[4], u32x4 (&B1)[4], f32x4 (&rg)[4], int64_t inext) {
        const u32x4 *fa = ldsw + buf * WH_STAGE + sa;
        const u32x4 *na = ldsw + (buf ^ 1) * WH_STAGE + sa, *nb = ldsw + (buf ^ 1) * WH_STAGE + sb;
        group(A0, B0);                                  // a_hi . b_hi
        piece(rg, inext, buf ^ 1, 0);
#pragma unroll
        for (int i = 0; i < 2; i++) A1[i] = fa[WH_PLANE + i * 8];
        piece(rg, inext, buf ^ 1, 1);
        group(A0, B1);                                  // a_hi . b_lo
        piece(rg, inext, buf ^ 1, 2);
        piece(rg, inext, buf ^ 1, 3);
        WH_FENCE();
        __syncthreads();        // stage buf ^ 1 is complete; every read of stage buf has been issued and has landed
        group(A1, B0);                                  // a_lo . b_hi
#pragma unroll
        for (int i = 0; i < 2; i++) A0[i] = na[i * 8];                          // next tile: a hi
#pragma unroll
        for (int j = 0; j < 4; j++) B1[j] = nb[j * 8];                          //            b hi (the next step's B0)
        WH_FENCE();
#pragma unroll
        for (int j = 0; j < 4; j++) B0[j] = nb[WH_PLANE + j * 8];               //            b lo (the next step's B1)
    };

    issue(rgA, 0);
    wait_vm<0>(rgA[0], rgA[1], rgA[2], rgA[3]);
#pragma unroll
    for (int j = 0; j < 4; j++) piece(rgA, 0, 0, j);
    issue(rgB, 1);
    __syncthreads();
    {
        const u32x4 *fa = ldsw + sa, *fb = ldsw + sb;
#pragma unroll
        for (int i = 0; i < 2; i++) fA0[i] = fa[i * 8];
#pragma unroll
        for (int j = 0; j < 4; j++) fB0[j] = fb[j * 8];
#pragma unroll
        for (int j = 0; j < 4; j++) fB1[j] = fb[WH_PLANE + j * 8];
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
#undef WH_FENCE
    float *pw = p.part_w + (int64_t)bz * p.GH * p.H2;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + wn * 128 + j * 32 + li;
            if (n >= p.H2) continue;
            const float inv = exp2i(-(e_g + (n < H ? e_x : e_h)));
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (m < p.GH) pw[(int64_t)m * p.H2 + n] = acc[i][j][r] * inv;
            }
        }
    // bias gradient: the four row-quad owners of a column add up through LDS
    if (blockIdx.x != 0) return;   // block-uniform
    float *fl = reinterpret_cast<float *>(ldsw);
    if (op == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) fl[rq * WH_BM + 4 * cq + j] = bs[j];
    }
    __syncthreads();
    if (tid < WH_BM && m0 + tid < p.GH)
        p.part_b[(int64_t)bz * p.GH + m0 + tid] =
            (fl[tid] + fl[WH_BM + tid]) + (fl[2 * WH_BM + tid] + fl[3 * WH_BM + tid]);
}

template <int H, int GC>
int launch_fwdh_t(pn_context *ctx, hipStream_t stream, const SeqFwdParams &sp) {
    constexpr int RB = 1;       // row blocks of 32 paths per wave (the kernel keeps the parameter; 2 measured slower)
    constexpr int MT = 32 * RB;
    const size_t lds_bytes = (size_t)2 * MT * (4 * H + 16) + (size_t)(MT * sp.L + MT) * 4;
    auto kern = seq_fwdh_kernel<H, GC, RB>;
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), (int)lds_bytes)) return rc;
    SeqFwdParams lp = sp;
    int blocks = 0;
    if (int rc = plan_tiling(ctx, reinterpret_cast<const void *>(kern), H / 32 * 64, lds_bytes, sp.P, MT, /*small_first=*/false,
                             &lp.tiling, &blocks))
        return rc;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(H / 32 * 64), lds_bytes, stream, lp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}
template <int H, int GC>
constexpr size_t bwdh_lds_bytes(int L) {
    constexpr int MT = 32, G = GC == 3 ? 4 : GC;
    return (size_t)2 * MT * (2 * G * H + 16) + (size_t)(MT * L + MT) * 4 + 32 + (size_t)2 * MT * (H / 4);
}
template <int H, int GC>
int launch_bwdh_t(pn_context *ctx, hipStream_t stream, const SeqBwdParams &sp) {
    constexpr int MT = 32;
    const size_t lds_bytes = bwdh_lds_bytes<H, GC>(sp.L);
    auto kern = seq_bwdh_kernel<H, GC>;
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), (int)lds_bytes)) return rc;
    SeqBwdParams lp = sp;
    int blocks = 0;
    if (int rc = plan_tiling(ctx, reinterpret_cast<const void *>(kern), H / 32 * 64, lds_bytes, sp.P, MT,
                             /*small_first=*/true, &lp.tiling, &blocks))
        return rc;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(H / 32 * 64), lds_bytes, stream, lp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}
template <int H, int GC>
int launch_fwdzw_t(pn_context *ctx, hipStream_t stream, const SeqFwdParams &sp) {
    constexpr int MT = 32;
    const size_t lds_bytes = (size_t)2 * MT * (2 * H + 16) + (size_t)(MT * sp.L) * 4;
    auto kern = seq_fwdzw_kernel<H, GC>;
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), (int)lds_bytes)) return rc;
    hipLaunchKernelGGL(kern, dim3((sp.P + MT - 1) / MT), dim3(H / 32 * 64), lds_bytes, stream, sp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}
template <int GC>
int dispatch_fwdzw(pn_context *ctx, hipStream_t s, int H, const SeqFwdParams &sp) {
    switch (H) {
        case 32: return launch_fwdzw_t<32, GC>(ctx, s, sp);
        case 64: return launch_fwdzw_t<64, GC>(ctx, s, sp);
        case 96: return launch_fwdzw_t<96, GC>(ctx, s, sp);
        case 128: return launch_fwdzw_t<128, GC>(ctx, s, sp);
        case 160: return launch_fwdzw_t<160, GC>(ctx, s, sp);
        case 192: return launch_fwdzw_t<192, GC>(ctx, s, sp);
        case 224: return launch_fwdzw_t<224, GC>(ctx, s, sp);
        case 256: return launch_fwdzw_t<256, GC>(ctx, s, sp);
    }
    PN_FAIL(PN_ERR_ARG, "hidden size %d not supported", H);
}
template <int GC>
int dispatch_fwdh(pn_context *ctx, hipStream_t s, int H, const SeqFwdParams &sp) {
    switch (H) {
        case 32: return launch_fwdh_t<32, GC>(ctx, s, sp);
        case 64: return launch_fwdh_t<64, GC>(ctx, s, sp);
        case 96: return launch_fwdh_t<96, GC>(ctx, s, sp);
        case 128: return launch_fwdh_t<128, GC>(ctx, s, sp);
        case 160: return launch_fwdh_t<160, GC>(ctx, s, sp);
        case 192: return launch_fwdh_t<192, GC>(ctx, s, sp);
        case 224: return launch_fwdh_t<224, GC>(ctx, s, sp);
        case 256: return launch_fwdh_t<256, GC>(ctx, s, sp);
    }
    PN_FAIL(PN_ERR_ARG, "hidden size %d not supported", H);
}
template <int GC>
int dispatch_bwdh(pn_context *ctx, hipStream_t s, int H, const SeqBwdParams &sp) {
    switch (H) {
        case 32: return launch_bwdh_t<32, GC>(ctx, s, sp);
        case 64: return launch_bwdh_t<64, GC>(ctx, s, sp);
        case 96: return launch_bwdh_t<96, GC>(ctx, s, sp);
        case 128: return launch_bwdh_t<128, GC>(ctx, s, sp);
        case 160: return launch_bwdh_t<160, GC>(ctx, s, sp);
        case 192: return launch_bwdh_t<192, GC>(ctx, s, sp);
        case 224: return launch_bwdh_t<224, GC>(ctx, s, sp);
        case 256: return launch_bwdh_t<256, GC>(ctx, s, sp);
    }
    PN_FAIL(PN_ERR_ARG, "hidden size %d not supported", H);
}

}  // namespace

#if PN_TRACE_H
extern "C" int pn_debug_set_trace_h(long long *dev_buf) {     // tuning builds only; not part of the ABI
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_h), &dev_buf, sizeof dev_buf) == hipSuccess ? 0 : -4;
}
#endif

// the tile split of a launch (host arithmetic only; not part of the ABI): out = {n_big, n_small, small_rows, blocks}
extern "C" int pn_debug_seq_tiling(int64_t P, int slots, int cus, int mode, int small_first, int32_t out[4]) {
    pn::SeqTiling tg;
    int blocks = 0;
    seq_tiling_for(P, 32, slots, cus, mode, small_first != 0, &tg, &blocks);
    out[0] = tg.n_big; out[1] = tg.n_small; out[2] = tg.small_rows; out[3] = blocks;
    return 0;
}

namespace pn {


int launch_range_rows(void *stream, const float *rows, int64_t nrows, int H, const int32_t *count, SeqRange *range) {
    hipStream_t s = (hipStream_t)stream;        // (range->x was cleared by launch_pack_fb, ordered before this launch)
    const int64_t n4 = nrows * (H / 4);
    // (on the step's critical path between the bank and the recurrence: one or two 16-byte loads per thread, 11 -> ~4 us at
    //  the headline shape's 5.5 MB)
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (n4 + 1023) / 1024));
    hipLaunchKernelGGL(range_rows_kernel, dim3(blocks), dim3(256), 0, s, rows, nrows, H / 4, count, range);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int launch_pack_fb(void *stream, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, int H, int G, int Gw,
                   int gru, int clear_x, SeqRange *range, void *part, void *Wp, float *biasc, void *WpT) {
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(range_part_kernel, dim3(RANGE_PARTS), dim3(256), 0, s, w_ih, w_hh, (int64_t)Gw * H * H / 4, clear_x, range,
                       reinterpret_cast<uint32_t *>(part));
    PN_CHECK_HIP(hipGetLastError());
    const int nb = (G * H * H / 4 + 255) / 256;         // workgroups of either packing (WpT == nullptr: the forward's alone)
    hipLaunchKernelGGL(pack_fb_kernel, dim3((unsigned)(WpT ? 2 * nb : nb)), dim3(256), 0, s, w_ih, w_hh, b_ih, b_hh, H, G, gru,
                       reinterpret_cast<const uint32_t *>(part), range, nb, reinterpret_cast<u32x4 *>(Wp), biasc,
                       reinterpret_cast<u32x4 *>(WpT));
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int launch_seq_fwdh(pn_context *ctx, void *stream, int H, int gc, const SeqFwdParams &sp) {
    if (!sp.range) PN_FAIL(PN_ERR_ARG, "seq_fwdh: operand ranges missing");
    hipStream_t s = (hipStream_t)stream;
    return gc == 3 ? dispatch_fwdh<3>(ctx, s, H, sp) : gc == 4 ? dispatch_fwdh<4>(ctx, s, H, sp) : dispatch_fwdh<1>(ctx, s, H, sp);
}

int launch_seq_fwdzw(pn_context *ctx, void *stream, int H, int gc, const SeqFwdParams &sp) {
    if (!sp.range || !sp.ZW) PN_FAIL(PN_ERR_ARG, "seq_fwdzw: operand ranges / ZW missing");
    hipStream_t s = (hipStream_t)stream;
    return gc == 3 ? dispatch_fwdzw<3>(ctx, s, H, sp) : gc == 4 ? dispatch_fwdzw<4>(ctx, s, H, sp) : dispatch_fwdzw<1>(ctx, s, H, sp);
}

int launch_seq_bwdh(pn_context *ctx, void *stream, int H, int gc, const SeqBwdParams &sp) {
    if (!sp.range) PN_FAIL(PN_ERR_ARG, "seq_bwdh: operand ranges missing");
    hipStream_t s = (hipStream_t)stream;        // (range->dg was cleared by the forward's launch_pack_fb)
    return gc == 3 ? dispatch_bwdh<3>(ctx, s, H, sp) : gc == 4 ? dispatch_bwdh<4>(ctx, s, H, sp) : dispatch_bwdh<1>(ctx, s, H, sp);
}

int launch_wgradh(pn_context *ctx, void *stream, const WgradParams &wp, int H, int nsplit) {
    if (!wp.range) PN_FAIL(PN_ERR_ARG, "wgradh: operand ranges missing");
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(wgradh_kernel), WH_LDS_BYTES)) return rc;
    hipLaunchKernelGGL(wgradh_kernel, dim3((wp.H2 + WH_BN - 1) / WH_BN, (wp.GH + WH_BM - 1) / WH_BM, nsplit), dim3(WH_THREADS),
                       WH_LDS_BYTES, (hipStream_t)stream, wp, H);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

}  // namespace pn
