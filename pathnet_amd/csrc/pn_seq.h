// pn_seq.h -- launch parameters of the three recurrent kernels (forward recurrence, BPTT, weight gradient), shared by
// pn_pagg.hip (the bf16 x 3 kernels for every hidden size up to 256), pn_seqh.hip (the fp16 x 2 kernels, the default) and
// pn_seq4.hip (the bf16 x 3 weight-gradient GEMM of the headline shape).  Not part of the ABI.
#pragma once
#include <cstdint>

#include "pn_internal.h"

namespace pn {

// ---- operand ranges of the fp16 two-plane kernels (pn_seqh.hip) --------------------------------------------------------
// Largest magnitudes, kept in DEVICE memory as the bit patterns of non-negative floats (atomicMax on uint32 orders them);
// every kernel derives its power-of-two operand scales from them when it runs -- no host round trip, capturable.
struct SeqRange {           // (= struct pn_seq_range of include/pathnet_hip.h: pn_pagg_range_offset hands its place to the caller)
    uint32_t x;             // max |Z| over the bank rows a call gathers from (the bank GEMM's epilogue, or range_rows_kernel)
    // how wide the rows' magnitudes are spread, from a SAMPLE of tiles of Z (32 x 32 or 256 consecutive values): sum and count
    // of the biased fp32 exponents of the sampled tiles' non-zero maxima.  exponent(x) - x_esum / x_cnt is how far the largest
    // value sits above the typical tile, in bits; the fp16 two-plane split keeps 22 bits for rows within ~2^18 of x and loses one
    // per factor of two below that.  Nothing in the library reads the two: the caller decides (pathnet_amd/modules.py falls back
    // to seq_math = bf16x3, whose three bf16 planes carry fp32's own exponent range, when the spread exceeds its window)
    int32_t x_esum;
    uint32_t x_cnt;
    uint32_t w_ih, w_hh;    // max |W_ih|, max |W_hh|                          (pack_fb_kernel, from range_part_kernel's partials)
    uint32_t dg;            // max |dG| of the BPTT launch the weight-gradient GEMM follows (seq_bwdh_kernel)
};

// tile geometry of the fp16 recurrent kernels (pn_seqh.hip "tile geometry"): n_big tiles of 32 paths and n_small tiles of
// small_rows (8 / 16 / 24; 0 = every tile has 32) -- the small ones after the big ones, or before them (small_first)
struct SeqTiling {
    int n_big, n_small, small_rows, small_first;
};
struct SeqTile {
    int q0, rows;
};

struct SeqFwdParams {
    const float *Z;         // [N*L, H] bank output (post activation)
    const int32_t *rowidx;  // [P, L]
    const int32_t *slotof;  // [P]
    const float *Wp;        // packed recurrent weights
    const float *biasc;     // [G*H]
    float *hn;              // [P, H] final hidden state per slot'
    float *saved;           // [P, L, SV, H]  SV = 5 (i,f,g,o,c) for LSTM, 1 (h_t) for RNN; may be null
    float *xh;              // [P, L, 2H]     the recurrent GEMM's input rows [x_t (after dropout) | h_{t-1}]:
                            //                the weight-gradient GEMM of the backward reads them back; may be null
    uint8_t *keep;          // [P, L, H/4]    built-in dropout: keep bits of columns 4c .. 4c+3 in bits 0-3 (the backward
                            //                reads them instead of re-drawing the Philox stream); may be null
    int P, L;               // slots of this launch (one micro-batch), path length
    int64_t Pmask;          // slots of the WHOLE batch: the dropout counters / the explicit mask are [L, Pmask, H]
    float p_drop;
    uint64_t seed;
    const pn_step_state *dyn;   // seed in device memory when set (hipGraph replay)
    const float *mask;      // [L, Pmask, H] explicit mask (reference order: original slot q) or null
    const SeqRange *range;  // fp16 kernels: operand ranges (device)
    float xmul;             // fp16 kernels: bound of the factor dropout applies to a gathered row (1 / (1 - p), or 16 for explicit masks)
    const float *ZW;        // seq_fwdzw_kernel: [rows of Z, G*H] = Z . W_ih^T + b (inference without dropout)
    SeqTiling tiling;       // fp16 kernels: filled in by the launcher
};

struct SeqBwdParams {
    const float *saved;     // [P, L, SV, H]
    const uint8_t *keep;    // [P, L, H/4] keep bits of the forward's built-in dropout, or null
    const float *dhn;       // [P, H]
    const int32_t *rowidx, *slotof;
    const float *WpT;
    float *dG;              // [P, L, G*H] pre-activation gate gradients (input of the weight-gradient GEMM)
    float *dZ;              // [N*L, H]    += d x_t   (atomic scatter: the backward of the row gather)
    int P, L;
    int merge0;             // step 0 scatters the W paths of a node into one table row (homo / PAGG index plans): add up runs first
    int64_t Pmask;          // slots of the whole batch (explicit mask [L, Pmask, H])
    float p_drop;
    uint64_t seed;
    const float *mask;
    SeqRange *range;        // fp16 kernels: reads w_ih / w_hh, leaves max |dG| in dg
    int store_dx;           // fp16 kernel, deterministic mode: rowidx is the identity -- mask * dx is STORED (no zero-fill, no atomics)
    SeqTiling tiling;       // fp16 kernels: filled in by the launcher
};

struct WgradParams {
    const float *dG;   // [R, GH]
    const float *xh;   // [R, 2H]
    int64_t R;
    int GH, H2;
    int64_t rows_per_split;
    float *part_w;     // [nsplit, GH, 2H]
    float *part_b;     // [nsplit, GH]
    const SeqRange *range;  // fp16 kernel: x (with xmul) and dg give the operand scales
    float xmul;
};

// ---- pn_rgrad.hip: C [M, N] += A^T . B over the rows of a large graph (node-level weight gradients) ----------------------
struct RgradParams {
    const float *A;         // [., lda]: reduction rows x M columns (d Z' or d Xh)
    const float *gate;      // same indexing as A, or null: an element of A counts where gate > 0 (ReLU backward)
    const float *B;         // [., ldb]: reduction rows x N columns (Xh or X)
    int64_t lda, ldb;
    int64_t R;              // reduction rows (an upper bound when seg is set)
    int M, N;               // multiples of 4
    float *C;               // [M, ldc] +=  (atomics)
    int64_t ldc;
    float *rowsum;          // [M] += column sums of the gated A (the bias gradient), or null
    const int32_t *seg, *list;      // compact rows: A row = seg[0] + k, B row = list[seg[0] + k], k < seg[1] - seg[0]; or both null
};
bool rgrad_pays(const pn_context *ctx, int64_t R, int M, int N);       // from 2 048 reduction rows on (pn_rgrad.hip)
int launch_rgrad(pn_context *ctx, void *stream, const RgradParams &p);

// ---- pn_seq3.hip: the recurrent kernels on the bf16 matrix pipe (six MFMAs per fp32 product, three planes) -----------------------
// every multiple of 32 up to 256 as hidden size; gc: 4 = LSTM, 1 = tanh RNN, 3 = GRU on the LSTM's four gate slots
// GRU rows of the caller's [3H, H] weights behind the four gate slots r, z, nx, nh
__host__ __device__ __forceinline__ int gru_weight_row(int slot, int j, int H) { return (slot < 2 ? slot : 2) * H + j; }
int launch_pack_fwd3(void *stream, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, int H, int G, int gru,
                     void *Wp, float *biasc);
int launch_pack_bwd3(void *stream, const float *w_ih, const float *w_hh, int H, int G, int gru, void *WpT);
int launch_seq_fwd3(pn_context *ctx, void *stream, int H, int gc, const SeqFwdParams &sp);
int launch_seq_bwd3(pn_context *ctx, void *stream, int H, int gc, const SeqBwdParams &sp);
// the weight-gradient GEMM's output tile and K tile (the caller lays out the K splits); nsplit partials to part_w / part_b
constexpr int WG_BM = 256, WG_BN = 256, WG_KT = 32;
int launch_wgrad3(pn_context *ctx, void *stream, const WgradParams &wp, int nsplit);
// g_W_ih / g_W_hh / g_b_* (+)= the nsplit partials of any of the weight-gradient GEMMs, added in split order
int launch_wgrad_reduce(void *stream, const float *part_w, const float *part_b, int nsplit, int GH, int H, int accumulate, int gru,
                        float *g_w_ih, float *g_w_hh, float *g_b_ih, float *g_b_hh);

// ---- pn_seq4.hip: the bf16 x 3 weight-gradient GEMM with two LDS stages (hidden size 128, four gate slots) ----------------
// SEQ4_WGRAD when the shape is the one it is built for and the context knob PN_SEQ4 has bit 2 set (default), else 0
enum { SEQ4_WGRAD = 4 };
int seq4_select(const pn_context *ctx, int H, int G, int L);
// the weight-gradient GEMM; nsplit row splits as laid out by the caller (part_w / part_b hold nsplit partials)
int launch_wgrad4(pn_context *ctx, void *stream, const WgradParams &wp, int nsplit);


// ---- pn_seqh.hip: the recurrent kernels on the fp16 matrix pipe (three MFMAs per fp32 product, two planes) ------------
// every multiple of 32 up to 256 as hidden size; gc: 4 = LSTM, 1 = tanh RNN, 3 = GRU on the LSTM's four gate slots
// range->x = max(range->x, |rows[r, :]|) over r < (count ? *count : rows); ordered after the launch_pack_fb that cleared it
int launch_range_rows(void *stream, const float *rows, int64_t nrows, int H, const int32_t *count, SeqRange *range);
// max |W_ih| / |W_hh| (RANGE_PARTS partial maxima in `part`, 256 bytes of the workspace; range->w_ih / w_hh stored for the
// recurrent kernels; range->dg and, with clear_x, range->x cleared for the launches that add to them) and both packings: the
// forward's fragments + summed biases (Wp, biasc) and, unless WpT is null, the BPTT's -- two launches on `stream`
int launch_pack_fb(void *stream, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, int H, int G, int Gw,
                   int gru, int clear_x, SeqRange *range, void *part, void *Wp, float *biasc, void *WpT);
int launch_seq_fwdh(pn_context *ctx, void *stream, int H, int gc, const SeqFwdParams &sp);
// inference forward over pre-projected rows (sp.ZW): only the W_hh half of the products remains
int launch_seq_fwdzw(pn_context *ctx, void *stream, int H, int gc, const SeqFwdParams &sp);
int launch_seq_bwdh(pn_context *ctx, void *stream, int H, int gc, const SeqBwdParams &sp);
int launch_wgradh(pn_context *ctx, void *stream, const WgradParams &wp, int H, int nsplit);

}  // namespace pn
