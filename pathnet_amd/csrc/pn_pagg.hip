// pn_pagg.hip -- the path aggregator ("PAGG") forward and backward on gfx950.
//
// Replaces the forward() bodies of
//   PathNet       /root/reference/PathNet_run.py:172-211   (variant HETERO)
//   PathNet_homo  /root/reference/PathNet_run.py:239-278   (variant HOMO)
//   PAGG          /root/reference/baseline/GPRGNN/src/copy.py:327-359   (variant PAGG)
// and the autograd backward the reference gets from loss.backward() (PathNet_run.py:351).
//
// How the reference's op sequence maps onto kernels (this file: the call's orchestration -- shapes, workspace, streams -- and
// the kernels that are neither GEMMs nor the recurrence; pn_gemm.hip, pn_seqh.hip / pn_seq3.hip hold those)
//   fc0 (+ReLU)                      :175 / :242-243      -> gemm_kernel (pn_gemm.hip)            Xh[N,H]
//   L Linear layers on P*L gathered rows, stack, select by distance code
//                                    :185-191 / :249-255  -> ONE gemm over the N nodes:
//        Z[v, d, :] = nets[d](Xh[v])  for every node v and code d  (N*L rows instead of P*L*L),
//        after which "gather + select" is a single row gather  Z[node(q,t)*L + code(q,t)].
//        Same dot products per emitted row, ~S*W*L/N times fewer FLOPs, no [P*L, L, H] temporary.
//   X[neis] gather, flip / view quirks :179-184 / :246     -> plan_kernel (index plan only)
//   dropout, nn.LSTM / nn.RNN          :194-195 / :264-265 -> seq_fwdh_kernel (pn_seqh.hip, fp16 x 2 planes, the default) or
//        seq_fwd3_kernel (pn_seq3.hip, bf16 x 3): coalesced row gather into an LDS path tile, MFMA for
//        [x_t ; h_{t-1}] x [W_ih ; W_hh]^T, cell math in registers, h_t back to LDS, L steps without leaving the CU;
//        hidden sizes above 256: the step-by-step recurrence gen_*_kernel below
//   attention over the W paths, mean, concat ego, dropout, fc2
//                                    :196-210 / :266-277  -> pool_fwd_kernel: one workgroup per node, wave shuffles for
//        the per-path dot products and the reduction over W
//   loss.backward()                    :351                -> pool_bwd_wg_kernel (pn_pagg_train_step: pool_step_kernel = pooling
//        forward + cross entropy + pooling backward of a node in one launch), seq_bwdh / seq_bwd3 (BPTT + atomic scatter of the
//        gather backward), wgradh / wgrad3 / wgrad4 (dG^T . [x|h]), gemm_kernel for the bank / fc0 gradients; deterministic
//        mode: det_scatter_kernel and fixed-order finishes instead of the atomics
//
// Precision: inputs, outputs, accumulation and all element-wise math are fp32.  The node-level GEMMs use the fp32-input MFMA
// (an exact FMA chain) or bf16 x 3 planes on large graphs; the three recurrent GEMMs evaluate every fp32 product on the 16-bit
// matrix pipe over exact plane splits of both operands (pn_kernels.h) -- the contract is 1e-5 on fp32 logits against the
// reference CPU path, tests measure 3e-7 against float64.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <type_traits>

#include "pn_gemm.h"
#include "pn_internal.h"
#include "pn_kernels.h"
#include "pn_seq.h"

using namespace pn;

// Settled by measurement in rounds 1-4 and no longer build options (profiles/HISTORY_r1_r4.md): descending tile order in the BPTT,
// strided K tiles in the weight gradient, the attention-weight terms through per-workgroup partials (atomics serialised), one row
// block per wave in the forward (two: 0.411 vs 0.308 ms), no s_setprio around the MFMA phases, the independent small launches on the
// context's second stream.
namespace {

// ================================================================================================
// index plan (the reference's view/flip/transposes collapsed into index arithmetic)
//   slot' = group * W + member is the order every later kernel uses (pooling groups contiguous).
//   original sequence slot q:  HOMO/PAGG q = slot'            (group = q / W)
//                              HETERO    q = member * S + group  (h_n.view(W, S, H), PathNet_run.py:196-197)
//   step t of slot q reads   HOMO/PAGG node ids[q, t]
//                            HETERO    r = q*L + t, node ids[r % P, L-1 - r / P]  (flip + reshape, :182-183)
//   and always the code codes[q, t] (:184 / :248).
// ================================================================================================
// A call may cover a slice of the batch (pn_pagg_shape: S_total, group_begin): `slot` below is the position in the WHOLE
// batch's slot' order, (group_begin + local group) * W + member; ids / codes are the whole batch's arrays.
struct PlanDims {
    int S_total, W, L, N;
    int64_t P_total;          // S_total * W
    int64_t row_base;         // batch position of row 0 of ids / codes (non-zero: the arrays hold a slice; HOMO / PAGG)
};
__device__ __forceinline__ void plan_entry(int variant, const int32_t *ids, const uint8_t *codes, const PlanDims &d,
                                           int64_t slot, int t, int64_t &q, int &node, int &code) {
    if (variant == PN_VARIANT_HETERO) {
        const int64_t g = slot / d.W, mem = slot - g * d.W;
        q = mem * d.S_total + g;
        const int64_t r = q * d.L + t;
        node = ids[(r % d.P_total) * d.L + (d.L - 1 - (int)(r / d.P_total))];
    } else {
        q = slot;
        node = ids[(q - d.row_base) * d.L + t];
    }
    code = codes[(q - d.row_base) * d.L + t];
}

// slots [slot_begin, slot_begin + count) of the batch -> rowidx / egoidx / slotof of the local slots 0 .. count-1
// rank: null (row of the full table: node * L + code) or the compact row of (code, node) -- compact_rows below
__global__ void plan_kernel(int variant, const int32_t *__restrict__ ids, const uint8_t *__restrict__ codes, PlanDims d,
                            int64_t slot_begin, int64_t count, const int32_t *__restrict__ rank, int32_t *__restrict__ rowidx,
                            int32_t *__restrict__ egoidx, int32_t *__restrict__ slotof) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * d.L) return;
    const int64_t ls = i / d.L;
    const int t = (int)(i - ls * d.L);
    int64_t q;
    int node, code;
    plan_entry(variant, ids, codes, d, slot_begin + ls, t, q, node, code);
    node = min(max(node, 0), d.N - 1);
    code = min(code, d.L - 1);
    const int32_t row = rank ? rank[(int64_t)code * d.N + node] : node * d.L + code;
    rowidx[i] = row;
    if (t == 0) {
        slotof[ls] = (int32_t)q;      // position in the whole batch: what the dropout counters / explicit masks index
        // attention ego: HOMO uses the transformed row of (q, step 0) (ego_full, :259-260);
        // HETERO uses the untransformed Xh row of path q's first node (neis[0], :199)
        egoidx[ls] = variant == PN_VARIANT_HETERO ? min(max(ids[(q - d.row_base) * d.L], 0), d.N - 1) : row;
    }
}

// ---- touched-row compaction ---------------------------------------------------------------------------------------------
// The distance bank is applied to (node, code) rows, N * L of them, whatever the batch reads (bank-before-gather, DESIGN.md
// section 4).  When the batch's path steps cannot touch half of them (a 10 M-node graph with 100 000 masked nodes: 24 M path
// steps, 60 M rows) the bank, its backward and the tables Z / dZ themselves shrink to the rows that ARE touched:
//   flags[code * N + node] = 1 for every path step (mark_rows_kernel), exclusive scan -> rank, the compact row of every
//   touched (code, node); list[compact row] = node; seg[c] = first compact row of code c (the rows are code-major, so each
//   code's rows are one contiguous GEMM against that code's weights); the index plan writes compact rows.
__global__ void mark_rows_kernel(int variant, const int32_t *__restrict__ ids, const uint8_t *__restrict__ codes, PlanDims d,
                                 int64_t slot_begin, int64_t count, uint8_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * d.L) return;
    const int64_t ls = i / d.L;
    int64_t q;
    int node, code;
    plan_entry(variant, ids, codes, d, slot_begin + ls, (int)(i - ls * d.L), q, node, code);
    flags[(int64_t)min(code, d.L - 1) * d.N + min(max(node, 0), d.N - 1)] = 1;
}
constexpr int SCAN_BLOCK = 256 * 16;        // flags per workgroup
__global__ __launch_bounds__(256) void scan_count_kernel(const uint8_t *__restrict__ flags, int64_t n, int32_t *__restrict__ bsum) {
    __shared__ int red[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 16;
    int c = 0;
    if (base + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(flags + base);
        c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);        // (flags are 0 / 1 bytes)
    } else {
        for (int k = 0; k < 16; k++) c += base + k < n ? flags[base + k] : 0;
    }
    c = (int)wave_sum((float)c);        // <= 1024 per wave: exact in fp32
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// exclusive scan of the block counts in place (one workgroup walks them: <= 15 k blocks for 60 M rows); total -> seg[L]
__global__ __launch_bounds__(1024) void scan_blocks_kernel(int32_t *__restrict__ bsum, int nb, int32_t *__restrict__ total) {
    __shared__ int part[1024];
    const int per = (nb + 1023) / 1024, lo = threadIdx.x * per, hi = min(nb, lo + per);
    int s = 0;
    for (int i = lo; i < hi; i++) s += bsum[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 1024; i++) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; i++) {
        const int v = bsum[i];
        bsum[i] = run;
        run += v;
    }
}
__global__ __launch_bounds__(256) void scan_rank_kernel(const uint8_t *__restrict__ flags, int64_t n, int N, int L,
                                                        const int32_t *__restrict__ bsum, int32_t *__restrict__ rank,
                                                        int32_t *__restrict__ list, int32_t *__restrict__ seg) {
    __shared__ int wsum[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 16;
    uint8_t f[16];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        f[k] = base + k < n ? flags[base + k] : 0;
        c += f[k];
    }
    // exclusive scan of the 256 per-thread counts: inside the wave by shuffles, across the four waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int run = bsum[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int64_t i = base + k;
        if (i < n) {
            rank[i] = run;
            const int code = (int)(i / N), node = (int)(i - (int64_t)code * N);
            if (node == 0) seg[code] = run;         // first row of a code (rows are code-major)
            if (f[k]) list[run] = node;
            run += f[k];
        }
    }
}

// rows[slot', t, :] = table[rowidx, :]   (stand-alone gather, also the HBM-roofline microbenchmark)
template <int VEC>
__global__ __launch_bounds__(256) void gather_kernel(int variant, const float *__restrict__ table,
                                                     const int32_t *__restrict__ ids,
                                                     const uint8_t *__restrict__ codes, PlanDims d, int64_t slot_begin,
                                                     int64_t count, int H, float *__restrict__ rows) {
    const int hv = H / VEC;
    const int64_t total = count * d.L * hv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / hv;
        const int c = (int)(i - row * hv);
        const int64_t ls = row / d.L;
        const int t = (int)(row - ls * d.L);
        int64_t q;
        int node, code;
        plan_entry(variant, ids, codes, d, slot_begin + ls, t, q, node, code);
        node = min(max(node, 0), d.N - 1);
        code = min(code, d.L - 1);
        const int64_t src = ((int64_t)node * d.L + code) * hv + c;
        if (VEC == 4)
            reinterpret_cast<float4 *>(rows)[i] = reinterpret_cast<const float4 *>(table)[src];
        else
            rows[i] = table[src];
    }
}

// ---- order-agnostic path encoders of the ablation ("mean" / "sum" instead of the recurrent cell): h_n of a path is the
//      mean (sum) over its L steps of the dropped-out rows the recurrent kernels would consume.  One thread per (path,
//      4 columns); the backward scatters scale * mask * d h_n back onto the L table rows.
struct SeqReduceParams {
    const float *Z;          // [N*L, H]
    const int32_t *rowidx;   // [P, L]
    const int32_t *slotof;   // [P]
    float *hn;               // [P, H]   forward output
    const float *dhn;        // [P, H]   backward input
    float *dZ;               // [N*L, H] backward output (atomic scatter)
    int P, L, H;
    int64_t Pmask;
    float scale;             // 1/L (mean) or 1 (sum)
    float p_drop;
    uint64_t seed;
    const pn_step_state *dyn;
    const float *mask;       // [L, Pmask, H] or null
};

template <bool BACKWARD>
__global__ __launch_bounds__(256) void seq_reduce_kernel(SeqReduceParams p) {
    const int hv = p.H / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.P * hv) return;
    const int q = (int)(i / hv), c4 = (int)(i - (int64_t)q * hv);
    const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
    const int64_t slot = p.slotof[q];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BACKWARD) {
        g = reinterpret_cast<const float4 *>(p.dhn)[i];
        g.x *= p.scale; g.y *= p.scale; g.z *= p.scale; g.w *= p.scale;
    }
    for (int t = 0; t < p.L; t++) {
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.mask)
            m = reinterpret_cast<const float4 *>(p.mask)[((int64_t)t * p.Pmask + slot) * hv + c4];
        else if (p.p_drop > 0.0f)
            m = dropout4(seed, ((uint64_t)t * p.Pmask + slot) * hv + c4, 1u, p.p_drop);     // the recurrent kernels' counters
        const size_t row = (size_t)(uint32_t)p.rowidx[(int64_t)q * p.L + t];
        if (BACKWARD) {
            float *d = p.dZ + row * p.H + 4 * c4;
            atomicAdd(d + 0, g.x * m.x); atomicAdd(d + 1, g.y * m.y); atomicAdd(d + 2, g.z * m.z); atomicAdd(d + 3, g.w * m.w);
        } else {
            const float4 v = reinterpret_cast<const float4 *>(p.Z)[row * hv + c4];
            acc.x += v.x * m.x; acc.y += v.y * m.y; acc.z += v.z * m.z; acc.w += v.w * m.w;
        }
    }
    if (!BACKWARD) {
        acc.x *= p.scale; acc.y *= p.scale; acc.z *= p.scale; acc.w *= p.scale;
        reinterpret_cast<float4 *>(p.hn)[i] = acc;
    }
}


// ================================================================================================
// Generic recurrence for hidden sizes the fused kernels do not cover (256 < H <= 1024, e.g. the reference's -hid=1024
// runs, results/result_for_Nba.txt): step by step -- gather + dropout kernel, one fp32 MFMA GEMM for the gate
// pre-activations, an element-wise cell kernel; backward the same way round plus one GEMM for the weight gradients.
// Same tensors and layouts as the fused path (saved [P,L,SV,H], xh [P,L,2H], dG [P,L,GH]), same dropout counters, so
// pooling, micro-batches, slices and the caller see no difference.  A functional path, ~10 launches per step.
// ================================================================================================
struct GenParams {
    const float *Z;            // [N*L, H]
    const int32_t *rowidx;     // [P, L]
    const int32_t *slotof;     // [P]
    float *xh;                 // [P, L, 2H]
    float *pre;                // [P, L, GH]  forward: gate pre-activations of step t; backward: dG
    float *saved;              // [P, L, SV, H] or null
    float *state;              // [P, H] running cell / hidden state (forward), d c / the GRU's direct term (backward)
    float *hn;                 // [P, H]
    float *dh;                 // [P, H] backward: d loss / d h_t, updated in place
    const float *gx;           // [P, 2H] backward: [dx_t | dh_{t-1}] of the step's GEMM
    float *dZ;
    const float *biasc;
    int P, L, H, G, cell, t;
    int64_t Pmask;
    float p_drop;
    uint64_t seed;
    const pn_step_state *dyn;
    const float *mask;
};

__device__ __forceinline__ float4 gen_mask(const GenParams &p, uint64_t seed, int t, int64_t slot, int c4) {
    const int hv = p.H / 4;
    if (p.mask) return reinterpret_cast<const float4 *>(p.mask)[((int64_t)t * p.Pmask + slot) * hv + c4];
    if (p.p_drop > 0.0f) return dropout4(seed, ((uint64_t)t * p.Pmask + slot) * hv + c4, 1u, p.p_drop);
    return make_float4(1.f, 1.f, 1.f, 1.f);
}

// xh[q, t, 0:H] = mask * Z[row(q, t)];  t = 0 also clears the h half
__global__ __launch_bounds__(256) void gen_x_kernel(GenParams p) {
    const int hv = p.H / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.P * hv) return;
    const int q = (int)(i / hv), c4 = (int)(i - (int64_t)q * hv);
    const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
    const float4 m = gen_mask(p, seed, p.t, p.slotof[q], c4);
    const size_t row = (size_t)(uint32_t)p.rowidx[(int64_t)q * p.L + p.t];
    float4 v = reinterpret_cast<const float4 *>(p.Z)[row * hv + c4];
    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
    float4 *dst = reinterpret_cast<float4 *>(p.xh) + ((size_t)q * p.L + p.t) * (2 * hv) + c4;
    dst[0] = v;
    if (p.t == 0) dst[hv] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// one element (path q, hidden unit j) of the cell update of step t
__global__ __launch_bounds__(256) void gen_cell_fwd_kernel(GenParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.P * p.H) return;
    const int H = p.H, q = (int)(i / H), j = (int)(i - (int64_t)q * H);
    const float *pre = p.pre + ((size_t)q * p.L + p.t) * ((size_t)p.G * H) + j;
    const int SV = p.G == 4 ? 5 : 1;
    float *sv = p.saved ? p.saved + (((size_t)q * p.L + p.t) * SV) * H + j : nullptr;
    float h;
    if (p.cell == 3) {                     // GRU on the four slots r, z, nx, nh
        const float rg = sigmoidf_(pre[0]), zg = sigmoidf_(pre[H]), nh = pre[3 * (size_t)H];
        const float ng = tanhf_(pre[2 * (size_t)H] + rg * nh);
        const float hp = p.t > 0 ? p.state[i] : 0.0f;
        h = (1.0f - zg) * ng + zg * hp;
        p.state[i] = h;
        if (sv) { sv[0] = rg; sv[H] = zg; sv[2 * (size_t)H] = ng; sv[3 * (size_t)H] = nh; sv[4 * (size_t)H] = hp; }
    } else if (p.G == 4) {                 // LSTM
        const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[H]), gg = tanhf_(pre[2 * (size_t)H]), og = sigmoidf_(pre[3 * (size_t)H]);
        const float c = fg * (p.t > 0 ? p.state[i] : 0.0f) + ig * gg;
        p.state[i] = c;
        h = og * tanhf_(c);
        if (sv) { sv[0] = ig; sv[H] = fg; sv[2 * (size_t)H] = gg; sv[3 * (size_t)H] = og; sv[4 * (size_t)H] = c; }
    } else {                                // tanh RNN
        h = tanhf_(pre[0]);
        if (sv) sv[0] = h;
    }
    if (p.t == p.L - 1)
        p.hn[i] = h;
    else
        p.xh[((size_t)q * p.L + p.t + 1) * (2 * (size_t)H) + H + j] = h;
}

// cell backward of step t: dh (in place buffer) and the carried state -> the gate-slot gradients dG[q, t, :]
__global__ __launch_bounds__(256) void gen_cell_bwd_kernel(GenParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.P * p.H) return;
    const int H = p.H, q = (int)(i / H), j = (int)(i - (int64_t)q * H);
    const int SV = p.G == 4 ? 5 : 1;
    const float *sv = p.saved + (((size_t)q * p.L + p.t) * SV) * H + j;
    float *d = p.pre + ((size_t)q * p.L + p.t) * ((size_t)p.G * H) + j;
    const float dhv = p.dh[i];
    if (p.cell == 3) {
        const float rg = sv[0], zg = sv[H], ng = sv[2 * (size_t)H], nh = sv[3 * (size_t)H], hp = sv[4 * (size_t)H];
        const float dnp = dhv * (1.0f - zg) * (1.0f - ng * ng);
        d[0] = dnp * nh * rg * (1.0f - rg);
        d[H] = dhv * (hp - ng) * zg * (1.0f - zg);
        d[2 * (size_t)H] = dnp;
        d[3 * (size_t)H] = dnp * rg;
        p.state[i] = dhv * zg;              // the direct path d h_t / d h_{t-1}
    } else if (p.G == 4) {
        const float ig = sv[0], fg = sv[H], gg = sv[2 * (size_t)H], og = sv[3 * (size_t)H], c = sv[4 * (size_t)H];
        const float cprev = p.t > 0 ? sv[-(ptrdiff_t)H] : 0.0f;       // slot 4 of step t-1
        const float tc = tanhf_(c);
        const float dct = (p.t < p.L - 1 ? p.state[i] : 0.0f) + dhv * og * (1.0f - tc * tc);
        d[0] = dct * gg * ig * (1.0f - ig);
        d[H] = dct * cprev * fg * (1.0f - fg);
        d[2 * (size_t)H] = dct * ig * (1.0f - gg * gg);
        d[3 * (size_t)H] = dhv * tc * og * (1.0f - og);
        p.state[i] = dct * fg;
    } else {
        const float h = sv[0];
        d[0] = dhv * (1.0f - h * h);
    }
}

// after the step's GEMM gx = dG_t . [W_ih | W_hh]: scatter mask * dx into dZ, dh_{t-1} into the dh buffer
__global__ __launch_bounds__(256) void gen_scatter_kernel(GenParams p) {
    const int hv = p.H / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.P * hv) return;
    const int q = (int)(i / hv), c4 = (int)(i - (int64_t)q * hv);
    const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
    const float4 m = gen_mask(p, seed, p.t, p.slotof[q], c4);
    const float4 *g = reinterpret_cast<const float4 *>(p.gx) + (size_t)q * (2 * hv) + c4;
    const float4 dx = g[0];
    const size_t row = (size_t)(uint32_t)p.rowidx[(int64_t)q * p.L + p.t];
    float *dz = p.dZ + row * p.H + 4 * c4;
    atomicAdd(dz + 0, dx.x * m.x); atomicAdd(dz + 1, dx.y * m.y); atomicAdd(dz + 2, dx.z * m.z); atomicAdd(dz + 3, dx.w * m.w);
    if (p.t > 0) {
        float4 dh = g[hv];
        if (p.cell == 3) {
            const float4 direct = reinterpret_cast<const float4 *>(p.state)[i];
            dh.x += direct.x; dh.y += direct.y; dh.z += direct.z; dh.w += direct.w;
        }
        reinterpret_cast<float4 *>(p.dh)[i] = dh;
    }
}

// Wcat [GH, 2H] = [W_ih | W_hh] in fp32 (GRU: the four slots with their zero halves) and its transpose, biasc as in
// pack_fwd3_kernel
__global__ void gen_pack_kernel(const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                const float *__restrict__ b_ih, const float *__restrict__ b_hh, int H, int G, int gru,
                                float *__restrict__ Wcat, float *__restrict__ WcatT, float *__restrict__ biasc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)G * H) {
        if (!gru) {
            biasc[i] = b_ih[i] + b_hh[i];
        } else {
            const int slot = (int)(i / H), j = (int)(i - (int64_t)slot * H), wr = gru_weight_row(slot, j, H);
            biasc[i] = slot < 2 ? b_ih[wr] + b_hh[wr] : slot == 2 ? b_ih[wr] : b_hh[wr];
        }
    }
    if (i >= (int64_t)G * H * 2 * H) return;
    const int m = (int)(i / (2 * H)), k = (int)(i - (int64_t)m * 2 * H), slot = m / H, j = m - slot * H;
    const int row = gru ? gru_weight_row(slot, j, H) : m;
    float v = k < H ? w_ih[(int64_t)row * H + k] : w_hh[(int64_t)row * H + (k - H)];
    if (gru && ((slot == 2 && k >= H) || (slot == 3 && k < H))) v = 0.0f;
    Wcat[i] = v;
    if (WcatT) WcatT[(int64_t)k * G * H + m] = v;      // [2H, GH]: the BPTT's GEMM wants its B operand K-contiguous too
}

// (SeqFwdParams: pn_seq.h)

// ================================================================================================
// pool_fwd_kernel: one workgroup (4 waves) per pooling group (= output node).
// ================================================================================================
struct PoolParams {
    int variant, S, W, H, C, N;      // N: rows of Xh (sel is clamped to it, like ids in plan_kernel)
    int64_t goff;           // position of local group 0 in the whole batch (dropout counters / explicit mask rows)
    const float *hn;        // [P, H] in slot' order
    const float *ego_tab;   // Z (HOMO) or Xh (HETERO)
    const int32_t *egoidx;  // [P] row of ego_tab
    const float *Xh;        // [N, H]
    const int32_t *sel;     // [S]
    const float *att_w, *att_b, *fc2_w, *fc2_b;
    float p_drop;
    uint64_t seed;
    const pn_step_state *dyn;
    const float *mask;      // [S, 2H] or null
    float *coef;            // [P] pooling coefficient per slot' (1+att | softmax | 1)
    float *rawsc;           // [P] raw attention score (pre LeakyReLU) per slot'
    float *layer1;          // [S, 2H] classifier input after dropout
    float *out;             // [S, C]
};

__device__ __forceinline__ void pool_fwd_body(const PoolParams &p, float *lds) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = blockIdx.x, H = p.H, W = p.W;
    float *sc = lds;                                            // [W] scores, then coefficients
    int *s_erow = reinterpret_cast<int *>(sc + W);              // [W] ego rows of the group's members
    float *part4 = reinterpret_cast<float *>(s_erow + W);       // [4][H] the waves' partial pooled sums
    float *l1s = part4 + 4 * H;                                 // [2H] classifier input
    const float inv_w = 1.0f / (float)W;

    if (p.variant != PN_VARIANT_PAGG) {
        // the ego rows, fetched up front: the score loop then has no index -> row dependent load pair
        for (int mem = tid; mem < W; mem += 256) s_erow[mem] = p.egoidx[(int64_t)g * W + mem];
        __syncthreads();
        // attention scores, 8 members per wave at a time: lane = (member lane>>3, eighth of H lane&7); the eight partial
        // dot products of a member are summed with three xor-shuffles
        const float ab = p.att_b[0];
        const int m8 = lane >> 3, part = lane & 7, jw = H / 8;
        for (int m0 = 8 * wave; m0 < W; m0 += 32) {
            const int mem = m0 + m8;
            const int memc = min(mem, W - 1);
            const int64_t s = (int64_t)g * W + memc;
            const float4 *h4 = reinterpret_cast<const float4 *>(p.hn + s * H + part * jw);
            const float4 *e4 = reinterpret_cast<const float4 *>(p.ego_tab + (int64_t)s_erow[memc] * H + part * jw);
            const float4 *a4 = reinterpret_cast<const float4 *>(p.att_w + part * jw);
            const float4 *b4 = reinterpret_cast<const float4 *>(p.att_w + H + part * jw);
            float acc = 0.0f;
            for (int j = 0; j < jw / 4; j++) {
                const float4 hv = h4[j], ev = e4[j], av = a4[j], bv = b4[j];
                acc += hv.x * av.x + hv.y * av.y + hv.z * av.z + hv.w * av.w;
                acc += ev.x * bv.x + ev.y * bv.y + ev.z * bv.z + ev.w * bv.w;
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (part == 0 && mem < W) {
                sc[mem] = acc + ab;
                p.rawsc[(int64_t)g * W + mem] = acc + ab;
            }
        }
        __syncthreads();
    }
    if (p.variant == PN_VARIANT_HETERO) {
        // softmax over the W members of LeakyReLU(score) (F.softmax implicit dim 0 of [W,S,1]); every wave reduces
        // all W scores (same order, same result), then each thread rewrites its own entries
        float mx = -3.4e38f;
        for (int mem = lane; mem < W; mem += 64) {
            float v = sc[mem];
            v = v > 0.0f ? v : 0.01f * v;
            mx = fmaxf(mx, v);
        }
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int mem = lane; mem < W; mem += 64) {
            float v = sc[mem];
            v = v > 0.0f ? v : 0.01f * v;
            sum += expf(v - mx);
        }
        sum = wave_sum(sum);
        __syncthreads();
        for (int mem = tid; mem < W; mem += 256) {
            float v = sc[mem];
            v = v > 0.0f ? v : 0.01f * v;
            sc[mem] = expf(v - mx) / sum;
        }
    } else if (p.variant == PN_VARIANT_HOMO) {
        for (int mem = tid; mem < W; mem += 256) sc[mem] = 1.0f + sc[mem];
    } else {
        for (int mem = tid; mem < W; mem += 256) sc[mem] = 1.0f;
    }
    __syncthreads();
    for (int mem = tid; mem < W; mem += 256) p.coef[(int64_t)g * W + mem] = sc[mem];

    // pooled = mean_w coef_w * h_w: each wave sums its quarter of the members, the quarters meet in LDS (fixed order)
    {
        const int per = (W + 3) / 4, mem_end = min(W, (wave + 1) * per);
        for (int j = lane; j < H; j += 64) {
            float acc = 0.0f;
#pragma unroll 8
            for (int mem = wave * per; mem < mem_end; mem++) acc += sc[mem] * p.hn[((int64_t)g * W + mem) * H + j];
            part4[wave * H + j] = acc;
        }
    }
    __syncthreads();
    // layer1 = dropout([Xh[sel[g]] ; pooled])
    float *l1 = p.layer1 + (int64_t)g * 2 * H;
    const float *ego = p.Xh + (int64_t)min(max(p.sel[g], 0), p.N - 1) * H;
    for (int j = tid; j < H; j += 256) {
        float a = ego[j], b = (part4[j] + part4[H + j] + part4[2 * H + j] + part4[3 * H + j]) * inv_w;
        const uint64_t gg = (uint64_t)(p.goff + g);
        if (p.mask) {
            a *= p.mask[gg * 2 * H + j];
            b *= p.mask[gg * 2 * H + H + j];
        } else if (p.p_drop > 0.0f) {
            const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
            const float4 m0 = dropout4(seed, (gg * 2 * H + j) >> 2, 2u, p.p_drop);
            const float4 m1 = dropout4(seed, (gg * 2 * H + H + j) >> 2, 2u, p.p_drop);
            const int e0 = j & 3;
            a *= e0 == 0 ? m0.x : e0 == 1 ? m0.y : e0 == 2 ? m0.z : m0.w;
            b *= e0 == 0 ? m1.x : e0 == 1 ? m1.y : e0 == 2 ? m1.z : m1.w;
        }
        l1[j] = a;
        l1[H + j] = b;
        l1s[j] = a;
        l1s[H + j] = b;
    }
    __syncthreads();
    for (int c = wave; c < p.C; c += 4) {
        float part = 0.0f;
        for (int j = lane; j < 2 * H; j += 64) part += l1s[j] * p.fc2_w[(int64_t)c * 2 * H + j];
        part = wave_sum(part);
        if (lane == 0) p.out[(int64_t)g * p.C + c] = part + p.fc2_b[c];
    }
}
__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolParams p) {
    extern __shared__ float lds[];
    pool_fwd_body(p, lds);
}


// ================================================================================================
// BACKWARD
// ================================================================================================
// ---- pooling / attention / classifier backward: one wavefront per group -------------------------
struct PoolBwdParams {
    int variant, S, W, H, C, N;
    int64_t goff;
    const float *hn, *ego_tab;
    const int32_t *egoidx, *sel;
    const float *att_w, *fc2_w, *g_out, *coef, *rawsc;
    float p_drop;
    uint64_t seed;
    const pn_step_state *dyn;
    const float *mask;
    float *dhn;        // [P, H]
    float *dXh;        // [N, H]  (+= ego of the classifier input; HETERO: += attention ego)
    float *dego;       // table the attention-ego gradient goes to: dZ (HOMO) or dXh (HETERO)
    float *g_att_w, *g_att_b;
    // deterministic mode (all three set, else null): instead of atomics the kernel stores the ego half of d layer1 per
    // group, d score per path and the attention-weight terms per workgroup; det_scatter_kernel / det_att_reduce_kernel add
    float *det_sel;    // [S, H]
    float *det_ds;     // [P]
    float *det_att;    // [workgroups][2H + 4]
};

// HI: 64-column chunks of H a lane walks (4 covers H <= 256, the fused kernels' range; 16 covers the generic path's H <= 1024)
template <int HI>
__global__ __launch_bounds__(256) void pool_bwd_kernel(PoolBwdParams p) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int H = p.H, W = p.W;
    float *dco = lds + wave * (2 * W + H);   // [W] d coef, then [W] d score, then [H] d pooled / W
    float *dsc = dco + W;
    float *dp = dsc + W;
    float *red = lds + 4 * (2 * W + H);      // [4][2H] per-wave attention-weight partials
    int *s_erow = reinterpret_cast<int *>(lds + 4 * (2 * W + H) + 8 * H) + wave * W;   // [4][W] ego rows of the group
    float *s_coef = lds + 4 * (2 * W + H) + 8 * H + 4 * W + wave * W;                  // [4][W]
    const int g = blockIdx.x * 4 + wave;
    const bool active = g < p.S;
    const float inv_w = 1.0f / (float)W;
    float gaw_h[HI], gaw_e[HI], gab = 0.0f;
#pragma unroll
    for (int i = 0; i < HI; i++) gaw_h[i] = gaw_e[i] = 0.0f;

    if (active) {
        for (int mem = lane; mem < W; mem += 64) {
            const int64_t s = (int64_t)g * W + mem;
            s_erow[mem] = p.variant == PN_VARIANT_PAGG ? 0 : p.egoidx[s];
            s_coef[mem] = p.coef[s];
        }
    }
    if (active) {
        for (int j = lane; j < H; j += 64) {
            float a = 0.0f, b = 0.0f;
            for (int c = 0; c < p.C; c++) {
                const float go = p.g_out[(int64_t)g * p.C + c];
                a += go * p.fc2_w[(int64_t)c * 2 * H + j];
                b += go * p.fc2_w[(int64_t)c * 2 * H + H + j];
            }
            const uint64_t gg = (uint64_t)(p.goff + g);
            if (p.mask) {
                a *= p.mask[gg * 2 * H + j];
                b *= p.mask[gg * 2 * H + H + j];
            } else if (p.p_drop > 0.0f) {
                const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
                a *= dropout1(seed, gg * 2 * H + j, 2u, p.p_drop);
                b *= dropout1(seed, gg * 2 * H + H + j, 2u, p.p_drop);
            }
            if (p.det_sel)
                p.det_sel[(int64_t)g * H + j] = a;
            else
                atomicAdd(&p.dXh[(int64_t)min(max(p.sel[g], 0), p.N - 1) * H + j], a);
            dp[j] = b * inv_w;
        }
        __builtin_amdgcn_wave_barrier();
        {
            const int m8 = lane >> 3, part = lane & 7, jw = H / 8;
            for (int m0 = 0; m0 < W; m0 += 8) {
                const int mem = m0 + m8;
                const float4 *h4 =
                    reinterpret_cast<const float4 *>(p.hn + ((int64_t)g * W + min(mem, W - 1)) * H + part * jw);
                float acc = 0.0f;
                for (int j = 0; j < jw / 4; j++) {
                    const float4 hv = h4[j];
                    const float *d = dp + part * jw + 4 * j;
                    acc += hv.x * d[0] + hv.y * d[1] + hv.z * d[2] + hv.w * d[3];
                }
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                acc += __shfl_xor(acc, 4, 64);
                if (part == 0 && mem < W) dco[mem] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (p.variant == PN_VARIANT_HETERO) {
            float tot = 0.0f;
            for (int mem = lane; mem < W; mem += 64) tot += p.coef[(int64_t)g * W + mem] * dco[mem];
            tot = wave_sum(tot);
            for (int mem = lane; mem < W; mem += 64) {
                const int64_t s = (int64_t)g * W + mem;
                dsc[mem] = p.coef[s] * (dco[mem] - tot) * (p.rawsc[s] > 0.0f ? 1.0f : 0.01f);
            }
        } else {
            for (int mem = lane; mem < W; mem += 64) dsc[mem] = p.variant == PN_VARIANT_HOMO ? dco[mem] : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
        // The attention-ego gradient of consecutive members usually lands on the same table row (all W paths of a
        // node start at that node): it is accumulated in registers and flushed with one atomic per row change.
        int64_t cur_row = -1;
        float ego_acc[HI];
#pragma unroll
        for (int i = 0; i < HI; i++) ego_acc[i] = 0.0f;
        auto flush = [&]() {
            if (cur_row < 0 || p.det_ds) return;
#pragma unroll
            for (int i = 0; i < HI; i++) {
                const int j = lane + 64 * i;
                if (j < H) atomicAdd(&p.dego[cur_row + j], ego_acc[i]);
                ego_acc[i] = 0.0f;
            }
        };
#pragma unroll 4
        for (int mem = 0; mem < W; mem++) {
            const int64_t s = (int64_t)g * W + mem;
            const float ds = dsc[mem], cf = s_coef[mem];
            const int64_t erow = (int64_t)s_erow[mem] * H;
            if (p.variant != PN_VARIANT_PAGG && erow != cur_row) {   // wave-uniform
                flush();
                cur_row = erow;
            }
#pragma unroll
            for (int i = 0; i < HI; i++) {
                const int j = lane + 64 * i;
                if (j < H) {
                    float dh = cf * dp[j];
                    if (p.variant != PN_VARIANT_PAGG) {
                        dh += ds * p.att_w[j];
                        gaw_h[i] += ds * p.hn[s * H + j];
                        gaw_e[i] += ds * p.ego_tab[erow + j];
                        ego_acc[i] += ds * p.att_w[H + j];
                    }
                    p.dhn[s * H + j] = dh;
                }
            }
            gab += ds;
            if (p.det_ds && lane == 0) p.det_ds[s] = ds;
        }
        if (p.variant != PN_VARIANT_PAGG) flush();
    }
    if (p.variant == PN_VARIANT_PAGG) return;   // block-uniform
#pragma unroll
    for (int i = 0; i < HI; i++) {
        const int j = lane + 64 * i;
        if (j < H) {
            red[wave * 2 * H + j] = gaw_h[i];
            red[wave * 2 * H + H + j] = gaw_e[i];
        }
    }
    __syncthreads();
    if (p.det_att) {
        float *out = p.det_att + (int64_t)blockIdx.x * (2 * H + 4);
        for (int j = threadIdx.x; j < 2 * H; j += 256) out[j] = red[j] + red[2 * H + j] + red[4 * H + j] + red[6 * H + j];
        if (lane == 0) out[2 * H + wave] = active ? gab : 0.0f;
        return;
    }
    for (int j = threadIdx.x; j < 2 * H; j += 256)
        atomicAdd(&p.g_att_w[j], red[j] + red[2 * H + j] + red[4 * H + j] + red[6 * H + j]);
    if (lane == 0 && active) atomicAdd(p.g_att_b, gab);
}

// ---- the same with a workgroup per group: the W members of a group split over the four waves.  One wave per group runs
//      its ~50 dependent memory round trips alone on its SIMD (1 299 waves on 1 024 SIMDs at the headline shape); four
//      times the waves hide them.  (Round 2 had rejected this layout for quadrupling the atomics on the 2H + 1
//      attention-weight addresses; those now go through per-workgroup partials, det_att: [groups][2H + 4].)
template <int HI>
__device__ __forceinline__ void pool_bwd_wg_body(const PoolBwdParams &p, float *lds) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int H = p.H, W = p.W;
    float *dco = lds;                        // [W] d coef
    float *dsc = dco + W;                    // [W] d score
    float *dp = dsc + W;                     // [H] d pooled / W
    float *red = dp + H;                     // [4][2H] per-wave attention-weight partials; before that [4][H] ego partials
    int *s_erow = reinterpret_cast<int *>(red + 8 * H);    // [W] ego rows of the group
    float *s_coef = reinterpret_cast<float *>(s_erow + W);  // [W]
    float *s_tot = s_coef + W;               // [4] hetero: partial sums of coef * d coef
    const int g = blockIdx.x;
    const float inv_w = 1.0f / (float)W;
    for (int mem = tid; mem < W; mem += 256) {
        const int64_t s = (int64_t)g * W + mem;
        s_erow[mem] = p.variant == PN_VARIANT_PAGG ? 0 : p.egoidx[s];
        s_coef[mem] = p.coef[s];
    }
    for (int j = tid; j < H; j += 256) {
        float a = 0.0f, b = 0.0f;
        for (int c = 0; c < p.C; c++) {
            const float go = p.g_out[(int64_t)g * p.C + c];
            a += go * p.fc2_w[(int64_t)c * 2 * H + j];
            b += go * p.fc2_w[(int64_t)c * 2 * H + H + j];
        }
        const uint64_t gg = (uint64_t)(p.goff + g);
        if (p.mask) {
            a *= p.mask[gg * 2 * H + j];
            b *= p.mask[gg * 2 * H + H + j];
        } else if (p.p_drop > 0.0f) {
            const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
            a *= dropout1(seed, gg * 2 * H + j, 2u, p.p_drop);
            b *= dropout1(seed, gg * 2 * H + H + j, 2u, p.p_drop);
        }
        if (p.det_sel)
            p.det_sel[(int64_t)g * H + j] = a;
        else
            atomicAdd(&p.dXh[(int64_t)min(max(p.sel[g], 0), p.N - 1) * H + j], a);
        dp[j] = b * inv_w;
    }
    __syncthreads();
    {       // d coef[mem] = hn[mem] . d pooled / W: eight lanes per member, 32 members per pass
        const int m32 = tid >> 3, part = tid & 7, jw = H / 8;
        for (int m0 = 0; m0 < W; m0 += 32) {
            const int mem = m0 + m32;
            const float4 *h4 = reinterpret_cast<const float4 *>(p.hn + ((int64_t)g * W + min(mem, W - 1)) * H + part * jw);
            float acc = 0.0f;
            for (int j = 0; j < jw / 4; j++) {
                const float4 hv = h4[j];
                const float *d = dp + part * jw + 4 * j;
                acc += hv.x * d[0] + hv.y * d[1] + hv.z * d[2] + hv.w * d[3];
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (part == 0 && mem < W) dco[mem] = acc;
        }
    }
    __syncthreads();
    if (p.variant == PN_VARIANT_HETERO) {
        float tot = 0.0f;
        for (int mem = tid; mem < W; mem += 256) tot += s_coef[mem] * dco[mem];
        tot = wave_sum(tot);
        if (lane == 0) s_tot[wave] = tot;
        __syncthreads();
        tot = (s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3]);
        for (int mem = tid; mem < W; mem += 256) {
            const int64_t s = (int64_t)g * W + mem;
            dsc[mem] = s_coef[mem] * (dco[mem] - tot) * (p.rawsc[s] > 0.0f ? 1.0f : 0.01f);
        }
    } else {
        for (int mem = tid; mem < W; mem += 256) dsc[mem] = p.variant == PN_VARIANT_HOMO ? dco[mem] : 0.0f;
    }
    __syncthreads();
    // per member: d h_n, the attention-weight terms and the attention-ego term; wave w takes members w, w + 4, ...
    float gaw_h[HI], gaw_e[HI], ego_acc[HI], gab = 0.0f;
#pragma unroll
    for (int i = 0; i < HI; i++) gaw_h[i] = gaw_e[i] = ego_acc[i] = 0.0f;
    const bool has_att = p.variant != PN_VARIANT_PAGG;
    // the members' ego rows: usually one row for the whole group (HOMO: all W paths of a node start at the node) -- then the
    // waves' sums meet in LDS and the group flushes once; otherwise (HETERO, or paths that start elsewhere) a row per
    // member, flushed as it comes
    int same = 1;
    for (int mem = tid; mem < W; mem += 256) same &= s_erow[mem] == s_erow[0];
    const bool one_row = __syncthreads_and(same) != 0;
#pragma unroll 2
    for (int mem = wave; mem < W; mem += 4) {
        const int64_t s = (int64_t)g * W + mem;
        const float ds = dsc[mem], cf = s_coef[mem];
        const int64_t erow = (int64_t)s_erow[mem] * H;
#pragma unroll
        for (int i = 0; i < HI; i++) {
            const int j = lane + 64 * i;
            if (j < H) {
                float dh = cf * dp[j];
                if (has_att) {
                    dh += ds * p.att_w[j];
                    gaw_h[i] += ds * p.hn[s * H + j];
                    gaw_e[i] += ds * p.ego_tab[erow + j];
                    const float eg = ds * p.att_w[H + j];
                    if (one_row)
                        ego_acc[i] += eg;
                    else if (!p.det_ds)
                        atomicAdd(&p.dego[erow + j], eg);
                }
                p.dhn[s * H + j] = dh;
            }
        }
        gab += ds;
        if (p.det_ds && lane == 0) p.det_ds[s] = ds;
    }
    if (!has_att) return;       // block-uniform
    if (one_row && !p.det_ds) {
#pragma unroll
        for (int i = 0; i < HI; i++) {
            const int j = lane + 64 * i;
            if (j < H) red[wave * H + j] = ego_acc[i];
        }
        __syncthreads();
        const int64_t erow = (int64_t)s_erow[0] * H;
        for (int j = tid; j < H; j += 256)
            atomicAdd(&p.dego[erow + j], (red[j] + red[H + j]) + (red[2 * H + j] + red[3 * H + j]));
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < HI; i++) {
        const int j = lane + 64 * i;
        if (j < H) {
            red[wave * 2 * H + j] = gaw_h[i];
            red[wave * 2 * H + H + j] = gaw_e[i];
        }
    }
    __syncthreads();
    float *out = p.det_att + (int64_t)blockIdx.x * (2 * H + 4);
    for (int j = tid; j < 2 * H; j += 256) out[j] = red[j] + red[2 * H + j] + red[4 * H + j] + red[6 * H + j];
    if (lane == 0) out[2 * H + wave] = gab;
}
template <int HI>
__global__ __launch_bounds__(256) void pool_bwd_wg_kernel(PoolBwdParams p) {
    extern __shared__ float lds[];
    pool_bwd_wg_body<HI>(p, lds);
}

// ---- pooling forward, loss and pooling backward of a group in ONE launch (pn_pagg_train_step, default mode) -----------------
// Between the recurrence and the BPTT a training step has, per masked node and independent of every other node: attention +
// pooling + classifier (pool_fwd_kernel), softmax cross entropy of its C logits (cross_entropy_kernel) and the pooling /
// attention backward (pool_bwd_wg_kernel) -- three dependent launches of ~20 us each for work that touches 40 KB per node.
// One workgroup per node runs the three bodies back to back: the node's h_n and ego rows are re-read from the CU's own
// caches, the logits and their gradient never leave the workgroup's sight.  The same code as the three kernels (the
// bodies ARE those kernels'), so logits, g_out, d h_n and every gradient term are bit-identical to the three launches;
// the loss is the fixed-order sum of the per-node terms (loss_sum_kernel, off the critical path).
// Replaces /root/reference/PathNet_run.py:196-210 / :266-277 (forward), :346 (loss) and autograd's backward of both.
struct PoolStepParams {
    PoolParams f;
    PoolBwdParams b;
    const int64_t *target;  // [S] class of each group
    float scale;            // d loss / d (row loss): grad_scale / rows of the whole batch
    float *gout;            // [S, C] d loss / d logits (read by the classifier's weight-gradient GEMM)
    float *lossg;           // [S] logsumexp - logit[target] per group
};
template <int HI>
__global__ __launch_bounds__(256) void pool_step_kernel(PoolStepParams p) {
    extern __shared__ float lds[];
    pool_fwd_body(p.f, lds);
    __syncthreads();            // (workgroup-scope fence: out[g, :] written above is visible below; LDS is free again)
    if (threadIdx.x == 0) {     // the row's cross entropy exactly as cross_entropy_kernel computes it
        const int g = blockIdx.x, classes = p.f.C;
        const float *x = p.f.out + (int64_t)g * classes;
        float m = x[0];
        for (int c = 1; c < classes; c++) m = fmaxf(m, x[c]);
        float sum = 0.0f;
        for (int c = 0; c < classes; c++) sum += expf(x[c] - m);
        const float lse = m + logf(sum);
        const int t = (int)p.target[g];
        p.lossg[g] = lse - x[t];
        float *go = p.gout + (int64_t)g * classes;
        for (int c = 0; c < classes; c++) go[c] = (expf(x[c] - lse) - (c == t ? 1.0f : 0.0f)) * p.scale;
    }
    __syncthreads();
    pool_bwd_wg_body<HI>(p.b, lds);
}
// ---- the same step with the node's rows held in REGISTERS (round 6).  pool_step_kernel's bodies change the thread layout from
// phase to phase (eight lanes per member for the dot products, a lane per column for the sums) and therefore read the node's
// W x H final hidden states four times from memory, behind one another: ~15 dependent round trips per node, 47 us for 1 299
// nodes and 0.27 / 1.3 ms at Pubmed / BGP size, where the launch is bound by that cache traffic.  Here wave w owns members
// w, w + 4, ... (MM of them) and a lane columns lane + 64 i (HI of them) throughout: the rows are loaded ONCE, all at once
// (MM x HI registers), dot products over H are wave sums, sums over W are lane-local plus one exchange through LDS -- five
// round trips and eight barriers per node.  Same arithmetic, other summation orders than the three kernels (which the
// deterministic mode keeps: pn_pagg_train_step's bitwise tests run there); W <= 4 MM, H <= 64 HI.
#ifdef PN_POOL_TRACE
__device__ long long g_pool_trace[2048 * 12];
#define PTRACE(n) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_pool_trace[blockIdx.x * 12 + (n)] = wall_clock64(); } while (0)
#else
#define PTRACE(n) do { } while (0)
#endif
template <int MM, int HI>
__global__ __launch_bounds__(256, 6) void pool_step2_kernel(PoolStepParams p) {      // (six workgroups per CU: 1 536 slots)
    extern __shared__ float lds[];
    const PoolParams &f = p.f;
    const PoolBwdParams &b = p.b;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = blockIdx.x, H = f.H, W = f.W, C = f.C;
    const bool has_att = f.variant != PN_VARIANT_PAGG;
    // per-member scalars (wave-uniform) live in LDS, [wave][MM] each: as register arrays they cost 40 VGPRs and two of the six
    // workgroups a CU can hold
    float *cf = lds + wave * MM;        // [4][MM] pooling coefficients
    float *dsl = lds + 4 * MM + wave * MM;      // [4][MM] d score
    float *sc = lds + 8 * MM;           // [4 MM] activated scores in member order (HETERO softmax)
    float *part4 = sc + 4 * MM;         // [4][H]  waves' partial pooled sums; later [4][2H] attention-weight partials
    float *l1s = part4 + 8 * H;         // [2H] classifier input
    float *dp = l1s + 2 * H;            // [H]  d pooled / W
    float *lg = dp + H;                 // [C] logits, then [C] their gradient
    float *s_tot = lg + 2 * C;          // [4]
    const float inv_w = 1.0f / (float)W;
    const int64_t s0 = (int64_t)g * W;
    PTRACE(0);

    // ---- everything the node needs from memory, requested at once
    float hn[MM][HI], e_r[HI], aw_h[HI], aw_e[HI];
    const int e0 = has_att ? f.egoidx[s0] : 0;
    // (the node's own row of Xh, its target and the attention weights are asked for here as well, with the rows: loaded where they
    //  are used they were two more dependent round trips behind barriers)
    const int sel_g = min(max(f.sel[g], 0), f.N - 1);
    const int tgt = (int)p.target[g];
    int same = 1;
    const float *hn_g = f.hn + s0 * H;          // (a scalar base + one 32-bit offset per load: twenty 64-bit addresses would not fit)
    // Every load below is unconditional (clamped addresses) and its value always used -- rows past W / columns past H are zeroed by a
    // MULTIPLICATION, the members' start rows compared whatever the variant (PAGG: a valid dummy).  As selects, hipcc moved each load
    // into a branch of its own with a wait behind it: thirty dependent round trips, 18 of the kernel's 40 us (r06_glue.txt section 23).
    const int32_t *ego_g = has_att ? f.egoidx + s0 : reinterpret_cast<const int32_t *>(hn_g);
    int egv[MM];
#pragma unroll
    for (int k = 0; k < MM; k++) {
        const uint32_t memc = (uint32_t)min(wave + 4 * k, W - 1);
        egv[k] = ego_g[memc];
#pragma unroll
        for (int i = 0; i < HI; i++) hn[k][i] = hn_g[memc * (uint32_t)H + (uint32_t)min(lane + 64 * i, H - 1)];
    }
#pragma unroll
    for (int k = 0; k < MM; k++) {
        same &= (int)(egv[k] == e0) | (int)!has_att;
#pragma unroll
        for (int i = 0; i < HI; i++) hn[k][i] *= (wave + 4 * k < W && lane + 64 * i < H) ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < HI; i++) {
        const int j = lane + 64 * i, jc = min(j, H - 1);
        aw_h[i] = (has_att && j < H) ? f.att_w[jc] : 0.0f;
        aw_e[i] = (has_att && j < H) ? f.att_w[H + jc] : 0.0f;
        e_r[i] = (has_att && j < H) ? f.ego_tab[(int64_t)e0 * H + jc] : 0.0f;
    }
    const float xh_own = f.Xh[(int64_t)sel_g * H + min(tid, H - 1)];
    const bool one_row = __syncthreads_and(same) != 0;
    PTRACE(1);
    // ---- scores and pooling coefficients of this wave's members (raw scores -> cf[], LeakyReLU slopes -> bits of `neg`)
    uint32_t neg = 0;       // bit k: member k's raw score <= 0
    if (has_att) {
        const float ab = f.att_b[0];
        float edot = 0.0f;
#pragma unroll
        for (int i = 0; i < HI; i++) edot += e_r[i] * aw_e[i];
        edot = wave_sum_u(edot);
#pragma unroll
        for (int k = 0; k < MM; k++) {
            const int mem = wave + 4 * k;
            float part = 0.0f;
#pragma unroll
            for (int i = 0; i < HI; i++) part += hn[k][i] * aw_h[i];
            float ed = edot;
            if (!one_row) {         // (rare: a member whose path starts elsewhere)
                const int64_t er = (int64_t)f.egoidx[s0 + min(mem, W - 1)] * H;
                float pe = 0.0f;
#pragma unroll
                for (int i = 0; i < HI; i++) {
                    const int j = lane + 64 * i;
                    if (j < H) pe += f.ego_tab[er + j] * aw_e[i];
                }
                ed = wave_sum_u(pe);
            }
            const float raw = wave_sum_u(part) + ed + ab;
            if (!(raw > 0.0f)) neg |= 1u << k;
            if (lane == 0) {
                cf[k] = raw;
                if (mem < W) f.rawsc[s0 + mem] = raw;
            }
        }
    } else if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MM; k++) {
            cf[k] = 0.0f;
            if (wave + 4 * k < W) f.rawsc[s0 + wave + 4 * k] = 0.0f;
        }
    }
    if (f.variant == PN_VARIANT_HETERO) {       // softmax over the W members of LeakyReLU(score)
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < MM; k++) {
                const int mem = wave + 4 * k;
                const float r = cf[k];
                if (mem < W) sc[mem] = r > 0.0f ? r : 0.01f * r;
            }
        }
        __syncthreads();
        const float v = lane < W ? sc[lane] : -3.4e38f;       // (W <= 4 MM <= 64)
        const float mx = wave_max(v);
        const float sum = wave_sum_u(lane < W ? expf(v - mx) : 0.0f);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < MM; k++) {
                const float r = cf[k];
                cf[k] = expf((r > 0.0f ? r : 0.01f * r) - mx) / sum;
            }
        }
    } else if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MM; k++) cf[k] = f.variant == PN_VARIANT_HOMO ? 1.0f + cf[k] : 1.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (cf[] of this wave: written by lane 0, read by the wave)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < MM; k++)
            if (wave + 4 * k < W) f.coef[s0 + wave + 4 * k] = cf[k];
    }
    // ---- pooled = mean_w coef_w h_w
    PTRACE(2);
#pragma unroll
    for (int i = 0; i < HI; i++) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < MM; k++) acc += cf[k] * hn[k][i];       // (members past W hold zeros)
        const int j = lane + 64 * i;
        if (j < H) part4[wave * H + j] = acc;
    }
    __syncthreads();
    PTRACE(3);
    // ---- layer1 = dropout([Xh[sel[g]] ; pooled]); the masks stay in registers for the backward
    float m_a = 1.0f, m_b = 1.0f;       // this thread's column tid (< H)
    if (tid < H) {
        const int j = tid;
        const uint64_t gg = (uint64_t)(f.goff + g);
        float a = xh_own, bq = (part4[j] + part4[H + j] + part4[2 * H + j] + part4[3 * H + j]) * inv_w;
        if (f.mask) {
            m_a = f.mask[gg * 2 * H + j];
            m_b = f.mask[gg * 2 * H + H + j];
        } else if (f.p_drop > 0.0f) {
            const uint64_t seed = f.dyn ? f.dyn->seed : f.seed;
            m_a = dropout1(seed, gg * 2 * H + j, 2u, f.p_drop);
            m_b = dropout1(seed, gg * 2 * H + H + j, 2u, f.p_drop);
        }
        a *= m_a;
        bq *= m_b;
        float *l1 = f.layer1 + (int64_t)g * 2 * H;
        l1[j] = a;
        l1[H + j] = bq;
        l1s[j] = a;
        l1s[H + j] = bq;
    }
    __syncthreads();
    PTRACE(4);
    for (int c = wave; c < C; c += 4) {
        // (a fixed trip count with clamped addresses: the row's loads leave together instead of one round trip per 64 columns)
        const float *wrow = f.fc2_w + (int64_t)c * 2 * H;
        float wv[2 * HI];
#pragma unroll
        for (int q = 0; q < 2 * HI; q++) wv[q] = wrow[min(lane + 64 * q, 2 * H - 1)];
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 2 * HI; q++) {
            const int j = lane + 64 * q;
            if (j < 2 * H) part += l1s[j] * wv[q];
        }
        part = wave_sum_u(part);
        if (lane == 0) {
            const float v = part + f.fc2_b[c];
            f.out[(int64_t)g * C + c] = v;
            lg[c] = v;
        }
    }
    __syncthreads();
    PTRACE(5);
    if (wave == 0) {        // the row's cross entropy as cross_entropy_kernel computes it, a lane per class (the sum in another order)
        float m = -3.4e38f;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, lg[c]);
        m = wave_max(m);
        float sum = 0.0f;
        for (int c = lane; c < C; c += 64) sum += expf(lg[c] - m);
        const float lse = m + logf(wave_sum_u(sum));
        const int t = tgt;
        if (lane == 0) p.lossg[g] = lse - lg[t];
        float *go = p.gout + (int64_t)g * C;
        for (int c = lane; c < C; c += 64) {
            const float gv = (expf(lg[c] - lse) - (c == t ? 1.0f : 0.0f)) * p.scale;
            go[c] = gv;
            lg[C + c] = gv;
        }
    }
    __syncthreads();
    PTRACE(6);
    // ---- backward: d layer1 = g_out . fc2_w (masked): its ego half onto the node's row of d Xh, its pooled half / W = dp
    if (tid < H) {
        const int j = tid;
        float a = 0.0f, bq = 0.0f;
        for (int c0 = 0; c0 < C; c0 += 4) {     // four classes' loads in flight (clamped rows, masked sums)
            float wa[4], wb[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float *wrow = f.fc2_w + (int64_t)min(c0 + q, C - 1) * 2 * H;
                wa[q] = wrow[j];
                wb[q] = wrow[H + j];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float go = c0 + q < C ? lg[C + c0 + q] : 0.0f;
                a += go * wa[q];
                bq += go * wb[q];
            }
        }
        atomicAdd(&b.dXh[(int64_t)sel_g * H + j], a * m_a);
        dp[j] = bq * m_b * inv_w;
    }
    __syncthreads();
    PTRACE(7);
    float dpr[HI];
#pragma unroll
    for (int i = 0; i < HI; i++) {
        const int j = lane + 64 * i;
        dpr[i] = j < H ? dp[j] : 0.0f;
    }
    {       // d coef of this wave's members -> dsl[]; HETERO: through the softmax
        float tpart = 0.0f;
#pragma unroll
        for (int k = 0; k < MM; k++) {
            float part = 0.0f;
#pragma unroll
            for (int i = 0; i < HI; i++) part += hn[k][i] * dpr[i];
            const float dco = wave_sum_u(part);
            if (wave + 4 * k < W) tpart += cf[k] * dco;
            if (lane == 0) dsl[k] = dco;
        }
        if (f.variant == PN_VARIANT_HETERO) {
            if (lane == 0) s_tot[wave] = tpart;
            __syncthreads();
            const float tot = (s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3]);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < MM; k++) dsl[k] = cf[k] * (dsl[k] - tot) * ((neg >> k) & 1u ? 0.01f : 1.0f);
            }
        } else if (f.variant != PN_VARIANT_HOMO && lane == 0) {
#pragma unroll
            for (int k = 0; k < MM; k++) dsl[k] = 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    PTRACE(8);
    // ---- per member: d h_n; the attention-weight and attention-ego terms accumulate in registers
    float gaw_h[HI], gaw_e[HI], ego_acc[HI], gab = 0.0f;
#pragma unroll
    for (int i = 0; i < HI; i++) gaw_h[i] = gaw_e[i] = ego_acc[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < MM; k++) {
        const int mem = wave + 4 * k;
        if (mem >= W) break;            // (wave-uniform)
        const float ds = dsl[k], ck = cf[k];
        const int64_t er = (has_att && !one_row) ? (int64_t)f.egoidx[s0 + mem] * H : 0;
#pragma unroll
        for (int i = 0; i < HI; i++) {
            const int j = lane + 64 * i;
            if (j < H) {
                float dh = ck * dpr[i];
                if (has_att) {
                    dh += ds * aw_h[i];
                    gaw_h[i] += ds * hn[k][i];
                    const float eg = ds * aw_e[i];
                    if (one_row) {
                        gaw_e[i] += ds * e_r[i];
                        ego_acc[i] += eg;
                    } else {
                        gaw_e[i] += ds * f.ego_tab[er + j];
                        atomicAdd(&b.dego[er + j], eg);
                    }
                }
                b.dhn[(s0 + mem) * H + j] = dh;
            }
        }
        gab += ds;
    }
    PTRACE(9);
    if (!has_att) return;       // block-uniform
    float *red = part4;         // [4][2H]
    if (one_row) {
#pragma unroll
        for (int i = 0; i < HI; i++) {
            const int j = lane + 64 * i;
            if (j < H) red[wave * H + j] = ego_acc[i];
        }
        __syncthreads();
        if (tid < H) atomicAdd(&b.dego[(int64_t)e0 * H + tid], (red[tid] + red[H + tid]) + (red[2 * H + tid] + red[3 * H + tid]));
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < HI; i++) {
        const int j = lane + 64 * i;
        if (j < H) {
            red[wave * 2 * H + j] = gaw_h[i];
            red[wave * 2 * H + H + j] = gaw_e[i];
        }
    }
    __syncthreads();
    float *out = b.det_att + (int64_t)g * (2 * H + 4);
    for (int j = tid; j < 2 * H; j += 256) out[j] = red[j] + red[2 * H + j] + red[4 * H + j] + red[6 * H + j];
    if (lane == 0) out[2 * H + wave] = gab;
    PTRACE(10);
}
inline size_t pool_step2_lds_bytes(int MM, int H, int C) { return (size_t)(12 * MM + 8 * H + 2 * H + H + 2 * C + 8) * sizeof(float); }

// loss[0] (+)= scale * sum of the rows' terms, in cross_entropy_kernel's order (one workgroup of 1024 threads)
__global__ __launch_bounds__(1024) void loss_sum_kernel(const float *__restrict__ lossg, int rows, float scale,
                                                        float *__restrict__ loss, int store) {
    __shared__ float part[16];
    float mine = 0.0f;
    for (int r = threadIdx.x; r < rows; r += 1024) mine += lossg[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 16; w++) t += part[w];
        if (store)
            *loss = t * scale;
        else
            atomicAdd(loss, t * scale);
    }
}

// ================================================================================================
// Deterministic scatter-add (pn_pagg_shape.deterministic): dst[key[i]] += contribution i, added in the order of i.
// The default backward scatters with fp32 atomics, whose order -- and with it the last bits of every gradient -- changes
// from run to run (the reference inherits the same from torch's index_add / embedding backward).
// Here the (key, i) pairs are sorted by key (stable radix sort, pn_sort.hip), the sorted array is cut into chunks of
// DET_CHUNK positions, and one group of H/4 lanes walks a chunk in order:
//   pass 1  a run of equal keys that lies inside the chunk is added to dst by its one owner; a run that crosses a chunk
//           boundary leaves its piece in cpart[chunk][lead | trail];
//   pass 2  the chunk holding the head of a crossing run adds the pieces up in chunk order and adds the sum to dst.
// Every dst row has exactly one writer per pass and a fixed order of additions: bitwise reproducible.
// A contribution is a row of `rows` ([K, H]) or scal[i] * vec[0..H) (the attention-ego term of the pooling backward).
// The BPTT kernels need no second version for this: in deterministic mode they are handed the identity as their row
// index and the zero-filled contribution buffer as their table -- every atomic then lands on an address of its own,
// i.e. is a store -- and the scatter proper happens here.
// ================================================================================================
constexpr int DET_CHUNK = 32;
struct DetScatterParams {
    const int32_t *skey, *ssrc;     // sorted keys (with the scatter's segment number above `kmask`); the unsorted position of every one
    int32_t kmask;
    int64_t K, chunk0;              // positions; first chunk of this scatter's region of cpart
    const float *rows;              // [K, H] or null
    const float *scal, *vec;        // [K], [H]
    float *dst;                     // [., H]
    float *cpart;                   // [chunks][2][H]
    int H;
};

struct DetScatterPair {          // blockIdx.y picks the scatter: the pooling backward's two run in one launch
    DetScatterParams s[2];
};

template <int PASS>
__global__ __launch_bounds__(256) void det_scatter_kernel(DetScatterPair pp) {
    const DetScatterParams &p = pp.s[blockIdx.y];
    const int hv = p.H / 4, gpb = 256 / hv;
    const int grp = threadIdx.x / hv, ln = threadIdx.x - grp * hv;
    if (grp >= gpb) return;
    const int64_t c = (int64_t)blockIdx.x * gpb + grp, b = c * DET_CHUNK;
    if (b >= p.K) return;
    const int64_t e = min(p.K, b + (int64_t)DET_CHUNK);
    float4 *dst = reinterpret_cast<float4 *>(p.dst);
    float4 *cpart = reinterpret_cast<float4 *>(p.cpart) + p.chunk0 * 2 * hv;
    const int32_t km = p.kmask;
    auto add_to = [&](float4 *d, const float4 &a) {
        float4 v = *d;
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        *d = v;
    };
    if (PASS == 1) {
        const float4 vec = p.rows ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4 *>(p.vec)[ln];
        int key = p.skey[b] & km;
        bool lead = b > 0 && (p.skey[b - 1] & km) == key;      // the chunk opens inside a run
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // four positions per trip: their contributions are requested together (the walk was one dependent 512-byte row load
        // per position -- latency, 85 us for the 208 k rows of the headline shape) and added in the order of the positions
        constexpr int DU = 4;
        for (int64_t i0 = b; i0 < e; i0 += DU) {
            int kk[DU];
            float4 vv[DU];
#pragma unroll
            for (int j = 0; j < DU; j++) {
                const int64_t i = min(i0 + j, e - 1);   // (past the end: a harmless re-load of the last position)
                kk[j] = p.skey[i] & km;
                const int64_t src = p.ssrc[i];
                if (p.rows) {
                    vv[j] = reinterpret_cast<const float4 *>(p.rows)[src * hv + ln];
                } else {
                    const float sc = p.scal[src];
                    vv[j] = make_float4(sc * vec.x, sc * vec.y, sc * vec.z, sc * vec.w);
                }
            }
#pragma unroll
            for (int j = 0; j < DU; j++) {
                if (i0 + j >= e) break;
                const int k = kk[j];
                if (k != key) {                          // the run of `key` ends inside the chunk
                    if (lead)
                        cpart[(c * 2 + 0) * hv + ln] = acc;
                    else
                        add_to(dst + (int64_t)key * hv + ln, acc);
                    lead = false;
                    key = k;
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                acc.x += vv[j].x; acc.y += vv[j].y; acc.z += vv[j].z; acc.w += vv[j].w;
            }
        }
        const bool cont = e < p.K && (p.skey[e] & km) == key;   // the last run goes on in the next chunk
        if (lead)
            cpart[(c * 2 + 0) * hv + ln] = acc;
        else if (cont)
            cpart[(c * 2 + 1) * hv + ln] = acc;
        else
            add_to(dst + (int64_t)key * hv + ln, acc);
    } else {
        const int key = p.skey[e - 1] & km;
        if (!(e < p.K && (p.skey[e] & km) == key)) return;                      // nothing crosses this chunk's end
        if ((p.skey[b] & km) == key && b > 0 && (p.skey[b - 1] & km) == key) return;   // a middle chunk of the run: its head's chunk adds
        float4 acc = cpart[(c * 2 + 1) * hv + ln];
        for (int64_t c2 = c + 1;; c2++) {
            const int64_t e2 = min(p.K, (c2 + 1) * (int64_t)DET_CHUNK);
            const float4 v = cpart[(c2 * 2 + 0) * hv + ln];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            if ((p.skey[e2 - 1] & km) != key || !(e2 < p.K && (p.skey[e2] & km) == key)) break;
        }
        add_to(dst + (int64_t)key * hv + ln, acc);
    }
}

// keys of a scatter, clamped to [0, hi], beside the identity permutation the sort carries along
struct DetKeys3 {
    const int32_t *keys[3];
    int64_t n[3];
    int32_t hi[3];
    int shift;
};
__global__ __launch_bounds__(256) void det_keys3_kernel(DetKeys3 q, int32_t *__restrict__ keys, int32_t *__restrict__ pos) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t at = i;
    int seg = 0;
    while (seg < 3 && i >= q.n[seg]) i -= q.n[seg], seg++;
    if (seg >= 3) return;
    keys[at] = min(max(q.keys[seg][i], 0), q.hi[seg]) | (seg << q.shift);
    pos[at] = (int32_t)i;
}

__global__ __launch_bounds__(256) void det_iota_kernel(int32_t *__restrict__ iota, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) iota[i] = (int32_t)i;
}

// the pooling backward's per-workgroup attention terms ([nblk][2H + 4]: g_att_w partials, then one g_att_b term per
// wave) added up in a fixed order.  A workgroup owns 64 consecutive outputs: sixteen row groups of 64 lanes walk the
// partials' rows 16 apart -- every load a 256-byte run of one row -- and meet in LDS (round 6; a workgroup per output read
// the [nblk][260] array column-wise, one cache line per element: 19 us for 1.3 MB at the headline shape)
__global__ __launch_bounds__(1024) void det_att_reduce_kernel(const float *__restrict__ part, int nblk, int H,
                                                              float *__restrict__ g_att_w, float *__restrict__ g_att_b) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6, stride = 2 * H + 4;
    const int col = blockIdx.x * 64 + lane;
    float v = 0.0f;
    if (col < stride) {
        const float *src = part + col;
#pragma unroll 8
        for (int b = rg; b < nblk; b += 16) v += src[(int64_t)b * stride];
    }
    red[rg][lane] = v;
    __syncthreads();
    if (rg != 0) return;
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) t += red[r][lane];
    if (col < 2 * H) {
        g_att_w[col] += t;
    } else if (col < stride) {      // the four per-wave bias terms sit in one workgroup (4 | 2H): lanes c .. c + 3
        red[0][lane] = t;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (col == 2 * H) *g_att_b += (red[0][lane] + red[0][lane + 1]) + (red[0][lane + 2] + red[0][lane + 3]);
    }
}

// out [R][F4] = in [R][F] with zeros in columns F .. F4 - 1 (F4 = F rounded up to 4): 16-byte stores, dword loads
__global__ __launch_bounds__(256) void pad_rows_kernel(const float *__restrict__ in, int64_t R, int F, int F4, float *__restrict__ out) {
    const int q4 = F4 / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * q4) return;
    const int64_t r = i / q4;
    const int c = (int)(i - r * q4) * 4;
    const float *src = in + r * F + c;
    float4 v;
    v.x = src[0];                       // (c < F always)
    v.y = c + 1 < F ? src[1] : 0.0f;
    v.z = c + 2 < F ? src[2] : 0.0f;
    v.w = c + 3 < F ? src[3] : 0.0f;
    reinterpret_cast<float4 *>(out)[i] = v;
}

// zero-fill of up to 12 buffers in one launch (the accumulated gradients of a backward)
struct ZeroList {
    float *ptr[12];
    unsigned long long count[12];
    int n;
};
__global__ __launch_bounds__(256) void zero_kernel(ZeroList z) {
    for (int b = 0; b < z.n; b++) {
        float *p = z.ptr[b];
        const unsigned long long cnt = z.count[b];
        // 16-byte stores on the aligned body, scalars on the ragged ends
        const unsigned long long head = min(cnt, (unsigned long long)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) / 4);
        const unsigned long long body4 = (cnt - head) / 4;
        float4 *p4 = reinterpret_cast<float4 *>(p + head);
        for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < body4;
             i += (unsigned long long)gridDim.x * blockDim.x)
            p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (blockIdx.x == 0) {
            for (unsigned long long i = threadIdx.x; i < head; i += blockDim.x) p[i] = 0.0f;
            for (unsigned long long i = head + body4 * 4 + threadIdx.x; i < cnt; i += blockDim.x) p[i] = 0.0f;
        }
    }
}

// ================================================================================================
// shape bookkeeping and workspace layout
// ================================================================================================
enum { CELL_LSTM = 1, CELL_RNN = 2, CELL_GRU = 3, CELL_MEAN = 4, CELL_SUM = 5 };   // = PN_CELL_*
struct Dims {
    int variant, N, F, H, C, S, W, L, G, SV;    // G: gate slots of the recurrent kernels (4: LSTM and GRU, 1: RNN, 0: mean / sum)
    int cell, Gw;                               // cell kind; Gw: gates of the caller's weight tensors (4, 1, 3, 0)
    bool generic;                               // H > 256: the step-by-step recurrence (gen_*_kernel) instead of the fused kernels
    int S_total, group_begin;
    int Sb;             // pooling groups per micro-batch
    int nb;             // micro-batches of this call
    int64_t P_total;    // paths of the whole batch
    bool det;           // pn_pagg_shape.deterministic: the backward adds in a fixed order (no floating-point atomics)
    bool compact;       // the bank runs over the (node, code) rows this call's paths touch (compact_rows)
    int64_t ZR;         // rows of Z / dZ: N * L, or the bound of the touched rows
    int math;           // PN_SEQ_MATH_F16X2 / PN_SEQ_MATH_BF16X3 for the fused recurrent kernels, 0 when the call has none
};

int make_dims(const pn_pagg_shape &s, Dims &d) {
    if (s.variant < 0 || s.variant > 2) PN_FAIL(PN_ERR_ARG, "unknown variant %d", s.variant);
    if (s.H < 32 || s.H > 1024 || s.H % 32 != 0)
        PN_FAIL(PN_ERR_ARG, "hidden size %d not supported (multiples of 32 up to 1024; fused kernels up to 256)", s.H);
    if (s.N < 1 || s.F < 1 || s.C < 1 || s.S < 0 || s.W < 1 || s.L < 1 || s.L > 64)
        PN_FAIL(PN_ERR_ARG, "bad aggregator shape N=%d F=%d C=%d S=%d W=%d L=%d", s.N, s.F, s.C, s.S, s.W, s.L);
    // the pooling kernels keep per-walk scores, coefficients, ego rows and a few rows of H floats in LDS
    if (64 * (int64_t)s.W + 48 * s.H > 64 * 1024)
        PN_FAIL(PN_ERR_ARG, "W=%d walks per node exceed the pooling kernels' LDS budget (W <= %d at H=%d)", s.W,
                (64 * 1024 - 48 * s.H) / 64, s.H);
    if (s.variant == PN_VARIANT_PAGG && s.L != 4)
        PN_FAIL(PN_ERR_ARG, "PAGG has exactly four distance layers nei0..nei3 (copy.py:310-313); L=%d", s.L);
    if (s.cell < 0 || s.cell > CELL_SUM) PN_FAIL(PN_ERR_ARG, "unknown cell %d", s.cell);
    const int S_total = s.S_total > 0 ? s.S_total : s.S;
    if (s.group_begin < 0 || (int64_t)s.group_begin + s.S > S_total || s.batch_groups < 0)
        PN_FAIL(PN_ERR_ARG, "bad slice: groups [%d, +%d) of a batch of %d, batch_groups=%d", s.group_begin, s.S,
                S_total, s.batch_groups);
    // int32 indices: a node's table row (node * L + code) and a path's position in the batch
    if ((int64_t)S_total * s.W > 2000000000LL || (int64_t)s.N * s.L > 2000000000LL)
        PN_FAIL(PN_ERR_ARG, "index space exceeds int32 (S_total*W = %lld paths, N*L = %lld rows)",
                (long long)S_total * s.W, (long long)s.N * s.L);
    d.variant = s.variant;
    d.N = s.N, d.F = s.F, d.H = s.H, d.C = s.C, d.S = s.S, d.W = s.W, d.L = s.L;
    d.cell = s.cell ? s.cell : (s.variant == PN_VARIANT_PAGG ? CELL_RNN : CELL_LSTM);
    d.G = d.cell == CELL_RNN ? 1 : (d.cell == CELL_LSTM || d.cell == CELL_GRU) ? 4 : 0;
    d.Gw = d.cell == CELL_GRU ? 3 : d.G;
    d.SV = d.G == 4 ? 5 : 1;
    d.generic = s.H > 256 && d.G > 0;
    d.S_total = S_total;
    d.group_begin = s.group_begin;
    d.Sb = (s.batch_groups > 0 && s.batch_groups < s.S) ? s.batch_groups : s.S;
    d.nb = d.Sb > 0 ? (s.S + d.Sb - 1) / d.Sb : 1;
    d.P_total = (int64_t)S_total * s.W;
    d.det = s.deterministic != 0;
    {
        // compaction pays when the path steps of this call cannot touch half of the N * L rows (pn_pagg_shape.compact forces
        // it on / off: tests, A/B); the hetero class reads paths of the whole batch, a slice marks the whole batch's steps
        const int64_t steps = (int64_t)(d.variant == PN_VARIANT_HETERO ? S_total : s.S) * s.W * s.L, rows = (int64_t)s.N * s.L;
        if (s.compact < 0 || s.compact > PN_COMPACT_OFF) PN_FAIL(PN_ERR_ARG, "unknown compact mode %d", s.compact);
        d.compact = s.S > 0 && (s.compact == PN_COMPACT_AUTO ? 2 * steps < rows : s.compact == PN_COMPACT_ON);
        d.ZR = d.compact ? std::min(rows, steps) : rows;
    }
    if (s.seq_math < 0 || s.seq_math > PN_SEQ_MATH_F16X2) PN_FAIL(PN_ERR_ARG, "unknown seq_math %d", s.seq_math);
    d.math = (d.G > 0 && !d.generic) ? (s.seq_math == PN_SEQ_MATH_BF16X3 ? PN_SEQ_MATH_BF16X3 : PN_SEQ_MATH_F16X2) : 0;
    // rows of one micro-batch's [Pb * L, .] tensors are counted in int32 inside the kernels
    if ((int64_t)d.Sb * s.W * s.L > 2000000000LL)
        PN_FAIL(PN_ERR_ARG, "%lld path steps in one micro-batch exceed int32: set batch_groups", (long long)d.Sb * s.W * s.L);
    return PN_OK;
}

#ifndef PN_WGRAD_CUS_SHARED
#define PN_WGRAD_CUS_SHARED 224    // (tuning builds 176 .. 240: profiles/r06_glue.txt section 11 -- 208 won while fc0's weight
#endif                             //  gradient was a 150 us gemm_kernel launch beside it, 224 again once that ran on rgrad_kernel)
constexpr int WGRAD_CUS_SHARED = PN_WGRAD_CUS_SHARED;   // CUs of the weight-gradient launch when something is meant to run beside it (of 256)
// fc0's weight gradient runs on rgrad_kernel (16-byte row loads) also when F is not a multiple of 4 -- Cora's 1433 -- from a
// copy of X with padded rows, made per backward (pad_rows_kernel: 2 x 15 MB at the headline shape, ~5 us, for a GEMM that
// takes 150 us on the 48 CUs the recurrent weight gradient leaves it and ~60 as a row reduction).  Graphs from 2 048 nodes
// on whose copy stays below 256 MB; floats of the copy, 0: no copy.
inline size_t xpad_floats(const Dims &d) {
    if (d.F % 4 == 0 || d.N < 2048 || d.det) return 0;
    const size_t f4 = (size_t)(d.F + 3) / 4 * 4, n = (size_t)d.N * f4;
    return n * 4 <= ((size_t)256 << 20) ? n : 0;
}

struct WsLayout {
    size_t Xh, Z, range, wmax;                                           // node tables (first: reuse_tables relies on it); partial weight maxima
    size_t Wp, biasc, WpT, wpart, gpart, dZ, dXh;                        // weights / node-level backward
    size_t rowidx, egoidx, slotof, hn, saved, coef, rawsc, layer1, outb; // per micro-batch, forward
    size_t xh, keep, dG, dhn, dl1, gx, gout, bankT, xpad;                // per micro-batch, saved / backward
    size_t flags, rank, list, seg, bsum;                                 // touched-row compaction (Dims.compact)
    size_t dx, keys, iota, skey, stmp, cpart, dsel, dds, datt, dgemm;  // deterministic backward (Dims.det)
    size_t okey, osrc;                  // its three destination orders, sorted once per micro-batch into one block each: segment
                                        // offsets are running sums over the segments present (det_params), nothing is laid out here
    size_t stmp_bytes;
    int wgrad_split, wgrad_tiles;
    size_t total;
};

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

WsLayout ws_layout(const Dims &d) {
    WsLayout w{};
    const size_t N = d.N, H = d.H, L = d.L, G = d.G, SV = d.SV, Sb = d.Sb, Pb = (size_t)d.Sb * d.W;
    size_t at = 0;
    auto take = [&](size_t bytes) {
        size_t o = at;
        at = align256(at + bytes);
        return o;
    };
    w.Xh = take(N * H * 4);
    w.Z = take((size_t)d.ZR * H * 4);
    w.range = take(sizeof(SeqRange));       // operand ranges of the fp16 recurrent kernels; range.x belongs to Z
    w.wmax = take(64 * 4);                  // partial maxima of |W_ih| / |W_hh| (range_part_kernel -> pack_fb_kernel)
    w.Wp = take(G * H * 3 * H * 4);          // (G = 0, the mean / sum encoders: no recurrent weights, no saved gates)
    w.biasc = take(G * H * 4);
    w.WpT = take(G * H * 3 * H * 4);
    {
        // split of the Pb*L rows of the weight-gradient GEMM: one 8-wave workgroup per CU.  Such a workgroup takes the CU's
        // whole register file: with one on every CU the node-level GEMMs of the main stream (bank / fc0 backward) cannot
        // start until the launch is over, whatever stream they are on.  A short launch (a Cora-sized batch: 0.17 ms against
        // ~0.07 ms of launch-bound GEMMs) therefore leaves an eighth of the CUs free -- the kernel is HBM-bound enough to
        // lose 3 % on 224 CUs, the step wins 2.8 % (0.988 -> 0.959 ms; re-measured in round 6); where the weight gradient is 6-12x the GEMM chain
        // (Pubmed, BGP size: +10 % on the kernel for nothing hidden) it keeps every CU.  profiles/r04_wgrad_cus_ab.txt
        const size_t rows = Pb * L, tiles = std::max<size_t>(1, ((G * H + WG_BM - 1) / WG_BM) * ((2 * H + WG_BN - 1) / WG_BN));
        const double wg_flops = 2.0 * (double)rows * (double)(G * H) * (double)(2 * H);
        size_t cus = wg_flops <= 1.2e11 ? WGRAD_CUS_SHARED : 256;
        size_t nz = (cus + tiles - 1) / tiles;
        const size_t max_nz = (rows + 4 * WG_KT - 1) / (4 * WG_KT);
        if (nz > max_nz) nz = max_nz;
        if (nz < 1) nz = 1;
        w.wgrad_split = (int)nz;
        w.wgrad_tiles = (int)tiles;
        w.wpart = take(nz * (G * H * 2 * H + G * H) * 4);
    }
    {
        const int nz = std::max(gemm_split_count(d.N, d.H, d.F), gemm_split_count(d.N, d.H, d.L * d.H));
        w.gpart = take(nz > 1 ? (size_t)nz * N * H * 4 : 0);
    }
    w.dZ = take((size_t)d.ZR * H * 4);
    w.dXh = take(N * H * 4);
    w.rowidx = take(Pb * L * 4);
    w.egoidx = take(Pb * 4);
    w.slotof = take(Pb * 4);
    w.hn = take(Pb * H * 4);
    w.saved = take(Pb * L * SV * H * 4);
    w.coef = take(Pb * 4);
    w.rawsc = take(Pb * 4);
    w.layer1 = take(Sb * 2 * H * 4);
    w.outb = take(Sb * (size_t)d.C * 4);
    w.xh = take(Pb * L * 2 * H * 4);
    w.keep = take(Pb * L * (H / 4));
    w.dG = take(Pb * L * G * H * 4);
    w.dhn = take(Pb * H * 4);
    w.dl1 = take(Sb * 2 * H * 4 + 1024);
    w.gx = take(d.generic ? Pb * 2 * H * 4 : 0);        // [dx_t | dh_{t-1}] of a step of the generic recurrence
    w.gout = take(Sb * (size_t)d.C * 4);                // d loss / d logits of a micro-batch (pn_pagg_train_step)
    w.datt = take(Sb * (2 * H + 4) * 4);                // per-workgroup attention-weight terms of the pooling backward
    w.bankT = take(L * H * H * 4);                      // transposed bank weights (the dX GEMM on the bf16 x 3 kernel)
    w.xpad = take(xpad_floats(d) * 4);                  // X with its rows padded to a multiple of four floats (fc0's weight gradient on rgrad_kernel)
    w.flags = take(d.compact ? N * L + 16 : 0);
    w.rank = take(d.compact ? N * L * 4 : 0);
    w.list = take(d.compact ? (size_t)d.ZR * 4 : 0);
    w.seg = take(d.compact ? (L + 2) * 4 : 0);
    w.bsum = take(d.compact ? ((N * L + SCAN_BLOCK - 1) / SCAN_BLOCK + 1) * 4 : 0);
    if (d.det) {
        // contributions of the gather backward ([Pb * L, H] rows), their (destination row, position) pairs before and
        // after the sort, the sort's temporary storage, the sums of the segments that cross a chunk boundary; the
        // pooling backward's per-group / per-path / per-workgroup terms; the chunk sums of the weight-gradient GEMMs
        const size_t K = std::max(Pb * L, Sb), C = d.C, F = d.F;
        w.dx = take(Pb * L * H * 4);
        // the three scatters' (destination, position) pairs are sorted TOGETHER (prepare_det_orders): keys / positions before the
        // sort (keys, skey), after it (okey / osrc: one block, a segment per scatter), the identity the BPTT indexes with (iota)
        const size_t ord[3] = {Sb, Pb, Pb * L};        // DET_SEL, DET_EGO, DET_ROW
        const size_t Kt = ord[0] + ord[1] + ord[2];
        w.keys = take(Kt * 4);
        w.iota = take(K * 4);
        w.skey = take(Kt * 4);
        w.okey = take(Kt * 4);
        w.osrc = take(Kt * 4);
        w.stmp_bytes = sort_temp_reserve((int64_t)Kt);
        w.stmp = take(w.stmp_bytes);
        w.cpart = take((Kt / DET_CHUNK + 4) * 2 * H * 4);      // (a region per scatter: SEL and EGO run in one launch)
        w.dsel = take(Sb * H * 4);
        w.dds = take(Pb * 4);
        w.dgemm = take(std::max({det_gemm_floats(C, 2 * H), det_gemm_floats(d.compact ? H : L * H, H), det_gemm_floats(H, F)}) * 4);
    }
    w.total = at;
    return w;
}

// Independent launches go to the context's second stream (pn_context.hip): the index plan and the weight packing beside
// fc0 / bank in the forward; in the backward the classifier's weight gradient beside the pooling backward and the
// BPTT, and the recurrent weight-gradient GEMM (one 8-wave workgroup per CU, 104 KB of LDS) beside the node-level
// GEMMs that follow the BPTT.  The guard makes `stream` wait for whatever was forked on every way out.
struct JoinGuard {
    pn_context *ctx;
    hipStream_t stream;
    bool pending = false;
    int mark() {                        // the forked work is complete up to here on the second stream
        if (int rc = context_record_join(ctx)) return rc;
        pending = true;
        return PN_OK;
    }
    int join() {
        if (!pending) return PN_OK;
        pending = false;
        return context_join(ctx, stream);
    }
    ~JoinGuard() { (void)join(); }
};


// ---- one aggregator call, resolved: shapes, workspace pointers, and the stages that run once per micro-batch ---------
struct Call {
    pn_context *ctx;
    hipStream_t stream;
    const pn_pagg_args *a;
    Dims d;
    WsLayout w;
    char *ws;
    const float *Xh;
    float *Z;
    template <class T>
    T *at(size_t off) const { return reinterpret_cast<T *>(ws + off); }
    int groups(int b) const { return std::min(d.Sb, d.S - b * d.Sb); }                 // pooling groups of micro-batch b
    int64_t group0(int b) const { return (int64_t)d.group_begin + (int64_t)b * d.Sb; }  // its first group, batch-wide
    const int32_t *sel(int b) const { return a->sel + (a->index_rows_local ? (int64_t)b * d.Sb : group0(b)); }
};

int resolve_call(Call &c, pn_context *ctx, const pn_pagg_args *a, hipStream_t stream, const char *who) {
    if (!a) PN_FAIL(PN_ERR_ARG, "%s: null args", who);
    if (int rc = context_check_device(ctx)) return rc;
    if (int rc = make_dims(a->shape, c.d)) return rc;
    c.ctx = ctx;
    c.stream = stream;
    c.a = a;
    c.w = ws_layout(c.d);
    if (!a->workspace) PN_FAIL(PN_ERR_ARG, "%s: null workspace", who);
    if (a->workspace_bytes < (int64_t)c.w.total)
        PN_FAIL(PN_ERR_CAPACITY, "aggregator workspace holds %lld bytes, need %lld", (long long)a->workspace_bytes,
                (long long)c.w.total);
    c.ws = reinterpret_cast<char *>(a->workspace);
    c.Xh = a->Xh_in ? a->Xh_in : c.at<const float>(c.w.Xh);
    c.Z = c.at<float>(c.w.Z);
    if (a->index_rows_local && c.d.variant == PN_VARIANT_HETERO && (c.d.group_begin != 0 || c.d.S != c.d.S_total))
        PN_FAIL(PN_ERR_ARG, "%s: the hetero class reads paths of the whole batch (PathNet_run.py:196-197): "
                "index_rows_local needs the whole batch's ids / codes", who);
    return PN_OK;
}

int run_plan(const Call &c, hipStream_t s, int b) {
    const Dims &d = c.d;
    const int64_t count = (int64_t)c.groups(b) * d.W, n = count * d.L;
    const PlanDims pd{d.S_total, d.W, d.L, d.N, d.P_total, c.a->index_rows_local ? (int64_t)d.group_begin * d.W : 0};
    hipLaunchKernelGGL(plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d.variant, c.a->ids, c.a->codes,
                       pd, c.group0(b) * d.W, count, d.compact ? c.at<const int32_t>(c.w.rank) : nullptr,
                       c.at<int32_t>(c.w.rowidx), c.at<int32_t>(c.w.egoidx), c.at<int32_t>(c.w.slotof));
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

// flags -> rank / list / seg for the (node, code) rows this call's paths touch (the hetero class: the whole batch's)
int run_compact_rows(const Call &c, hipStream_t s) {
    const Dims &d = c.d;
    const bool het = d.variant == PN_VARIANT_HETERO;
    const int64_t slot0 = het ? 0 : (int64_t)d.group_begin * d.W, count = (int64_t)(het ? d.S_total : d.S) * d.W, n = count * d.L;
    const int64_t rows = (int64_t)d.N * d.L;
    const PlanDims pd{d.S_total, d.W, d.L, d.N, d.P_total, c.a->index_rows_local ? (int64_t)d.group_begin * d.W : 0};
    uint8_t *flags = c.at<uint8_t>(c.w.flags);
    PN_CHECK_HIP(hipMemsetAsync(flags, 0, (size_t)rows, s));
    hipLaunchKernelGGL(mark_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d.variant, c.a->ids, c.a->codes, pd,
                       slot0, count, flags);
    const int nb = (int)((rows + SCAN_BLOCK - 1) / SCAN_BLOCK);
    int32_t *seg = c.at<int32_t>(c.w.seg), *bsum = c.at<int32_t>(c.w.bsum);
    hipLaunchKernelGGL(scan_count_kernel, dim3(nb), dim3(256), 0, s, flags, rows, bsum);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, s, bsum, nb, seg + d.L);
    hipLaunchKernelGGL(scan_rank_kernel, dim3(nb), dim3(256), 0, s, flags, rows, d.N, d.L, bsum, c.at<int32_t>(c.w.rank),
                       c.at<int32_t>(c.w.list), seg);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

// ---- the generic recurrence (H > 256): step-by-step kernels + the fp32 GEMM ------------------------------------------
GenParams gen_params(const Call &c, int b) {
    const Dims &d = c.d;
    const pn_pagg_args *a = c.a;
    GenParams gp{};
    gp.Z = c.Z;
    gp.rowidx = c.at<int32_t>(c.w.rowidx);
    gp.slotof = c.at<int32_t>(c.w.slotof);
    gp.xh = c.at<float>(c.w.xh);
    gp.pre = c.at<float>(c.w.dG);
    gp.hn = c.at<float>(c.w.hn);
    gp.gx = c.at<float>(c.w.gx);
    gp.dZ = c.at<float>(c.w.dZ);
    gp.biasc = c.at<float>(c.w.biasc);
    gp.P = c.groups(b) * d.W;
    gp.L = d.L;
    gp.H = d.H;
    gp.G = d.G;
    gp.cell = d.cell;
    gp.Pmask = d.P_total;
    gp.p_drop = a->p_seq;
    gp.seed = a->seed;
    gp.dyn = a->step_state;
    gp.mask = a->mask_seq;
    return gp;
}

int run_seq_fwd_generic(const Call &c, int b, bool save) {
    const Dims &d = c.d;
    GenParams gp = gen_params(c, b);
    gp.saved = save ? c.at<float>(c.w.saved) : nullptr;
    gp.state = c.at<float>(c.w.dhn);            // (a backward-only buffer: the running cell / hidden state lives there)
    const int H = d.H, GH = d.G * H, P = gp.P;
    const float *Wcat = c.at<const float>(c.w.Wp);
    const unsigned bx = (unsigned)(((int64_t)P * (H / 4) + 255) / 256), be = (unsigned)(((int64_t)P * H + 255) / 256);
    StageTimer tm(c.ctx, ST_SEQ_FWD, c.stream);
    for (int t = 0; t < d.L; t++) {
        gp.t = t;
        hipLaunchKernelGGL(gen_x_kernel, dim3(bx), dim3(256), 0, c.stream, gp);
        PN_CHECK_HIP(hipGetLastError());
        // pre[q, :] = [x_t | h_{t-1}] . Wcat^T + b      (step 0: h_{-1} = 0, the x half of K suffices)
        if (int rc = launch_gemm3(c.stream, gp.xh + (size_t)t * 2 * H, (int64_t)d.L * 2 * H, Wcat, 2 * H,
                                  gp.pre + (size_t)t * GH, (int64_t)d.L * GH, gp.biasc, P, GH, t == 0 ? H : 2 * H))
            return rc;
        hipLaunchKernelGGL(gen_cell_fwd_kernel, dim3(be), dim3(256), 0, c.stream, gp);
        PN_CHECK_HIP(hipGetLastError());
    }
    return PN_OK;
}

// BPTT of micro-batch b (the weight gradient dG^T . [x | h] is the fused path's wgrad3_kernel: it has no size limit)
int run_seq_bwd_generic(const Call &c, int b) {
    const Dims &d = c.d;
    GenParams gp = gen_params(c, b);
    gp.saved = c.at<float>(c.w.saved);
    if (d.det) gp.rowidx = c.at<int32_t>(c.w.iota), gp.dZ = c.at<float>(c.w.dx);     // (see det_scatter_kernel)
    gp.dh = c.at<float>(c.w.dhn);               // d loss / d h_n from the pooling backward, then d h_{t-1} step by step
    gp.state = c.at<float>(c.w.hn);             // (the pooling backward is done with h_n: d c / the GRU's direct term live there)
    const int H = d.H, GH = d.G * H, P = gp.P;
    const float *WcatT = c.at<const float>(c.w.WpT);
    float *gx = c.at<float>(c.w.gx);
    const unsigned bx = (unsigned)(((int64_t)P * (H / 4) + 255) / 256), be = (unsigned)(((int64_t)P * H + 255) / 256);
    {
        StageTimer tm(c.ctx, ST_SEQ_BWD, c.stream);
        for (int t = d.L - 1; t >= 0; t--) {
            gp.t = t;
            hipLaunchKernelGGL(gen_cell_bwd_kernel, dim3(be), dim3(256), 0, c.stream, gp);
            PN_CHECK_HIP(hipGetLastError());
            // [dx_t | dh_{t-1}] = dG_t . [W_ih | W_hh]      (step 0: the dx half suffices)
            if (int rc = launch_gemm3(c.stream, gp.pre + (size_t)t * GH, (int64_t)d.L * GH, WcatT, GH, gx, 2 * H, nullptr, P,
                                      t == 0 ? H : 2 * H, GH))
                return rc;
            hipLaunchKernelGGL(gen_scatter_kernel, dim3(bx), dim3(256), 0, c.stream, gp);
            PN_CHECK_HIP(hipGetLastError());
        }
    }
    return PN_OK;
}

// true when max |Z| (SeqRange.x, the fp16 recurrence's operand range) is taken in the bank GEMM's epilogue: dense bank and
// fc0 both on gemm_kernel in this call -- fc0's launch clears the slot, the bank's adds to it, both on the caller's stream
// (range_rows_kernel and its place on the critical path between the bank and the recurrence are then not needed, and
// range_part_kernel on the packing stream must NOT clear the slot: it would race the bank)
inline bool bank_epilogue_range(const Call &c) {
    const Dims &d = c.d;
    const pn_pagg_args *a = c.a;
    return d.math == PN_SEQ_MATH_F16X2 && d.G > 0 && !d.generic && !d.compact && a->reuse_tables == 0 && !a->Xh_in &&
           !gemm3_pays(c.ctx, d.N, d.H, d.F, G3_FC0) && !gemm3_pays(c.ctx, d.N, d.L * d.H, d.H, G3_BANK);
}

int run_pack_fwd(const Call &c, hipStream_t s) {
    const Dims &d = c.d;
    if (d.G == 0) return PN_OK;         // mean / sum: nothing to pack
    if (d.generic) {
        const int64_t n = (int64_t)d.G * d.H * 2 * d.H;
        hipLaunchKernelGGL(gen_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c.a->w_ih, c.a->w_hh,
                           c.a->b_ih, c.a->b_hh, d.H, d.G, d.cell == CELL_GRU ? 1 : 0, c.at<float>(c.w.Wp),
                           c.at<float>(c.w.WpT), c.at<float>(c.w.biasc));
        PN_CHECK_HIP(hipGetLastError());
        return PN_OK;
    }
    if (d.math == PN_SEQ_MATH_F16X2) {      // two fp16 planes, scaled by the weights' own maxima (pn_seqh.hip)
        SeqRange *rg = c.at<SeqRange>(c.w.range);
        // (it also clears the slots this call's atomicMax launches add to; a reused dense bank keeps its range.x.)  A training
        // forward packs the BPTT's planes in the same launch, on the side stream under fc0 / the bank: the backward that follows
        // on this workspace then starts on its first real kernel (an inference forward uses that slot for Wcat)
        return launch_pack_fb(s, c.a->w_ih, c.a->w_hh, c.a->b_ih, c.a->b_hh, d.H, d.G, d.Gw, d.cell == CELL_GRU ? 1 : 0,
                              (d.compact || c.a->reuse_tables != 1) && !bank_epilogue_range(c), rg, c.at<void>(c.w.wmax),
                              c.at<void>(c.w.Wp), c.at<float>(c.w.biasc), c.a->no_save ? nullptr : c.at<void>(c.w.WpT));
    }
    return launch_pack_fwd3(s, c.a->w_ih, c.a->w_hh, c.a->b_ih, c.a->b_hh, d.H, d.G, d.cell == CELL_GRU ? 1 : 0, c.at<void>(c.w.Wp),
                            c.at<float>(c.w.biasc));
}

// the order-agnostic encoders (cell = mean / sum): forward or backward of micro-batch b
int run_seq_reduce(const Call &c, int b, bool backward) {
    const Dims &d = c.d;
    const pn_pagg_args *a = c.a;
    SeqReduceParams rp{};
    rp.Z = c.Z;
    rp.rowidx = c.at<int32_t>(c.w.rowidx);
    rp.slotof = c.at<int32_t>(c.w.slotof);
    rp.hn = c.at<float>(c.w.hn);
    rp.dhn = c.at<float>(c.w.dhn);
    rp.dZ = c.at<float>(c.w.dZ);
    rp.P = c.groups(b) * d.W;
    rp.L = d.L;
    rp.H = d.H;
    rp.Pmask = d.P_total;
    rp.scale = d.cell == CELL_MEAN ? 1.0f / (float)d.L : 1.0f;
    rp.p_drop = a->p_seq;
    rp.seed = a->seed;
    rp.dyn = a->step_state;
    rp.mask = a->mask_seq;
    if (backward && d.det) rp.rowidx = c.at<int32_t>(c.w.iota), rp.dZ = c.at<float>(c.w.dx);     // (see det_scatter_kernel)
    const int64_t n = (int64_t)rp.P * (d.H / 4);
    StageTimer tm(c.ctx, backward ? ST_SEQ_BWD : ST_SEQ_FWD, c.stream);
    if (backward)
        hipLaunchKernelGGL(seq_reduce_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, rp);
    else
        hipLaunchKernelGGL(seq_reduce_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, rp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

// Inference without dropout on the fp16 kernels: the input half of the gates is applied to the rows of the bank once
// (ZW = Z . W_ih^T + b, seq_fwdzw_kernel in pn_seqh.hip) when the table fits the slot the saved gates would take.
// PN_EVAL_ZW=0 switches it off (A/B runs, tests of the plain inference path).
inline bool use_zw(const Call &c) {
    const Dims &d = c.d;
    const pn_pagg_args *a = c.a;
    if (d.math != PN_SEQ_MATH_F16X2 || !a->no_save || a->p_seq > 0.0f || a->mask_seq) return false;
    if (knobs_of(c.ctx).eval_zw == 0) return false;
    const size_t need = (size_t)d.ZR * d.G * d.H * 4, have = (size_t)d.Sb * d.W * d.L * d.SV * d.H * 4;
    return need <= have && need <= ((size_t)2 << 30);
}

// bound of the factor the sequence dropout applies to a gathered row: 1 / (1 - p), or 16 for explicit masks (the contract
// of pn_pagg_shape.seq_math)
inline float seq_xmul(const pn_pagg_args *a) {
    return a->mask_seq ? 16.0f : a->p_seq > 0.0f ? 1.0f / (1.0f - a->p_seq) : 1.0f;
}

int run_seq_fwd(const Call &c, int b, bool save) {
    const Dims &d = c.d;
    const pn_pagg_args *a = c.a;
    if (d.G == 0) return run_seq_reduce(c, b, false);
    if (d.generic) return run_seq_fwd_generic(c, b, save);
    SeqFwdParams sp{};
    sp.Z = c.Z;
    sp.rowidx = c.at<int32_t>(c.w.rowidx);
    sp.slotof = c.at<int32_t>(c.w.slotof);
    sp.Wp = c.at<float>(c.w.Wp);
    sp.biasc = c.at<float>(c.w.biasc);
    sp.hn = c.at<float>(c.w.hn);
    sp.saved = save ? c.at<float>(c.w.saved) : nullptr;
    sp.xh = save ? c.at<float>(c.w.xh) : nullptr;
    sp.keep = (!save || a->mask_seq || !(a->p_seq > 0.0f)) ? nullptr : c.at<uint8_t>(c.w.keep);
    sp.P = c.groups(b) * d.W;
    sp.L = d.L;
    sp.Pmask = d.P_total;
    sp.p_drop = a->p_seq;
    sp.seed = a->seed;
    sp.dyn = a->step_state;
    sp.mask = a->mask_seq;
    StageTimer tm(c.ctx, ST_SEQ_FWD, c.stream);
    if (!save && use_zw(c)) {
        sp.range = c.at<SeqRange>(c.w.range);
        sp.ZW = c.at<const float>(c.w.saved);
        return launch_seq_fwdzw(c.ctx, c.stream, d.H, d.cell == CELL_GRU ? 3 : d.cell == CELL_LSTM ? 4 : 1, sp);
    }
    if (d.math == PN_SEQ_MATH_F16X2) {
        sp.range = c.at<SeqRange>(c.w.range);
        sp.xmul = seq_xmul(a);
        return launch_seq_fwdh(c.ctx, c.stream, d.H, d.cell == CELL_GRU ? 3 : d.cell == CELL_LSTM ? 4 : 1, sp);
    }
    return launch_seq_fwd3(c.ctx, c.stream, d.H, d.cell == CELL_GRU ? 3 : d.cell == CELL_LSTM ? 4 : 1, sp);
}

PoolParams pool_fwd_params(const Call &c, int b, float *out) {
    const Dims &d = c.d;
    const pn_pagg_args *a = c.a;
    const int homo = d.variant == PN_VARIANT_HOMO;
    PoolParams pp{};
    pp.variant = d.variant;
    pp.S = c.groups(b);
    pp.W = d.W;
    pp.H = d.H;
    pp.C = d.C;
    pp.N = d.N;
    pp.goff = c.group0(b);
    pp.hn = c.at<float>(c.w.hn);
    pp.ego_tab = homo ? c.Z : c.Xh;
    pp.egoidx = c.at<int32_t>(c.w.egoidx);
    pp.Xh = c.Xh;
    pp.sel = c.sel(b);
    pp.att_w = a->att_w;
    pp.att_b = a->att_b;
    pp.fc2_w = a->fc2_w;
    pp.fc2_b = a->fc2_b;
    pp.p_drop = a->p_cls;
    pp.seed = a->seed;
    pp.dyn = a->step_state;
    pp.mask = a->mask_cls;
    pp.coef = c.at<float>(c.w.coef);
    pp.rawsc = c.at<float>(c.w.rawsc);
    pp.layer1 = c.at<float>(c.w.layer1);
    pp.out = out;
    return pp;
}
inline size_t pool_fwd_lds_bytes(const Dims &d) { return (size_t)(2 * d.W + 6 * d.H) * sizeof(float); }
int run_pool_fwd(const Call &c, int b, float *out) {
    const PoolParams pp = pool_fwd_params(c, b, out);
    StageTimer tm(c.ctx, ST_POOL_FWD, c.stream);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(pp.S), dim3(256), pool_fwd_lds_bytes(c.d), c.stream, pp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int check_forward_args(const Call &c, const char *who) {
    const pn_pagg_args *a = c.a;
    const Dims &d = c.d;
    if (!a->ids || !a->codes || !a->sel || !a->bank_w || !a->bank_b || !a->fc2_w || !a->fc2_b || !a->out)
        PN_FAIL(PN_ERR_ARG, "%s: null tensor", who);
    if (d.G > 0 && (!a->w_ih || !a->w_hh || !a->b_ih || !a->b_hh)) PN_FAIL(PN_ERR_ARG, "%s: recurrent weights missing", who);
    if (!a->Xh_in && !a->reuse_tables && (!a->X || !a->fc0_w || !a->fc0_b)) PN_FAIL(PN_ERR_ARG, "%s: X / fc0 missing", who);
    if (d.variant != PN_VARIANT_PAGG && (!a->att_w || !a->att_b)) PN_FAIL(PN_ERR_ARG, "%s: attention weights missing", who);
    return PN_OK;
}

// What a forward does before its first micro-batch: the touched rows, the index plan of micro-batch 0 and the packed
// weights (second stream), Xh = fc0(X), Z = bank(Xh); joined, so that the recurrence can follow.
// side_extra: one more launch for the second stream (pn_pagg_train_step: the zero fill of the backward's accumulators)
int run_tables(const Call &c, JoinGuard &joiner, const std::function<int(hipStream_t)> *side_extra = nullptr) {
    pn_context *ctx = c.ctx;
    hipStream_t stream = c.stream;
    const pn_pagg_args *a = c.a;
    const Dims &d = c.d;
    const int H = d.H, L = d.L;
    const int homo = d.variant == PN_VARIANT_HOMO;
    const bool epi_range = bank_epilogue_range(c);
    // the index plan (of the first micro-batch) and the weight packing do not depend on fc0 / bank: second stream,
    // joined before the recurrence
    if (d.compact) {        // the compact rows first: the index plan and the bank both read them
        StageTimer tm(ctx, ST_PLAN_PACK, stream);
        if (int rc = run_compact_rows(c, stream)) return rc;
    }
    hipStream_t pstream = stream;
    if (!profiling_every_stage(ctx))
        if (void *side = context_fork(ctx, stream)) pstream = (hipStream_t)side;
    {
        StageTimer tm(ctx, ST_PLAN_PACK, pstream);
        if (int rc = run_plan(c, pstream, 0)) return rc;
        if (int rc = run_pack_fwd(c, pstream)) return rc;
    }
    if (side_extra)
        if (int rc = (*side_extra)(pstream)) return rc;
    if (pstream != stream)
        if (int rc = joiner.mark()) return rc;
    if (!a->reuse_tables) {
        // fc0 (+ReLU for HOMO): Xh = X . fc0_w^T + fc0_b
        if (!a->Xh_in) {
            StageTimer tm(ctx, ST_FC0, stream);
            if (gemm3_pays(c.ctx, d.N, H, d.F, G3_FC0)) {          // large graphs: fp32 results from the bf16 matrix pipe (six MFMAs per product)
                if (int rc = launch_gemm3(stream, a->X, d.F, a->fc0_w, d.F, c.at<float>(c.w.Xh), H, a->fc0_b, d.N, H, d.F, homo))
                    return rc;
            } else if (int rc = launch_gemm_split(stream, a->X, d.F, 1, nullptr, a->fc0_w, d.F, 1, c.at<float>(c.w.Xh), H,
                                                  a->fc0_b, d.N, H, d.F, homo, GEMM_STORE, c.at<float>(c.w.gpart),
                                                  epi_range ? &c.at<SeqRange>(c.w.range)->x : nullptr))
                return rc;
        }
    }
    // node-sharded path: the caller's all-gather of Xh runs on its own stream; this one waits for it here, where Xh is first
    // read -- everything above (touched rows, plan, packing) has been enqueued under it
    if (a->Xh_in && a->Xh_ready) PN_CHECK_HIP(hipStreamWaitEvent(stream, (hipEvent_t)a->Xh_ready, 0));
    if (d.compact) {
        // distance bank over the touched rows: code by code, Z[compact row] = act(Xh[node] . bank_w[code]^T + bank_b[code])
        // (the touched set belongs to this batch: reuse_tables keeps the projected features only)
        StageTimer tm(ctx, ST_BANK, stream);
        const int mmax = (int)std::min<int64_t>(d.N, d.ZR);
        for (int code = 0; code < L; code++) {
            if (gemm3_pays(c.ctx, mmax, H, H, G3_BANK)) {
                if (int rc = launch_gemm3(stream, c.Xh, H, a->bank_w + (size_t)code * H * H, H, c.Z, H, a->bank_b + (size_t)code * H,
                                          mmax, H, H, homo, 0, nullptr, GEMM_IND_A_ROWS, c.at<const int32_t>(c.w.seg) + code,
                                          c.at<const int32_t>(c.w.list)))
                    return rc;
            } else if (int rc = launch_gemm(stream, c.Xh, H, 1, nullptr, a->bank_w + (size_t)code * H * H, H, 1, c.Z, H,
                                            a->bank_b + (size_t)code * H, mmax, H, H, homo, GEMM_STORE, 1, nullptr,
                                            GEMM_IND_A_ROWS, c.at<const int32_t>(c.w.seg) + code, c.at<const int32_t>(c.w.list)))
                return rc;
        }
    } else if (a->reuse_tables != 1) {      // (1: the dense Z of the previous forward is still valid; 2: only Xh is)
        // distance bank over every node: Z[v, d, :] = act(Xh[v] . bank_w[d]^T + bank_b[d])
        StageTimer tm(ctx, ST_BANK, stream);
        if (gemm3_pays(c.ctx, d.N, L * H, H, G3_BANK)) {
            if (int rc = launch_gemm3(stream, c.Xh, H, a->bank_w, H, c.Z, (int64_t)L * H, a->bank_b, d.N, L * H, H, homo)) return rc;
        } else if (int rc = launch_gemm(stream, c.Xh, H, 1, nullptr, a->bank_w, H, 1, c.Z, (int64_t)L * H, a->bank_b, d.N,
                                        L * H, H, homo, GEMM_STORE, 1, nullptr, GEMM_IND_NONE, nullptr, nullptr,
                                        epi_range ? &c.at<SeqRange>(c.w.range)->x : nullptr))
            return rc;
    }
    // the fp16 recurrence scales the gathered rows by a power of two taken from the largest |Z| of the rows just computed
    // (a reused dense Z keeps its record: it sits next to Z in the workspace)
    if (int rc = joiner.join()) return rc;      // the recurrence needs the plan and the packed weights
    // (after the join: the packing stream's range_part_kernel cleared the slot this launch adds to)
    if (d.math == PN_SEQ_MATH_F16X2 && (d.compact || a->reuse_tables != 1) && !epi_range) {
        StageTimer tm(ctx, ST_BANK, stream);
        if (int rc = launch_range_rows(stream, c.Z, d.ZR, H, d.compact ? c.at<const int32_t>(c.w.seg) + L : nullptr,
                                       c.at<SeqRange>(c.w.range)))
            return rc;
    }
    if (use_zw(c)) {
        // ZW [rows of Z, G*H] = Z . W_ih^T + (b_ih + b_hh)  (GRU: its four slots): Wcat in the slot of the backward's packed
        // weights, the table in the slot of the saved gates -- neither is used by an inference forward
        StageTimer tm(ctx, ST_BANK, stream);
        const int GH = d.G * H;
        const int64_t n = (int64_t)GH * 2 * H;
        float *Wcat = c.at<float>(c.w.WpT);
        hipLaunchKernelGGL(gen_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a->w_ih, a->w_hh, a->b_ih,
                           a->b_hh, H, d.G, d.cell == CELL_GRU ? 1 : 0, Wcat, (float *)nullptr, c.at<float>(c.w.biasc));
        PN_CHECK_HIP(hipGetLastError());
        const int M = (int)d.ZR;
        const bool g3 = H % G3_KT == 0 && (int64_t)((M + G3_BM - 1) / G3_BM) * ((GH + G3_BN - 1) / G3_BN) >= 256;
        if (g3) {
            if (int rc = launch_gemm3(stream, c.Z, H, Wcat, 2 * H, c.at<float>(c.w.saved), GH, c.at<float>(c.w.biasc), M, GH, H)) return rc;
        } else if (int rc = launch_gemm(stream, c.Z, H, 1, nullptr, Wcat, 2 * H, 1, c.at<float>(c.w.saved), GH, c.at<float>(c.w.biasc), M,
                                        GH, H, 0, GEMM_STORE, 1))
            return rc;
    }
    return PN_OK;
}

// ---- deterministic mode: the three scatters of a backward and their destination orders ------------------------------------
// dst[key[i]] += contribution i, added in the order of i, for  DET_SEL  the ego half of d layer1 onto the masked nodes' rows
// of d Xh (keys: sel),  DET_EGO  the attention-ego term onto the ego rows (keys: egoidx),  DET_ROW  the gather backward
// (keys: rowidx).  All three key arrays come out of the index plan -- nothing of the backward enters them -- so their
// stable sort by key runs ONCE per micro-batch, right behind the plan and on the context's second stream under the
// forward's recurrence (round 5; until then: a radix sort in front of each scatter, on the backward's critical path --
// 0.10 of the 0.27 ms the mode cost at the headline shape were the pooling backward's two sorts alone).
enum { DET_SEL = 0, DET_EGO = 1, DET_ROW = 2 };
struct DetOrder {
    const int32_t *keys;
    int64_t K, max_key;
};
inline DetOrder det_order(const Call &c, int which, int b) {
    const Dims &d = c.d;
    const int64_t Sb = c.groups(b), Pb = Sb * d.W;
    const bool homo = d.variant == PN_VARIANT_HOMO;
    switch (which) {
        case DET_SEL: return DetOrder{c.sel(b), Sb, (int64_t)d.N - 1};
        case DET_EGO: return DetOrder{c.at<const int32_t>(c.w.egoidx), Pb, homo ? d.ZR - 1 : (int64_t)d.N - 1};
        default: return DetOrder{c.at<const int32_t>(c.w.rowidx), Pb * d.L, d.ZR - 1};
    }
}

int prepare_det_orders(const Call &c, hipStream_t s, int b) {
    const bool has_att = c.d.variant != PN_VARIANT_PAGG;
    DetKeys3 q{};
    int64_t total = 0, max_key = 0;
    for (int which = 0; which < 3; which++) {
        const DetOrder o = det_order(c, which, b);
        q.keys[which] = o.keys;
        q.n[which] = (which == DET_EGO && !has_att) ? 0 : std::max<int64_t>(o.K, 0);
        q.hi[which] = (int32_t)o.max_key;
        total += q.n[which];
        if (q.n[which]) max_key = std::max(max_key, o.max_key);
    }
    if (total <= 0) return PN_OK;
    int bits = 1;
    while (bits < 31 && (max_key >> bits) != 0) bits++;
    if (bits + 2 > 31) PN_FAIL(PN_ERR_ARG, "deterministic mode: %lld table rows leave no room for the scatters' segment bits", (long long)max_key + 1);
    q.shift = bits;
    int32_t *ck = c.at<int32_t>(c.w.keys), *pos = c.at<int32_t>(c.w.skey);
    hipLaunchKernelGGL(det_keys3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, q, ck, pos);
    PN_CHECK_HIP(hipGetLastError());
    // (okey / osrc are the heads of the two contiguous output blocks; an empty EGO segment leaves its slot unused: the
    //  segments that follow start where the sort puts them only if the layout matches -- so the outputs are addressed by the
    //  running sum of the ACTUAL counts, which equals the layout's whenever a segment is present)
    return sort_pairs_i32(c.at<void>(c.w.stmp), c.w.stmp_bytes, ck, c.at<int32_t>(c.w.okey), pos, c.at<int32_t>(c.w.osrc), total,
                          bits + 2, s);
}

// the orders of micro-batch b on the second stream when there is one (joined by the guard on the way out), else in line
int fork_det_orders(const Call &c, JoinGuard &joiner, int b) {
    hipStream_t s = c.stream;
    if (!profiling_every_stage(c.ctx))
        if (void *side = context_fork(c.ctx, c.stream)) s = (hipStream_t)side;
    if (int rc = prepare_det_orders(c, s, b)) return rc;
    if (s != c.stream) return joiner.mark();
    return PN_OK;
}

// where the one sort left scatter `which` of micro-batch b: its offset is the sum of the segments before it that exist
DetScatterParams det_params(const Call &c, int which, int b, const float *rows, const float *scal, const float *vec, float *dst) {
    const bool has_att = c.d.variant != PN_VARIANT_PAGG;
    int64_t off = 0, max_key = 0;
    for (int w = 0; w < 3; w++) {
        const DetOrder o = det_order(c, w, b);
        const int64_t n = (w == DET_EGO && !has_att) ? 0 : std::max<int64_t>(o.K, 0);
        if (w < which) off += n;
        if (n) max_key = std::max(max_key, o.max_key);
    }
    int bits = 1;
    while (bits < 31 && (max_key >> bits) != 0) bits++;
    const int64_t K = (which == DET_EGO && !has_att) ? 0 : std::max<int64_t>(det_order(c, which, b).K, 0);
    return DetScatterParams{c.at<int32_t>(c.w.okey) + off, c.at<int32_t>(c.w.osrc) + off, (int32_t)((1u << bits) - 1u), K,
                            off / DET_CHUNK + which, rows, scal, vec, dst, c.at<float>(c.w.cpart), c.d.H};
}

// one or two scatters (the pooling backward's pair) per launch: pass 1, then pass 2
int run_det_scatters(const Call &c, const DetScatterParams *list, int count) {
    DetScatterPair pp{};
    int64_t kmax = 0;
    for (int i = 0; i < count; i++) pp.s[i] = list[i], kmax = std::max(kmax, list[i].K);
    if (kmax <= 0) return PN_OK;
    const int gpb = 256 / (c.d.H / 4);
    const int64_t chunks = (kmax + DET_CHUNK - 1) / DET_CHUNK;
    const dim3 grid((unsigned)((chunks + gpb - 1) / gpb), (unsigned)count);
    hipLaunchKernelGGL(det_scatter_kernel<1>, grid, dim3(256), 0, c.stream, pp);
    hipLaunchKernelGGL(det_scatter_kernel<2>, grid, dim3(256), 0, c.stream, pp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int run_det_scatter(const Call &c, int which, int b, const float *rows, const float *scal, const float *vec, float *dst) {
    const DetScatterParams p = det_params(c, which, b, rows, scal, vec, dst);
    return run_det_scatters(c, &p, 1);
}

}  // namespace

extern "C" {

int pn_pagg_workspace_bytes(const pn_pagg_shape *shape, int64_t *bytes) {
    if (!shape || !bytes) PN_FAIL(PN_ERR_ARG, "pn_pagg_workspace_bytes: null");
    Dims d;
    if (int rc = make_dims(*shape, d)) return rc;
    *bytes = (int64_t)ws_layout(d).total;
    return PN_OK;
}

int pn_pagg_shape_info(const pn_pagg_shape *shape, int64_t out[4]) {
    if (!shape || !out) PN_FAIL(PN_ERR_ARG, "pn_pagg_shape_info: null");
    Dims d;
    if (int rc = make_dims(*shape, d)) return rc;
    out[0] = d.compact ? 1 : 0;
    out[1] = d.ZR;
    out[2] = d.nb;
    out[3] = d.math;
    return PN_OK;
}

int pn_gemm_f32(const float *A, int64_t sAm, int64_t sAk, const float *B, int64_t sBn, int64_t sBk, float *C,
                int64_t ldc, const float *bias, int32_t M, int32_t N, int32_t K, int32_t relu, void *stream_) {
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) PN_FAIL(PN_ERR_ARG, "pn_gemm_f32: bad argument");
    if ((sAm != 1 && sAk != 1) || (sBn != 1 && sBk != 1)) PN_FAIL(PN_ERR_ARG, "pn_gemm_f32: one stride per operand must be 1");
    return launch_gemm((hipStream_t)stream_, A, sAm, sAk, nullptr, B, sBn, sBk, C, ldc, bias, M, N, K, relu,
                       GEMM_STORE, 1);
}

int pn_linear_forward(pn_context *ctx, const float *X, const float *W, const float *b, int32_t rows, int32_t in_f,
                      int32_t out_f, int32_t relu, float *Y, void *workspace, int64_t workspace_bytes, void *stream_) {
    static_assert(PN_LINEAR_SPLIT_MAX == GEMM_MAX_SPLIT, "header and kernel agree on the split bound");
    if (!X || !W || !Y || rows < 0 || in_f < 1 || out_f < 1) PN_FAIL(PN_ERR_ARG, "pn_linear_forward: bad argument");
    if (int rc = context_check_device(ctx)) return rc;
    if (rows == 0) return PN_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const bool split = workspace && workspace_bytes >= (int64_t)GEMM_MAX_SPLIT * rows * out_f * (int64_t)sizeof(float);
    StageTimer tm(ctx, ST_FC0, stream);
    return launch_gemm_split(stream, X, in_f, 1, nullptr, W, in_f, 1, Y, out_f, b, rows, out_f, in_f, relu, GEMM_STORE,
                             split ? reinterpret_cast<float *>(workspace) : nullptr);
}

int pn_linear_backward(pn_context *ctx, const float *dY, const float *gate, const float *X, const float *W, int32_t rows,
                       int32_t in_f, int32_t out_f, float *g_W, float *g_b, float *g_X, void *workspace,
                       int64_t workspace_bytes, void *stream_) {
    static_assert(PN_LINEAR_BWD_SPLIT_MAX == DET_MAX_SPLIT, "header and kernel agree on the split bound");
    hipStream_t stream = (hipStream_t)stream_;
    if (!dY || rows < 0 || in_f < 1 || out_f < 1) PN_FAIL(PN_ERR_ARG, "pn_linear_backward: bad argument");
    if (int rc = context_check_device(ctx)) return rc;
    StageTimer tm(ctx, ST_FC0_BWD, stream);
    if (g_W && !X) PN_FAIL(PN_ERR_ARG, "pn_linear_backward: g_W needs X");
    if (g_W || g_b) {       // both are accumulated with atomics: one zero-fill launch for the two
        ZeroList zl{};
        if (g_W) {
            zl.ptr[zl.n] = g_W;
            zl.count[zl.n++] = (unsigned long long)out_f * in_f;
        }
        if (g_b) {
            zl.ptr[zl.n] = g_b;
            zl.count[zl.n++] = (unsigned long long)out_f;
        }
        hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, stream, zl);
        PN_CHECK_HIP(hipGetLastError());
    }
    if (rows == 0) return PN_OK;
    // with a workspace: chunk sums added in a fixed order (bitwise reproducible); without: chunks race with atomics
    const bool det = workspace && workspace_bytes >= (int64_t)(det_gemm_floats(out_f, in_f) * sizeof(float));
    if (g_W && det) {
        if (int rc = launch_gemm_det(stream, dY, 1, out_f, gate, X, 1, in_f, g_W, in_f, out_f, in_f, rows, (rows + 255) / 256,
                                     g_b, reinterpret_cast<float *>(workspace)))
            return rc;
    } else if (g_W) {       // g_b rides along as the row sums of the (gated) A operand dY^T
        if (int rc = launch_gemm(stream, dY, 1, out_f, gate, X, 1, in_f, g_W, in_f, nullptr, out_f, in_f, rows, 0,
                                 GEMM_ATOMIC, (rows + 255) / 256, g_b))
            return rc;
    } else if (g_b) {
        if (int rc = launch_colsum(stream, dY, gate, out_f, rows, out_f, g_b, workspace != nullptr)) return rc;
    }
    if (g_X) {
        if (!W) PN_FAIL(PN_ERR_ARG, "pn_linear_backward: g_X needs W");
        if (int rc = launch_gemm(stream, dY, out_f, 1, gate, W, 1, in_f, g_X, in_f, nullptr, rows, in_f, out_f, 0,
                                 GEMM_STORE, 1))
            return rc;
    }
    return PN_OK;
}


int pn_pagg_paths_stream(pn_context *ctx, const pn_pagg_shape *shape, void *stream, void **out) {
    if (!shape || !out) PN_FAIL(PN_ERR_ARG, "pn_pagg_paths_stream: null");
    Dims d;
    if (int rc = make_dims(*shape, d)) return rc;
    // (run_tables: the plan of micro-batch 0 goes to the second stream unless every stage is timed; later micro-batches
    //  are planned on the caller's stream)
    void *side = (ctx && d.nb == 1 && !profiling_every_stage(ctx)) ? context_side_stream(ctx) : nullptr;
    *out = side ? side : stream;
    return PN_OK;
}

int pn_pagg_range_offset(const pn_pagg_shape *shape, int64_t *offset) {
    if (!shape || !offset) PN_FAIL(PN_ERR_ARG, "pn_pagg_range_offset: null");
    Dims d;
    if (int rc = make_dims(*shape, d)) return rc;
    static_assert(sizeof(SeqRange) == sizeof(pn_seq_range), "pn_seq_range is the public face of SeqRange");
    *offset = (d.math == PN_SEQ_MATH_F16X2 && d.G > 0 && !d.generic) ? (int64_t)ws_layout(d).range : -1;
    return PN_OK;
}

int pn_pagg_debug_offsets(const pn_pagg_shape *shape, int64_t out[4]) {
    if (!shape || !out) PN_FAIL(PN_ERR_ARG, "pn_pagg_debug_offsets: null");
    Dims d;
    if (int rc = make_dims(*shape, d)) return rc;
    const WsLayout w = ws_layout(d);
    out[0] = (int64_t)w.Xh;
    out[1] = (int64_t)w.Z;
    out[2] = (int64_t)w.hn;
    out[3] = (int64_t)w.layer1;
    return PN_OK;
}

int pn_pagg_gather(pn_context *ctx, const pn_pagg_shape *shape, const float *table, const int32_t *ids,
                   const uint8_t *codes, float *rows, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!shape || !table || !ids || !codes || !rows) PN_FAIL(PN_ERR_ARG, "pn_pagg_gather: null");
    if (int rc = context_check_device(ctx)) return rc;
    const pn_pagg_shape &s = *shape;
    if (s.S < 0 || s.W < 1 || s.L < 1 || s.H < 1 || s.N < 1) PN_FAIL(PN_ERR_ARG, "pn_pagg_gather: bad shape");
    const int S_total = s.S_total > 0 ? s.S_total : s.S;
    if (s.group_begin < 0 || (int64_t)s.group_begin + s.S > S_total) PN_FAIL(PN_ERR_ARG, "pn_pagg_gather: bad slice");
    const int64_t count = (int64_t)s.S * s.W, rowsn = count * s.L;
    if (rowsn == 0) return PN_OK;
    const bool vec = (s.H % 4 == 0) && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(rows)) % 16 == 0);
    const int64_t work = rowsn * (vec ? s.H / 4 : s.H);
    const int blocks = (int)std::min<int64_t>((work + 255) / 256, 256 * 32);
    const PlanDims pd{S_total, s.W, s.L, s.N, (int64_t)S_total * s.W, 0};
    const int64_t slot_begin = (int64_t)s.group_begin * s.W;
    StageTimer tm(ctx, ST_GATHER, stream);
    if (vec)
        hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(256), 0, stream, s.variant, table, ids, codes, pd,
                           slot_begin, count, s.H, rows);
    else
        hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, stream, s.variant, table, ids, codes, pd,
                           slot_begin, count, s.H, rows);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int pn_pagg_forward(pn_context *ctx, const pn_pagg_args *a, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    Call c;
    if (int rc = resolve_call(c, ctx, a, stream, "pn_pagg_forward")) return rc;
    const Dims &d = c.d;
    if (int rc = check_forward_args(c, "pn_pagg_forward")) return rc;
    if (d.S == 0) return PN_OK;
    JoinGuard joiner{ctx, stream};
    if (int rc = run_tables(c, joiner)) return rc;
    const bool save = d.nb == 1 && !a->no_save;  // several micro-batches: the backward re-runs each one's recurrence
    if (save && d.det)                           // the backward's scatter orders: second stream, under the recurrence
        if (int rc = fork_det_orders(c, joiner, 0)) return rc;
    for (int b = 0; b < d.nb; b++) {
        if (b > 0) {
            StageTimer tm(ctx, ST_PLAN_PACK, stream);
            if (int rc = run_plan(c, stream, b)) return rc;
        }
        if (int rc = run_seq_fwd(c, b, save)) return rc;
        if (int rc = run_pool_fwd(c, b, a->out + (size_t)b * d.Sb * d.C)) return rc;
    }
    return PN_OK;
}

// pn_pagg_backward, and -- with `target` -- pn_pagg_train_step: the forward of every micro-batch runs right before its
// backward (saved tensors in place), the loss gradient of its rows is formed in between
static int pagg_backward_impl(pn_context *ctx, const pn_pagg_args *a, void *stream_, const int64_t *target, float grad_scale,
                              float *loss) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool fused = target != nullptr;
    Call c;
    if (int rc = resolve_call(c, ctx, a, stream, fused ? "pn_pagg_train_step" : "pn_pagg_backward")) return rc;
    const Dims &d = c.d;
    if (fused) {
        if (d.S > 0)
            if (int rc = check_forward_args(c, "pn_pagg_train_step")) return rc;
        // (one micro-batch: the loss kernel overwrites -- launch_cross_entropy zero-fills only if it needs several workgroups)
        if (d.nb > 1 || d.S == 0) PN_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), stream));
    }
    // (S == 0 -- a rank of a sharded batch without masked nodes: its index arrays and g_out are empty tensors, i.e. NULL;
    //  the call still zero-fills every gradient it was given)
    if (!a->bank_w || !a->fc2_w || (d.S > 0 && (!a->ids || !a->codes || !a->sel || (!fused && !a->g_out))))
        PN_FAIL(PN_ERR_ARG, "pn_pagg_backward: null tensor");
    if (d.G > 0 && (!a->w_ih || !a->w_hh)) PN_FAIL(PN_ERR_ARG, "pn_pagg_backward: recurrent weights missing");
    if (!a->Xh_in && (!a->X || !a->fc0_w)) PN_FAIL(PN_ERR_ARG, "pn_pagg_backward: X / fc0 missing");
    if (a->Xh_in && !a->g_Xh) PN_FAIL(PN_ERR_ARG, "pn_pagg_backward: g_Xh is required with Xh_in");
    if (d.nb > 1 && !fused && ((d.G > 0 && (!a->b_ih || !a->b_hh)) || !a->fc2_b || (d.variant != PN_VARIANT_PAGG && !a->att_b)))
        PN_FAIL(PN_ERR_ARG, "pn_pagg_backward: micro-batches re-run the forward and need every forward tensor");
    const int H = d.H, L = d.L, G = d.G, GH = G * H, GwH = d.Gw * H;     // GH: gate slots, GwH: rows of the caller's weights
    const int homo = d.variant == PN_VARIANT_HOMO;
    const bool has_att = d.variant != PN_VARIANT_PAGG;
    const float *Xh = c.Xh;
    float *Z = c.Z;
    float *dZ = c.at<float>(c.w.dZ);
    float *dXh = a->Xh_in ? a->g_Xh : c.at<float>(c.w.dXh);
    float *dhn = c.at<float>(c.w.dhn), *dG = c.at<float>(c.w.dG);
    // every buffer the kernels below accumulate into (atomics / += / split-K) is cleared by ONE launch
    ZeroList zl{};
    auto zero = [&](float *ptr, size_t count) -> int {
        if (ptr && count) {
            if (zl.n >= 12) PN_FAIL(PN_ERR_ARG, "internal: zero list overflow");
            zl.ptr[zl.n] = ptr;
            zl.count[zl.n] = count;
            zl.n++;
        }
        return PN_OK;
    };
    auto flush_zero = [&](hipStream_t zs = nullptr) -> int {
        if (!zs) zs = stream;
        if (zl.n) {
            StageTimer tm(ctx, ST_ZERO_FILL, zs);       // (d Z and d Xh are the bulk: rows of the graph, not paths)
            hipLaunchKernelGGL(zero_kernel, dim3(512), dim3(256), 0, zs, zl);
            PN_CHECK_HIP(hipGetLastError());
            zl.n = 0;
        }
        return PN_OK;
    };
    float *scratch = c.at<float>(c.w.dl1);
    if (int rc = zero(dZ, (size_t)d.ZR * H)) return rc;
    if (int rc = zero(dXh, (size_t)d.N * H)) return rc;
    if (int rc = zero(a->g_att_w, has_att ? (size_t)2 * H : 0)) return rc;
    if (int rc = zero(a->g_att_b, has_att ? 1 : 0)) return rc;
    if (has_att && (!a->g_att_w || !a->g_att_b))
        if (int rc = zero(scratch, (size_t)2 * H + 1)) return rc;
    if (int rc = zero(a->g_fc2_b, (size_t)d.C)) return rc;
    if (int rc = zero(a->g_fc2_w, (size_t)d.C * 2 * H)) return rc;
    if (int rc = zero(a->g_bank_w, (size_t)L * H * H)) return rc;
    if (int rc = zero(a->g_bank_b, (size_t)L * H)) return rc;
    if (!a->Xh_in) {
        if (int rc = zero(a->g_fc0_w, (size_t)H * d.F)) return rc;
        if (int rc = zero(a->g_fc0_b, (size_t)H)) return rc;
    }
    if (d.S == 0) {
        if (int rc = flush_zero()) return rc;
        if (int rc = zero(a->g_w_ih, (size_t)GwH * H)) return rc;
        if (int rc = zero(a->g_w_hh, (size_t)GwH * H)) return rc;
        if (int rc = zero(a->g_b_ih, (size_t)GwH)) return rc;
        if (int rc = zero(a->g_b_hh, (size_t)GwH)) return rc;
        if (int rc = zero(a->g_X, (size_t)d.N * d.F)) return rc;
        if (int rc = flush_zero()) return rc;
        if (a->Xh_in && a->g_Xh_ready) PN_CHECK_HIP(hipEventRecord((hipEvent_t)a->g_Xh_ready, stream));    // (zeros: complete)
        return PN_OK;
    }
    // (the fused step clears them after the forward of its first micro-batch instead: the forward's gigabyte of saved
    //  tensors would push the freshly zeroed dZ out of the caches the scatter atomics want it in)
    if (!fused)
        if (int rc = flush_zero()) return rc;

    JoinGuard joiner{ctx, stream};
    if (fused) {
        // the zero fill rides on the second stream with the index plan (joined before the recurrence): off the chain
        // recurrence -> pooling -> BPTT, where it used to sit because the forward's gigabyte of saved tensors evicts the
        // freshly zeroed d Z -- 7 MB to fetch again, ~1 us of HBM time, against 5 us of launch in the chain
        const std::function<int(hipStream_t)> zero_side = [&](hipStream_t zs) { return flush_zero(zs); };
        if (int rc = run_tables(c, joiner, knobs_of(ctx).zero_early ? &zero_side : nullptr)) return rc;
    }
    // (per-stage timings are taken serially.  Deterministic mode: the classifier's weight gradient shares the chunk-sum
    //  buffer `dgemm` with the bank / fc0 backward and stays in line; the recurrent weight gradient -- its own partials,
    //  reduced in a fixed order -- goes to the second stream as in the default mode: streams do not reorder a kernel's sums)
    const bool overlap_ok = !profiling_every_stage(ctx);
    const bool side_ok = overlap_ok && !d.det;
    // pn_pagg_train_step, default mode: pooling forward + loss + pooling backward of a node as one launch (pool_step_kernel)
    const bool pool_step = fused && !d.det && knobs_of(ctx).pool_step != 0 && knobs_of(ctx).pool_bwd_wg != 0;
    const bool f16 = d.math == PN_SEQ_MATH_F16X2;
    const int seq4 = f16 ? 0 : seq4_select(ctx, H, G, L);
    SeqRange *range = c.at<SeqRange>(c.w.range);
    float *dgemm = d.det ? c.at<float>(c.w.dgemm) : nullptr;
    if (d.det) {        // the identity the BPTT kernels index the contribution buffer with
        const int64_t n = std::max<int64_t>((int64_t)d.Sb * d.W * L, d.Sb);
        hipLaunchKernelGGL(det_iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c.at<int32_t>(c.w.iota), n);
        PN_CHECK_HIP(hipGetLastError());
    }
    if (G > 0 && !d.generic) {
        StageTimer tm(ctx, ST_PLAN_PACK, stream);      // (its own bracket: ST_SEQ_BWD times the BPTT kernel alone)
        if (f16) {
            // (packed by the forward that left its saved tensors on this workspace -- run_pack_fwd; the fused step's
            //  run_tables above did the same)
        } else {
            if (int rc = launch_pack_bwd3(stream, a->w_ih, a->w_hh, H, G, d.cell == CELL_GRU ? 1 : 0, c.at<void>(c.w.WpT))) return rc;
        }
    }
    for (int b = 0; b < d.nb; b++) {
        const int Sb = c.groups(b);
        const int64_t Pb = (int64_t)Sb * d.W;
        const float *g_out = fused ? c.at<const float>(c.w.gout) : a->g_out + (size_t)b * d.Sb * d.C;
        if (fused || d.nb > 1) {
            // pn_pagg_backward with micro-batches: this micro-batch's recurrence again (same seed and positions in the
            // batch -> same dropout masks), now keeping what the BPTT needs.  pn_pagg_train_step: its one and only
            // forward, the logits go to the caller, their loss gradient to the workspace.
            if (!(fused && b == 0)) {       // (run_tables made the plan of micro-batch 0)
                StageTimer tm(ctx, ST_PLAN_PACK, stream);
                if (int rc = run_plan(c, stream, b)) return rc;
            }
            if (d.det)                      // the scatters' destination orders: second stream, under this forward
                if (int rc = fork_det_orders(c, joiner, b)) return rc;
            if (int rc = run_seq_fwd(c, b, true)) return rc;
            float *out_b = fused ? a->out + (size_t)b * d.Sb * d.C : c.at<float>(c.w.outb);
            if (!pool_step) {
                if (int rc = run_pool_fwd(c, b, out_b)) return rc;
                if (fused)
                    if (int rc = launch_cross_entropy(out_b, target + (size_t)b * d.Sb, Sb, d.C, grad_scale, loss,
                                                      c.at<float>(c.w.gout), stream, d.nb == 1))
                        return rc;
            }
            if (fused && b == 0)
                if (int rc = flush_zero()) return rc;
        }
        // classifier: g_fc2_w += g_out^T . layer1, g_fc2_b += colsum(g_out) -- nothing below reads them: second stream
        // (after the fused pooling step also the loss: the fixed-order sum of the per-node terms)
        // Where it runs: on the second stream, forked here -- or, when the recurrent weight gradient forks off behind the BPTT
        // anyway (defer_small), on `stream` right behind that fork together with the attention-weight reduction: every
        // fork / join is an event between two kernels of the queue and costs ~6 us of idle time there, and the BPTT waits
        // for none of the three (profiles/r06_glue.txt).
        const bool wgrad_wanted = G > 0 && (a->g_w_ih || a->g_w_hh || a->g_b_ih || a->g_b_hh);
        // (small_side, round 6: ONE fork behind the pooling backward instead -- the three small launches then run on the second
        //  stream while the BPTT runs, where defer_small put them in front of the node-level GEMMs, the longer of the two chains
        //  behind the BPTT.  Context knob PN_SMALL_SIDE, profiles/r06_glue.txt section 7)
        const bool small_side = side_ok && wgrad_wanted && ctx != nullptr && knobs_of(ctx).small_side != 0;
        const bool defer_small = side_ok && wgrad_wanted && ctx != nullptr && !small_side;
        hipStream_t small_stream = nullptr;     // set: where run_fc2_grad / att_reduce go (small_side)
        auto run_fc2_grad = [&]() -> int {
        hipStream_t cstream = small_stream ? small_stream : stream;
        if (side_ok && !defer_small && !small_stream)
            if (void *side = context_fork(ctx, stream)) cstream = (hipStream_t)side;
        if (pool_step) {
            hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(1024), 0, cstream, c.at<const float>(c.w.outb), Sb, grad_scale, loss,
                               d.nb == 1 ? 1 : 0);
            PN_CHECK_HIP(hipGetLastError());
        }
        {
            StageTimer tm(ctx, ST_FC2_GRAD, cstream);
            if (a->g_fc2_w && d.det) {
                if (int rc = launch_gemm_det(cstream, g_out, 1, d.C, nullptr, c.at<float>(c.w.layer1), 1, 2 * H, a->g_fc2_w,
                                             2 * H, d.C, 2 * H, Sb, (Sb + 127) / 128, a->g_fc2_b, dgemm))
                    return rc;
            } else if (a->g_fc2_w) {        // g_fc2_b = row sums of the A operand (g_out^T)
                if (int rc = launch_gemm(cstream, g_out, 1, d.C, nullptr, c.at<float>(c.w.layer1), 1, 2 * H, a->g_fc2_w,
                                         2 * H, nullptr, d.C, 2 * H, Sb, 0, GEMM_ATOMIC, (Sb + 127) / 128, a->g_fc2_b))
                    return rc;
            } else if (a->g_fc2_b) {
                if (int rc = launch_colsum(cstream, g_out, nullptr, d.C, Sb, d.C, a->g_fc2_b, d.det)) return rc;
            }
        }
        if (cstream != stream)
            if (int rc = joiner.mark()) return rc;
        return PN_OK;
        };
        if (!pool_step && !defer_small && !small_side)
            if (int rc = run_fc2_grad()) return rc;

        // pooling / attention backward -> dhn, dXh (ego rows), dZ or dXh (attention ego), g_att_*
        std::function<int()> att_reduce = [] { return PN_OK; };
        {
            PoolBwdParams pp{};
            pp.variant = d.variant;
            pp.S = Sb;
            pp.W = d.W;
            pp.H = H;
            pp.C = d.C;
            pp.N = d.N;
            pp.goff = c.group0(b);
            pp.hn = c.at<float>(c.w.hn);
            pp.ego_tab = homo ? Z : Xh;
            pp.egoidx = c.at<int32_t>(c.w.egoidx);
            pp.sel = c.sel(b);
            pp.att_w = a->att_w;
            pp.fc2_w = a->fc2_w;
            pp.g_out = g_out;
            pp.coef = c.at<float>(c.w.coef);
            pp.rawsc = c.at<float>(c.w.rawsc);
            pp.p_drop = a->p_cls;
            pp.seed = a->seed;
            pp.dyn = a->step_state;
            pp.mask = a->mask_cls;
            pp.dhn = dhn;
            pp.dXh = dXh;
            pp.dego = homo ? dZ : dXh;
            // attention gradients are optional outputs: fall back to scratch so the kernel needs no branches
            pp.g_att_w = a->g_att_w ? a->g_att_w : scratch;
            pp.g_att_b = a->g_att_b ? a->g_att_b : scratch + 2 * H;
            // the attention-weight terms always go through per-workgroup partials + an ordered reduce: 325 workgroups adding
            // to the same 2H + 1 addresses with atomics serialise
            pp.det_att = c.at<float>(c.w.datt);
            if (d.det) {
                pp.det_sel = c.at<float>(c.w.dsel);
                pp.det_ds = c.at<float>(c.w.dds);
            }
            const size_t lds_bytes = (size_t)(4 * (2 * d.W + H) + 8 * H + 8 * d.W) * sizeof(float);
            // a workgroup per group (four waves share its members) unless PN_POOL_BWD_WG=0; it needs the partials buffer
            const bool wg = (!has_att || pp.det_att) && knobs_of(ctx).pool_bwd_wg != 0;
            int att_blocks = (Sb + 3) / 4;
            if (pool_step) {        // pooling forward, loss and this backward in one launch (pool_step_kernel)
                if (!wg) PN_FAIL(PN_ERR_ARG, "internal: the fused pooling step needs the workgroup-per-node backward");
                att_blocks = Sb;
                PoolStepParams ps{};
                ps.f = pool_fwd_params(c, b, a->out + (size_t)b * d.Sb * d.C);
                ps.b = pp;
                ps.b.g_out = c.at<float>(c.w.gout);
                ps.target = target + (size_t)b * d.Sb;
                ps.scale = grad_scale;
                ps.gout = c.at<float>(c.w.gout);
                ps.lossg = c.at<float>(c.w.outb);       // (the fused step's logits go to the caller: the slot is free)
                const size_t lds_step = std::max(lds_bytes, pool_fwd_lds_bytes(d));
                {
                    StageTimer tm(ctx, ST_POOL_FWD, stream);
                    if (d.W <= 40 && H <= 128 && H >= 32 && knobs_of(ctx).pool_step >= 1 && knobs_of(ctx).pool_step != 2) {
                        // the node's rows in registers (pool_step2_kernel); PN_POOL_STEP=2: the three bodies back to back
                        hipLaunchKernelGGL((pool_step2_kernel<10, 2>), dim3(Sb), dim3(256), pool_step2_lds_bytes(10, H, d.C), stream, ps);
#ifdef PN_POOL_TRACE_TWICE      // (timing experiment only: the launch again, its rows now read a second time -- results are wrong)
                        hipLaunchKernelGGL((pool_step2_kernel<10, 2>), dim3(Sb), dim3(256), pool_step2_lds_bytes(10, H, d.C), stream, ps);
#endif
                    } else if (H <= 256) {
                        hipLaunchKernelGGL(pool_step_kernel<4>, dim3(Sb), dim3(256), lds_step, stream, ps);
                    } else {
                        if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(pool_step_kernel<16>), (int)lds_step)) return rc;
                        hipLaunchKernelGGL(pool_step_kernel<16>, dim3(Sb), dim3(256), lds_step, stream, ps);
                    }
                    PN_CHECK_HIP(hipGetLastError());
                }
                if (!defer_small && !small_side)
                    if (int rc = run_fc2_grad()) return rc;
            }
            {
            StageTimer tm(ctx, ST_POOL_BWD, stream);
            if (pool_step) {
            } else if (wg) {
                att_blocks = Sb;
                if (H <= 256) {
                    hipLaunchKernelGGL(pool_bwd_wg_kernel<4>, dim3(Sb), dim3(256), lds_bytes, stream, pp);
                } else {
                    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(pool_bwd_wg_kernel<16>), (int)lds_bytes)) return rc;
                    hipLaunchKernelGGL(pool_bwd_wg_kernel<16>, dim3(Sb), dim3(256), lds_bytes, stream, pp);
                }
            } else if (H <= 256) {
                hipLaunchKernelGGL(pool_bwd_kernel<4>, dim3((Sb + 3) / 4), dim3(256), lds_bytes, stream, pp);
            } else {
                if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(pool_bwd_kernel<16>), (int)lds_bytes)) return rc;
                hipLaunchKernelGGL(pool_bwd_kernel<16>, dim3((Sb + 3) / 4), dim3(256), lds_bytes, stream, pp);
            }
            PN_CHECK_HIP(hipGetLastError());
            }
            att_reduce = [=, &small_stream]() -> int {        // the attention weights' terms in workgroup order
                if (has_att && pp.det_att) {
                    hipLaunchKernelGGL(det_att_reduce_kernel, dim3((2 * H + 4 + 63) / 64), dim3(1024), 0,
                                       small_stream ? small_stream : stream, pp.det_att, att_blocks, H, pp.g_att_w, pp.g_att_b);
                    PN_CHECK_HIP(hipGetLastError());
                }
                return PN_OK;
            };
            if (small_side) {       // behind the pooling backward: loss sum, classifier gradient, attention reduce -> second stream
                if (void *side = context_fork(ctx, stream)) small_stream = (hipStream_t)side;
                if (int rc = run_fc2_grad()) return rc;
                if (int rc = att_reduce()) return rc;
                if (small_stream)
                    if (int rc = joiner.mark()) return rc;
                small_stream = nullptr;
            } else if (!defer_small)
                if (int rc = att_reduce()) return rc;
            if (d.det) {
                // (this micro-batch's orders were sorted on the second stream under its forward, when the backward re-ran it)
                if (fused || d.nb > 1)
                    if (int rc = joiner.join()) return rc;
                // the ego half of d layer1 onto the masked nodes' rows of dXh; the attention-ego term onto the ego rows
                // (one launch per pass for the two: they write different rows -- or, for the hetero / PAGG classes whose ego rows
                //  live in dXh as well, the same rows in a fixed order: SEL first would need a launch boundary, see below)
                const DetScatterParams two[2] = {det_params(c, DET_SEL, b, pp.det_sel, nullptr, nullptr, dXh),
                                                 det_params(c, DET_EGO, b, nullptr, pp.det_ds, a->att_w + H, pp.dego)};
                if (has_att && pp.dego != dXh) {
                    if (int rc = run_det_scatters(c, two, 2)) return rc;
                } else {        // both add into dXh: one after the other
                    if (int rc = run_det_scatters(c, two, 1)) return rc;
                    if (has_att)
                        if (int rc = run_det_scatters(c, two + 1, 1)) return rc;
                }
            }
        }

        // recurrent weight / bias gradients: [g_W_ih | g_W_hh] (+)= dG^T . XH, g_b (+)= colsum(dG) over the Pb * L rows of this
        // micro-batch: K-split partials into `wpart`, then their reduction.  Second stream, forked off `stream` where it is called.
        auto run_wgrad = [&]() -> int {
            hipStream_t wstream = stream;
            if (overlap_ok)
                if (void *side = context_fork(ctx, stream)) wstream = (hipStream_t)side;
            WgradParams wp{};
            wp.dG = dG;
            wp.xh = c.at<const float>(c.w.xh);
            wp.R = Pb * L;
            wp.GH = GH;
            wp.H2 = 2 * H;
            int nz = c.w.wgrad_split;
            // a node-sharded caller runs its reduce-scatter of d Xh (RCCL's kernels) and fc0's backward under this launch
            // (g_Xh_ready): they need CUs whose registers are not all taken, whatever the launch's length
            if (a->Xh_in && a->g_Xh_ready) nz = std::min(nz, std::max(1, (WGRAD_CUS_SHARED + c.w.wgrad_tiles - 1) / c.w.wgrad_tiles));
            int64_t rps = (wp.R + nz - 1) / nz;
            rps = (rps + WG_KT - 1) / WG_KT * WG_KT;
            wp.rows_per_split = rps;
            const int64_t ntiles = (wp.R + WG_KT - 1) / WG_KT;
            const int nz_used = (int)(ntiles < nz ? ntiles : nz);
            wp.part_w = c.at<float>(c.w.wpart);
            wp.part_b = wp.part_w + (size_t)nz * GH * 2 * H;
            {
                StageTimer tm(ctx, ST_WGRAD, wstream);
                int nz_red = nz_used;
                wp.range = range;
                wp.xmul = seq_xmul(a);
                if (f16 && !d.generic) {        // the two-stage pipeline on fp16 planes (pn_seqh.hip)
                    const int64_t nt16 = (wp.R + 15) / 16;
                    nz_red = (int)(nt16 < nz ? nt16 : nz);
                    if (int rc = launch_wgradh(ctx, wstream, wp, H, nz_red)) return rc;
                } else if (seq4 & SEQ4_WGRAD) {        // two-stage pipeline over K tiles of 16 rows (pn_seq4.hip)
                    const int64_t nt16 = (wp.R + 15) / 16;
                    nz_red = (int)(nt16 < nz ? nt16 : nz);
                    if (int rc = launch_wgrad4(ctx, wstream, wp, nz_red)) return rc;
                } else {
                    if (int rc = launch_wgrad3(ctx, wstream, wp, nz_used)) return rc;
                }
                if (int rc = launch_wgrad_reduce(wstream, wp.part_w, wp.part_b, nz_red, GH, H, b > 0 ? 1 : 0, d.cell == CELL_GRU ? 1 : 0,
                                                 a->g_w_ih, a->g_w_hh, a->g_b_ih, a->g_b_hh))
                    return rc;
            }
            if (wstream != stream)
                if (int rc = joiner.mark()) return rc;   // `stream` waits for the weight gradients on the way out
            return PN_OK;
        };

        // BPTT + gather-backward scatter (mean / sum encoders: the scatter alone)
        // (the fp16 BPTT stores its rows outright; the other kernels add into a zero-filled buffer)
        if (d.det && !(f16 && G > 0 && !d.generic))
            PN_CHECK_HIP(hipMemsetAsync(c.at<float>(c.w.dx), 0, (size_t)Pb * L * H * sizeof(float), stream));
        if (G == 0) {
            if (int rc = run_seq_reduce(c, b, true)) return rc;
        } else if (d.generic) {
            if (int rc = run_seq_bwd_generic(c, b)) return rc;
        } else {
            StageTimer tm(ctx, ST_SEQ_BWD, stream);
            SeqBwdParams sp{};
            sp.saved = c.at<float>(c.w.saved);
            sp.keep = (a->mask_seq || !(a->p_seq > 0.0f)) ? nullptr : c.at<const uint8_t>(c.w.keep);
            sp.dhn = dhn;
            sp.rowidx = c.at<int32_t>(c.w.rowidx);
            sp.slotof = c.at<int32_t>(c.w.slotof);
            sp.WpT = c.at<float>(c.w.WpT);
            sp.dG = dG;
            sp.dZ = dZ;
            sp.P = (int)Pb;
            sp.L = L;
            sp.Pmask = d.P_total;
            sp.p_drop = a->p_seq;
            sp.seed = a->seed;
            sp.mask = a->mask_seq;
            sp.merge0 = d.variant != PN_VARIANT_HETERO && !d.det;   // (the hetero plan's step-0 rows are other paths' far ends: no runs)
            if (d.det) sp.rowidx = c.at<int32_t>(c.w.iota), sp.dZ = c.at<float>(c.w.dx);     // (see det_scatter_kernel)
            sp.range = range;
            sp.store_dx = d.det && f16;
            if (f16) {
                const int gc = d.cell == CELL_GRU ? 3 : d.cell == CELL_LSTM ? 4 : 1;
                if (int rc = launch_seq_bwdh(ctx, stream, H, gc, sp)) return rc;
            } else if (int rc = launch_seq_bwd3(ctx, stream, H, d.cell == CELL_GRU ? 3 : d.cell == CELL_LSTM ? 4 : 1, sp))
                return rc;
        }
        if (d.det) {        // the stored mask * dx rows onto their table rows, in the order of the path steps
            StageTimer tm(ctx, ST_SEQ_BWD, stream);
            if (int rc = run_det_scatter(c, DET_ROW, b, c.at<float>(c.w.dx), nullptr, nullptr, dZ)) return rc;
        }

        if (wgrad_wanted)
            if (int rc = run_wgrad()) return rc;
        if (defer_small) {      // behind the fork: beside the weight-gradient GEMM, ahead of the node-level GEMMs
            if (int rc = run_fc2_grad()) return rc;
            if (int rc = att_reduce()) return rc;
        }
        // the next micro-batch rewrites the [x|h] rows and dG the weight-gradient GEMM is reading
        if (b + 1 < d.nb)
            if (int rc = joiner.join()) return rc;
    }
    // distance bank backward (ReLU gate for HOMO): dXh += dZ' . bank_w ; g_bank_w = dZ'^T . Xh
    const float *zgate = homo ? Z : nullptr;
    auto tm_bank = std::make_unique<StageTimer>(ctx, ST_BANK_BWD, stream);
    if (d.compact) {
        // over the touched rows, code by code (the compact rows of a code are distinct nodes: += without atomics):
        //   dXh[node] += dZ'[row] . bank_w[code],   g_bank_w[code] += dZ'[rows]^T . Xh[nodes],   g_bank_b[code] += colsum
        if (a->g_bank_b && !a->g_bank_w) PN_FAIL(PN_ERR_ARG, "pn_pagg_backward: compact rows need g_bank_w with g_bank_b");
        const int mmax = (int)std::min<int64_t>(d.N, d.ZR);
        const int32_t *seg = c.at<const int32_t>(c.w.seg), *list = c.at<const int32_t>(c.w.list);
        // (measured at the 10 M-node shape: the 128 x 128 tiles' read-modify-write of scattered dXh rows at two workgroups
        //  per CU is slower than the 64 x 64 kernel's, 24.1 vs 21.6 ms for the stage; PN_NODE_GEMM3 bit 3 switches it on)
        const bool g3 = (knobs_of(ctx).node_gemm3 & 8) && gemm3_pays(ctx, mmax, H, H, G3_BANK_DX);
        float *bankT = c.at<float>(c.w.bankT);
        if (g3)             // bank_w[code] [out, in] -> [in, out]: the reduction index (out) contiguous
            for (int code = 0; code < L; code++)
                launch_transpose(stream, a->bank_w + (size_t)code * H * H, H, H, bankT + (size_t)code * H * H);
        // first every code's contribution to d Xh (what a sharded caller's reduce-scatter waits for), then the weight gradients
        for (int code = 0; code < L; code++) {
            if (g3) {
                if (int rc = launch_gemm3(stream, dZ, H, bankT + (size_t)code * H * H, H, dXh, H, nullptr, mmax, H, H, 0, 1, zgate,
                                          GEMM_IND_C_ROWS, seg + code, list))
                    return rc;
            } else if (int rc = launch_gemm(stream, dZ, H, 1, zgate, a->bank_w + (size_t)code * H * H, 1, H, dXh, H, nullptr, mmax,
                                            H, H, 0, GEMM_ADD, 1, nullptr, GEMM_IND_C_ROWS, seg + code, list))
                return rc;
        }
        if (a->Xh_in && a->g_Xh_ready) PN_CHECK_HIP(hipEventRecord((hipEvent_t)a->g_Xh_ready, stream));
        for (int code = 0; code < L; code++) {
            if (a->g_bank_w && d.det) {
                if (int rc = launch_gemm_det(stream, dZ, 1, H, zgate, Xh, 1, H, a->g_bank_w + (size_t)code * H * H, H, H, H, mmax,
                                             (mmax + 255) / 256, a->g_bank_b ? a->g_bank_b + (size_t)code * H : nullptr, dgemm,
                                             GEMM_IND_K, seg + code, list))
                    return rc;
            } else if (a->g_bank_w && rgrad_pays(ctx, mmax, H, H)) {         // large graphs: the bf16 x 3 row-reduction kernel
                const RgradParams rp{dZ, zgate, Xh, H, H, mmax, H, H, a->g_bank_w + (size_t)code * H * H, H,
                                     a->g_bank_b ? a->g_bank_b + (size_t)code * H : nullptr, seg + code, list};
                if (int rc = launch_rgrad(ctx, stream, rp)) return rc;
            } else if (a->g_bank_w)
                if (int rc = launch_gemm(stream, dZ, 1, H, zgate, Xh, 1, H, a->g_bank_w + (size_t)code * H * H, H, nullptr, H,
                                         H, mmax, 0, GEMM_ATOMIC, (mmax + 255) / 256,
                                         a->g_bank_b ? a->g_bank_b + (size_t)code * H : nullptr, GEMM_IND_K, seg + code, list))
                    return rc;
        }
    } else {
    if (gemm3_pays(c.ctx, d.N, H, L * H, G3_BANK_DX)) {        // bank_w [L*H, H] -> [H, L*H]
        float *bankT = c.at<float>(c.w.bankT);
        launch_transpose(stream, a->bank_w, L * H, H, bankT);
        if (int rc = launch_gemm3(stream, dZ, (int64_t)L * H, bankT, (int64_t)L * H, dXh, H, nullptr, d.N, H, L * H, 0, 1, zgate))
            return rc;
    } else if (int rc = launch_gemm_split(stream, dZ, (int64_t)L * H, 1, zgate, a->bank_w, 1, H, dXh, H, nullptr, d.N, H, L * H,
                                          0, GEMM_ADD, c.at<float>(c.w.gpart)))
        return rc;
    if (a->Xh_in && a->g_Xh_ready) PN_CHECK_HIP(hipEventRecord((hipEvent_t)a->g_Xh_ready, stream));
    if (a->g_bank_w && d.det) {
        if (int rc = launch_gemm_det(stream, dZ, 1, (int64_t)L * H, zgate, Xh, 1, H, a->g_bank_w, H, L * H, H, d.N,
                                     (d.N + 255) / 256, a->g_bank_b, dgemm))
            return rc;
    } else if (a->g_bank_w && rgrad_pays(ctx, d.N, L * H, H)) {
        const RgradParams rp{dZ, zgate, Xh, (int64_t)L * H, H, d.N, L * H, H, a->g_bank_w, H, a->g_bank_b, nullptr, nullptr};
        if (int rc = launch_rgrad(ctx, stream, rp)) return rc;
    } else if (a->g_bank_w) {       // g_bank_b rides along as the row sums of the same (gated) A operand
        if (int rc = launch_gemm(stream, dZ, 1, (int64_t)L * H, zgate, Xh, 1, H, a->g_bank_w, H, nullptr, L * H, H, d.N,
                                 0, GEMM_ATOMIC, (d.N + 255) / 256, a->g_bank_b))
            return rc;
    } else if (a->g_bank_b) {
        if (int rc = launch_colsum(stream, dZ, zgate, (int64_t)L * H, d.N, L * H, a->g_bank_b, d.det)) return rc;
    }
    }

    tm_bank.reset();
    if (a->Xh_in) return PN_OK;   // the caller finishes fc0 after the reduce-scatter of g_Xh
    // fc0 backward (ReLU gate for HOMO)
    const float *xgate = homo ? Xh : nullptr;
    StageTimer tm_fc0(ctx, ST_FC0_BWD, stream);
    if (a->g_fc0_w && d.det) {
        if (int rc = launch_gemm_det(stream, dXh, 1, H, xgate, a->X, 1, d.F, a->g_fc0_w, d.F, H, d.F, d.N, (d.N + 255) / 256,
                                     a->g_fc0_b, dgemm))
            return rc;
    } else if (a->g_fc0_w && d.F % 4 == 0 && rgrad_pays(ctx, d.N, H, d.F)) {
        const RgradParams rp{dXh, xgate, a->X, H, d.F, d.N, H, d.F, a->g_fc0_w, d.F, a->g_fc0_b, nullptr, nullptr};
        if (int rc = launch_rgrad(ctx, stream, rp)) return rc;
    } else if (a->g_fc0_w && xpad_floats(d) > 0 && rgrad_pays(ctx, d.N, H, (d.F + 3) / 4 * 4)) {
        // rows of X padded to 16 bytes first (see xpad_floats); the padding columns are zeros, and outputs past F are not stored
        const int F4 = (d.F + 3) / 4 * 4;
        float *Xp = c.at<float>(c.w.xpad);
        const int64_t n4 = (int64_t)d.N * (F4 / 4);
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a->X, (int64_t)d.N, d.F, F4, Xp);
        PN_CHECK_HIP(hipGetLastError());
        const RgradParams rp{dXh, xgate, Xp, H, F4, d.N, H, d.F, a->g_fc0_w, d.F, a->g_fc0_b, nullptr, nullptr};
        if (int rc = launch_rgrad(ctx, stream, rp)) return rc;
    } else if (a->g_fc0_w) {
        if (int rc = launch_gemm(stream, dXh, 1, H, xgate, a->X, 1, d.F, a->g_fc0_w, d.F, nullptr, H, d.F, d.N, 0,
                                 GEMM_ATOMIC, (d.N + 255) / 256, a->g_fc0_b))
            return rc;
    } else if (a->g_fc0_b) {
        if (int rc = launch_colsum(stream, dXh, xgate, H, d.N, H, a->g_fc0_b, d.det)) return rc;
    }
    if (a->g_X)
        if (int rc = launch_gemm(stream, dXh, H, 1, xgate, a->fc0_w, 1, d.F, a->g_X, d.F, nullptr, d.N, d.F, H, 0,
                                 GEMM_STORE, 1))
            return rc;
    return PN_OK;
}

int pn_pagg_backward(pn_context *ctx, const pn_pagg_args *a, void *stream) {
    return pagg_backward_impl(ctx, a, stream, nullptr, 0.0f, nullptr);
}

int pn_pagg_train_step(pn_context *ctx, const pn_pagg_args *a, const int64_t *target, float grad_scale, float *loss,
                       void *stream) {
    if (!a) PN_FAIL(PN_ERR_ARG, "pn_pagg_train_step: null args");
    if (!loss || (a->shape.S > 0 && !target)) PN_FAIL(PN_ERR_ARG, "pn_pagg_train_step: null target / loss");
    if (a->shape.S == 0) {          // nothing to aggregate: zero loss, zero gradients
        PN_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), (hipStream_t)stream));
        return pagg_backward_impl(ctx, a, stream, nullptr, 0.0f, nullptr);
    }
    return pagg_backward_impl(ctx, a, stream, target, grad_scale, loss);
}

}  // extern "C"

#ifdef PN_POOL_TRACE
extern "C" int pn_debug_pool_trace(long long *host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pool_trace), sizeof(long long) * 2048 * 12) == hipSuccess ? 0 : -1;
}
#endif
