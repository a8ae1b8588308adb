// pn_seq3.hip -- the recurrent kernels of the aggregator on the bf16 matrix pipe (pn_pagg_shape.seq_math = bf16x3; rounds 1-3's
// default, selectable since round 4): fp32 products as six bf16 MFMAs over exact three-plane splits (pn_kernels.h), every hidden
// size that is a multiple of 32 up to 256, LSTM / GRU (on the LSTM's four gate slots) / tanh RNN.
//   seq_fwd3_kernel   nn.LSTM / nn.RNN forward fused with the path gather, dropout and the saved tensors
//                     (/root/reference/PathNet_run.py:164,179-195 / :228,246-265; baseline/GPRGNN/src/copy.py:308,334-349)
//   seq_bwd3_kernel   BPTT + the gather backward's atomic scatter        (autograd through the above, :348-351)
//   wgrad3_kernel     [g_W_ih | g_W_hh] = dG^T . [x | h], K-split partials;  wgrad_reduce_kernel adds the partials of either
//                     arithmetic in split order and hands out the caller's gate layout
// Split out of pn_pagg.hip in round 6; the kernels are unchanged.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "pn_internal.h"
#include "pn_kernels.h"
#include "pn_seq.h"

using namespace pn;

// tuning knobs (variant builds with -D...)
#ifndef PN_SEQ_RG
#define PN_SEQ_RG 1         // row groups (of 32 sequence slots) per workgroup of the recurrent kernels (2: measured slower)
#endif
#ifndef PN_FWD_WAVES
#define PN_FWD_WAVES 3      // __launch_bounds__ min waves per SIMD = workgroups per CU, forward (3: 168 registers)
#endif
#ifndef PN_BWD_WAVES
#define PN_BWD_WAVES 2      // backward: 2 (a 168-register build re-reads the (g, o) gradients and A fragments: slower)
#endif
#ifndef PN_BWD_NB
#define PN_BWD_NB 16        // cell backward: accumulator elements per batch of saved-gate loads (16: one memory round
                            // trip per step; 8: two, measured 2 % slower)
#endif

namespace {

// ================================================================================================
// The same recurrence on the bf16 matrix pipe (pn_kernels.h: fp32 = three bf16 planes, six MFMAs per product).
//   Weights: pack_fwd3_kernel splits [W_ih | W_hh] once per forward into B fragments of v_mfma_f32_32x32x16_bf16,
//     Wp3[(((w*KS + s)*3 + plane)*G + g)*64 + lane] (16 bytes) =
//         plane of Wcat[g*H + 32w + (lane & 31)][16 s + 8 (lane >> 5) .. +7],        KS = 2H/16 k-steps,
//     so the 3*G loads of one k-step are 3*G consecutive KB; step 0 (h_{-1} = 0) simply stops after the x half.
//   A operand: LDS holds the three planes of the tile [32][x_t | h_{t-1}] as bf16, row pitch 4H + 16 bytes
//     (conflict-free ds_read_b128).  x is split when the gathered rows are committed, h in the cell update.
// ================================================================================================
// GRU (torch gate order r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn))) rides on the four gate slots of the
// LSTM kernels: slot 0 = r, 1 = z over [x | h] as usual, slot 2 = "nx" = W_in x + b_in (its h half of the weights is
// zero), slot 3 = "nh" = W_hn h + b_hn (its x half is zero).  A quarter of the products multiplies zeros; in return
// the recurrence, its BPTT and the weight-gradient GEMM are the LSTM's kernels with another cell function.

__global__ void pack_fwd3_kernel(const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                 const float *__restrict__ b_ih, const float *__restrict__ b_hh, int H, int G, int gru,
                                 u32x4 *__restrict__ Wp, float *__restrict__ biasc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
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
    u32x4 q0, q1, q2;
    uint32_t x0, x1, x2;
    split3(v0.x, v0.y, x0, x1, x2); q0[0] = x0; q1[0] = x1; q2[0] = x2;
    split3(v0.z, v0.w, x0, x1, x2); q0[1] = x0; q1[1] = x1; q2[1] = x2;
    split3(v1.x, v1.y, x0, x1, x2); q0[2] = x0; q1[2] = x1; q2[2] = x2;
    split3(v1.z, v1.w, x0, x1, x2); q0[3] = x0; q1[3] = x1; q2[3] = x2;
    u32x4 *dst = Wp + ((int64_t)(w * KS + s) * 3 * G + g) * 64 + lane;
    dst[0] = q0;
    dst[G * 64] = q1;
    dst[2 * G * 64] = q2;
}

// RG row groups of 32 sequence slots per workgroup.  The waves (rg, w) of the RG groups walk the same weight stream
// in step (they meet at every barrier), so the fragments one of them pulls from L2 are L1 hits for the others: the
// L2 -> CU weight traffic, which bounds this kernel (4.4 GB per launch at RG = 1 on the bench workload, ~15 TB/s),
// drops by the factor RG.
// waves per SIMD the forward kernel is compiled for: H = 256 fills the LDS with one workgroup of 8 waves, H = 32 is
// a single wave per workgroup (no register cap: a spill next to the asm loads would be a hazard)
template <int H, int RG>
constexpr int fwd_waves() { return H == 32 && RG == 1 ? 1 : (H > 128 || RG > 1) ? 2 : PN_FWD_WAVES; }

// GC: 4 = LSTM, 1 = tanh RNN, 3 = GRU (on the LSTM's four gate slots, see pack_fwd3_kernel)
// RB: row blocks of 32 paths per wave.  RB = 2: a wave keeps two accumulator sets and uses every weight fragment twice --
// the fragment stream per path, which bounds the kernel, halves; the 64-row tile takes 101 KB of LDS, so
// one workgroup per CU, one wave per SIMD with the whole register file.
template <int H, int GC, int RG, int RB = 1>
__global__ __launch_bounds__(H / 32 * 64 * RG, (RB > 1 ? 1 : fwd_waves<H, RG>())) void seq_fwd3_kernel(SeqFwdParams p) {
    constexpr int G = GC == 3 ? 4 : GC;
    constexpr bool GRU = GC == 3;
    constexpr int MT = 32 * RG * RB;
    constexpr int NW = H / 32, NT = NW * 64 * RG, SV = (G == 4 ? 5 : 1);
    constexpr int KS = H / 8, KX = KS / 2;    // k-steps of 16 over [x | h]; the first KX walk x
    constexpr int PB = 4 * H + 16;            // row pitch of a plane of the tile [x | h], bytes: conflict-free ds_read_b128
    constexpr int PLANE = MT * PB;
    // three workgroups per CU (168 registers): no register room for the x_{t+1} rows or a second plane-0 fragment set,
    // the third workgroup covers those latencies instead
    constexpr bool PREFETCH_X = RB > 1 || fwd_waves<H, RG>() < 3, PING_PONG = RB > 1 || fwd_waves<H, RG>() < 3;
    // LDS: three bf16 planes of the tile [MT][x_t | h_{t-1}] | row indices
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    int *s_rowidx = reinterpret_cast<int *>(ldsb + 3 * PLANE);  // [MT][L] gather rows of this tile
    int *s_slotof = s_rowidx + MT * p.L;                        // [MT]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31;
    const int ws = wave % NW, r0 = 32 * RB * (wave / NW);       // column slice / first tile row of this wave
    const int q0 = blockIdx.x * MT;
    const int col = 32 * ws + li;
    const int ws_u = __builtin_amdgcn_readfirstlane(ws);        // the wave's column slice as a scalar (weight stream base)

    for (int i = tid; i < MT * p.L; i += NT) s_rowidx[i] = q0 + i / p.L < p.P ? p.rowidx[(int64_t)q0 * p.L + i] : 0;
    for (int i = tid; i < MT; i += NT) s_slotof[i] = q0 + i < p.P ? p.slotof[q0 + i] : 0;

    f32x16 cst[RB];
#pragma unroll
    for (int rb = 0; rb < RB; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) cst[rb][r] = 0.0f;
    const float keep_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (1.0f - p.p_drop))));
    const bool builtin_drop = !p.mask && p.p_drop > 0.0f;
    const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
    // per-path tensors are addressed as a 64-bit tile base (wave-uniform: SGPRs) + a 32-bit offset inside the tile,
    // so no tensor size is bounded by 2^32 elements
    const size_t tile_row = (size_t)q0 * (size_t)p.L;                       // first [P, L] row of this tile
    uint8_t *keep_t = p.keep ? p.keep + tile_row * (H / 4) : nullptr;
    float4 *xh4_t = p.xh ? reinterpret_cast<float4 *>(p.xh) + tile_row * (2 * H / 4) : nullptr;
    float *xh_t = p.xh ? p.xh + tile_row * (2 * H) : nullptr;
    float *saved_t = p.saved ? p.saved + tile_row * (SV * H) : nullptr;
    float *hn_t = p.hn + (size_t)q0 * H;
    __syncthreads();

    // ---- coalesced row gather of x_{t+1} (H*4 bytes per row).  With PREFETCH_X (two workgroups per CU) the loads
    //      are issued before step t's k loop and stay in flight under it -- asm loads, hipcc would sink ordinary
    //      ones to their use -- with the dropout keep bits drawn right behind them; with three workgroups per CU the
    //      rows are fetched in the cell-update phase instead (the other workgroups cover the latency).  Either
    //      way the mask is applied when the rows are committed to LDS.
    constexpr int NLD = 4 * RB;   // float4 per thread = MT * (H/4) / NT
    f32x4 xr[NLD];
    uint32_t keepbits = 0;        // 4 bits per row of this thread
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int tid_g = tid;              // re-derived per step (fresh_lane): the per-row offsets derived from it would
                                  // otherwise live (and spill) across the MFMA loop as loop invariants
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
            const int q = q0 + row;
            float4 v = make_float4(xr[i][0], xr[i][1], xr[i][2], xr[i][3]);
            if (q >= p.P) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.mask) {
                if (q < p.P) {
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
                if (keep_t && q < p.P) keep_t[((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)(H / 4) + c4] = (uint8_t)(b & 15u);
            }
            uint32_t a0, a1, a2, b0, b1, b2;
            split3(v.x, v.y, a0, a1, a2);
            split3(v.z, v.w, b0, b1, b2);
            unsigned char *d = ldsb + row * PB + 8 * c4;
            *reinterpret_cast<uint2 *>(d) = make_uint2(a0, b0);
            *reinterpret_cast<uint2 *>(d + PLANE) = make_uint2(a1, b1);
            *reinterpret_cast<uint2 *>(d + 2 * PLANE) = make_uint2(a2, b2);
            if (xh4_t && q < p.P) {
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
        tid_g = wave_u * 64 + fresh_lane();
        if (PREFETCH_X && t + 1 < p.L) gather_issue(t + 1);

        f32x16 acc[RB][G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float bias = p.biasc[g * H + col];      // (re-read per step: G registers less across the kernel)
#pragma unroll
            for (int rb = 0; rb < RB; rb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[rb][g][r] = bias;
        }

        // ---- [x_t ; h_{t-1}] x [W_ih ; W_hh]^T.  Weight fragments stream L2 -> VGPR ahead of their MFMAs: the
        //      plane-0 fragments (needed first) ping-pong between two register sets one k-step ahead, planes 1 and 2
        //      are re-fetched into their own registers as soon as the MFMAs that read them are issued (2/3 of a k-step
        //      ahead).  vmcnt is in order: [P0(s) P1(s) P2(s) P0(s+1)] in flight at the top of k-step s.
        //      Step 0 has h_{-1} = 0: it stops after the x half of K.
        {
            const int nsteps = t == 0 ? KX : KS;
            // wave-uniform stream base in SGPRs, one VGPR of lane offset (pn_kernels.h: async_load_frags)
            const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.Wp) + (size_t)ws_u * (KS * 3 * G * 1024);
            const int lane_k = fresh_lane();
            const uint32_t voff = lane_k * 16;
            const unsigned char *arow = ldsb + (r0 + (lane_k & 31)) * PB + 16 * (lane_k >> 5);
            u32x4 P0a[G], P0b[G], P1[G], P2[G];
            auto load = [&](u32x4 (&B)[G], int s, int pl) {
                async_load_frags<G>(B, wb + (size_t)(s * 3 + pl) * (G * 1024), voff);
            };
            // PING_PONG: vmcnt (in order) sees [P0(s) P1(s) P2(s) P0(s+1)] at the top of k-step s;
            // otherwise one register set per plane, [P0(s) P1(s) P2(s)]
            // A fragments are software-pipelined without extra registers: within a k-step the products run
            // a2.P0, a1.P0, a0.P0 | a1.P1, a0.P1 | a0.P2, so plane 2 of the A tile is dead after the first G MFMAs, plane
            // 1 after the P1 group, plane 0 at the end -- each is re-read for k-step s+1 right there, and the next
            // k-step again starts with a2 (read longest ago) and needs a0 (read last) only after 2G MFMAs.
            u32x4 a[RB][3];
            auto aread = [&](int rb, int s, int pl) {
                return *reinterpret_cast<const u32x4 *>(arow + rb * 32 * PB + 32 * s + pl * PLANE);
            };
            // all RB row blocks of a product before the next product: every weight fragment feeds RB MFMAs
            auto prod = [&](int pa, u32x4 (&B)[G]) {
#pragma unroll
                for (int rb = 0; rb < RB; rb++)
#pragma unroll
                    for (int g = 0; g < G; g++) acc[rb][g] = mfma_bf16(a[rb][pa], B[g], acc[rb][g]);
            };
            auto areads = [&](int s, int pl) {
#pragma unroll
                for (int rb = 0; rb < RB; rb++) a[rb][pl] = aread(rb, s, pl);
            };
            auto kstep = [&](int s, u32x4 (&P0)[G], u32x4 (&P0next)[G]) {
                const int sn = min(s + 1, nsteps - 1);
                if (PING_PONG) load(P0next, sn, 0);
                wait_frag<(PING_PONG ? 3 : 2) * G, G>(P0);
                prod(2, P0);
                areads(sn, 2);
                prod(1, P0);
                prod(0, P0);
                if (!PING_PONG) load(P0, sn, 0);
                wait_frag<2 * G, G>(P1);
                prod(1, P1);
                areads(sn, 1);
                prod(0, P1);
                load(P1, sn, 1);
                wait_frag<2 * G, G>(P2);
                prod(0, P2);
                areads(sn, 0);
                load(P2, sn, 2);
            };
#pragma unroll
            for (int pl = 0; pl < 3; pl++) areads(0, pl);
            {
                load(P0a, 0, 0);
                load(P1, 0, 1);
                load(P2, 0, 2);
#pragma unroll 1
                for (int s = 0; s < nsteps; s += 2) {
                    if (PING_PONG) {
                        kstep(s, P0a, P0b);
                        kstep(s + 1, P0b, P0a);
                    } else {
                        kstep(s, P0a, P0a);
                        kstep(s + 1, P0a, P0a);
                    }
                }
                wait_frag<0, G>(P0a);                         // drain (harmless re-loads of the last k-step)
                wait_frag<0, G>(P1);
                wait_frag<0, G>(P2);
            }
        }
        __syncthreads();  // every wave is done reading x_t / h_{t-1}

        // ---- cell update in registers; h_t goes back to LDS (split) for the next step ----------------------
        const int lane_o = fresh_lane();    // row offsets are re-derived in every step: hoisted out of the t loop they spill
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            float hv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = r0 + 32 * rb + acc_row(r, lane_o);
                const int q = q0 + row;
                float h;
                if (GRU) {
                    // saved: r, z, n, the pre-activation W_hn h + b_hn, h_{t-1}
                    const float rg = sigmoidf_(acc[rb][0][r]);
                    const float zg = sigmoidf_(acc[rb][G > 1 ? 1 : 0][r]);
                    const float nh = acc[rb][G > 3 ? 3 : 0][r];
                    const float ng = tanhf_(acc[rb][G > 2 ? 2 : 0][r] + rg * nh);
                    const float hp = cst[rb][r];
                    h = (1.0f - zg) * ng + zg * hp;
                    cst[rb][r] = h;
                    if (saved_t && q < p.P) {
                        float *sv = &at_bytes(saved_t, (((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)(SV * H) + col) * 4u);
                        sv[0] = rg; sv[H] = zg; sv[2 * H] = ng; sv[3 * H] = nh; sv[4 * H] = hp;
                    }
                } else if (G == 4) {
                    const float ig = sigmoidf_(acc[rb][0][r]);
                    const float fg = sigmoidf_(acc[rb][G > 1 ? 1 : 0][r]);
                    const float gg = tanhf_(acc[rb][G > 2 ? 2 : 0][r]);
                    const float og = sigmoidf_(acc[rb][G > 3 ? 3 : 0][r]);
                    const float c = fg * cst[rb][r] + ig * gg;
                    cst[rb][r] = c;
                    h = og * tanhf_(c);
                    if (saved_t && q < p.P) {
                        // (constant displacements on top of base + zext(offset) fold into the instructions' immediates)
                        float *sv = &at_bytes(saved_t, (((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)(SV * H) + col) * 4u);
                        sv[0] = ig; sv[H] = fg; sv[2 * H] = gg; sv[3 * H] = og; sv[4 * H] = c;
                    }
                } else {
                    h = tanhf_(acc[rb][0][r]);
                    if (saved_t && q < p.P) at_bytes(saved_t, (((uint32_t)row * (uint32_t)p.L + t) * (uint32_t)H + col) * 4u) = h;
                }
                hv[r] = h;
                if (q < p.P) {
                    if (t == p.L - 1)
                        at_bytes(hn_t, ((uint32_t)row * (uint32_t)H + col) * 4u) = h;
                    else if (xh_t)
                        at_bytes(xh_t, (((uint32_t)row * (uint32_t)p.L + t + 1) * (uint32_t)(2 * H) + H + col) * 4u) = h;
                }
            }
            if (t + 1 < p.L) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {       // accumulator registers r, r+1 are tile rows row, row+1
                    uint32_t h0, h1, h2;
                    split3(hv[r], hv[r + 1], h0, h1, h2);
                    unsigned char *d = ldsb + (r0 + 32 * rb + acc_row(r, lane_o)) * PB + 2 * (H + col);
                    *reinterpret_cast<uint16_t *>(d) = (uint16_t)h0;
                    *reinterpret_cast<uint16_t *>(d + PB) = (uint16_t)(h0 >> 16);
                    *reinterpret_cast<uint16_t *>(d + PLANE) = (uint16_t)h1;
                    *reinterpret_cast<uint16_t *>(d + PLANE + PB) = (uint16_t)(h1 >> 16);
                    *reinterpret_cast<uint16_t *>(d + 2 * PLANE) = (uint16_t)h2;
                    *reinterpret_cast<uint16_t *>(d + 2 * PLANE + PB) = (uint16_t)(h2 >> 16);
                }
            }
        }
        if (t + 1 < p.L) {
            tid_g = wave_u * 64 + fresh_lane();
            if (!PREFETCH_X) gather_issue(t + 1);
            gather_commit(t + 1);     // (every wave is past its reads of x_t)
            __syncthreads();
        }
    }
}


// ---- BPTT through the recurrent cell, fused with the gather-backward scatter ---------------------
// (SeqBwdParams: pn_seq.h)

// ---- the same BPTT on the bf16 matrix pipe (pn_kernels.h) --------------------------------------------------------
//   [dx_t | dh_{t-1}] = dG_t [32, G*H] . [W_ih | W_hh]:  K = G*H gate columns, wave w owns columns 32w..32w+31 of dx and of dh.
//   Weights: pack_bwd3_kernel, B fragments grouped in units of two k-steps (kk) x two output halves (nt),
//     WpT3[(((w*NU + u)*3 + plane)*4 + kk*2 + nt)*64 + lane] (16 bytes) =
//         plane of Wcat[k = 32u + 16kk + 8(lane >> 5) .. +7][n = nt*H + 32w + (lane & 31)],      NU = G*H/32 units.
//   A operand: the three bf16 planes of dG_t in LDS.  All four LSTM gates would take 3 x 32 x 4H x 2 B = 96 KB per
//     workgroup (one workgroup per CU); the tile therefore holds one gate pair at a time -- (i, f) then (g, o), K = 2H
//     each, 50 KB -- and the k loop runs in two passes with the (g, o) gradients parked in registers meanwhile.
__global__ void pack_bwd3_kernel(const float *__restrict__ w_ih, const float *__restrict__ w_hh, int H, int G, int gru,
                                 u32x4 *__restrict__ WpT) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
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
    u32x4 q0, q1, q2;
#pragma unroll
    for (int h = 0; h < 4; h++) {
        uint32_t x0, x1, x2;
        split3(v[2 * h], v[2 * h + 1], x0, x1, x2);
        q0[h] = x0; q1[h] = x1; q2[h] = x2;
    }
    u32x4 *dst = WpT + ((int64_t)(w * NU + u) * 3 * 4 + f) * 64 + lane;
    dst[0] = q0;
    dst[4 * 64] = q1;
    dst[8 * 64] = q2;
}

// RG row groups per workgroup share the weight stream through L1, as in seq_fwd3_kernel.
template <int H, int GC, int RG>
__global__ __launch_bounds__(H / 32 * 64 * RG, H == 32 && RG == 1 ? 1 : PN_BWD_WAVES) void seq_bwd3_kernel(SeqBwdParams p) {
    constexpr int G = GC == 3 ? 4 : GC;         // GC: 4 = LSTM, 1 = tanh RNN, 3 = GRU on the LSTM's four gate slots
    constexpr bool GRU = GC == 3;
    constexpr int MT = 32 * RG, NW = H / 32;
    constexpr int NT = NW * 64 * RG, GH = G * H, SV = (G == 4 ? 5 : 1);
    constexpr int NPASS = G == 4 ? 2 : 1, KP = GH / NPASS;      // K extent of one pass (one gate pair)
    constexpr int PB = 2 * KP + 16, PLANE = MT * PB;            // plane row pitch / plane size, bytes
    constexpr int NU = GH / 32, NUP = NU / NPASS;               // units of two k-steps, total / per pass
    constexpr int NB = PN_BWD_NB;       // accumulator elements per batch of saved-tensor loads in the cell backward
    constexpr bool CARRY_C = false;     // true: c_t stays in registers from one step to the next (16 registers the kernel does not have)
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    int *s_rowidx = reinterpret_cast<int *>(ldsb + 3 * PLANE);   // [MT][L] gather rows of this tile
    int *s_slotof = s_rowidx + MT * p.L;                         // [MT]
    uint8_t *s_keep = reinterpret_cast<uint8_t *>(s_slotof + MT); // [2][MT][H/4] dropout keep bits of step t (t & 1)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31;
    const int ws = wave % NW, r0 = 32 * (wave / NW);             // column slice / first tile row of this wave
    // tiles in descending order -- the forward wrote the saved tensors of the last tiles last, they are the ones still in
    // the 256 MB Infinity Cache when the backward starts
    const int q0 = (int)(gridDim.x - 1 - blockIdx.x) * MT;
    const int col = 32 * ws + li;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave), ws_u = __builtin_amdgcn_readfirstlane(ws);

    for (int i = tid; i < MT * p.L; i += NT) {
        const int q = q0 + i / p.L;
        s_rowidx[i] = q < p.P ? p.rowidx[(int64_t)q0 * p.L + i] : 0;
    }
    for (int i = tid; i < MT; i += NT) s_slotof[i] = q0 + i < p.P ? p.slotof[q0 + i] : 0;

    const float keep_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (1.0f - p.p_drop))));
    // 64-bit tile bases (wave-uniform) + 32-bit offsets inside the tile; rows past P read the tile's clamped last row
    const size_t tile_row = (size_t)q0 * (size_t)p.L;
    const float *saved_t = p.saved + tile_row * (SV * H);
    float *dG_t = p.dG + tile_row * GH;
    const uint8_t *keep_t = p.keep ? p.keep + tile_row * (H / 4) : nullptr;
    const float *dhn_t = p.dhn + (size_t)q0 * H;
    const int rows_here = min(MT, p.P - q0);            // >= 1
    f32x16 dh, dc, cnext;   // cnext: c_t of the step processed next (= c_{t-1} now)
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = r0 + acc_row(r, lane);
        const int rc = min(row, rows_here - 1);
        const float dh0 = at_bytes(dhn_t, ((uint32_t)rc * (uint32_t)H + col) * 4u);     // unconditional load, select afterwards
        dh[r] = row < rows_here ? dh0 : 0.0f;
        dc[r] = 0.0f;
        cnext[r] = G == 4 && CARRY_C ? at_bytes(saved_t, ((((uint32_t)rc * (uint32_t)p.L + (p.L - 1)) * SV + 4) * (uint32_t)H + col) * 4u) : 0.0f;
    }

    // the bf16 planes of two tile rows (accumulator registers r, r+1) of gate slot gs (0 or 1) of the resident pair
    auto put_pair = [&](int r, int lane_t, int gs, float v0, float v1) {
        uint32_t x0, x1, x2;
        split3(v0, v1, x0, x1, x2);
        unsigned char *d = ldsb + (r0 + acc_row(r, lane_t)) * PB + 2 * (gs * H + col);
        *reinterpret_cast<uint16_t *>(d) = (uint16_t)x0;
        *reinterpret_cast<uint16_t *>(d + PB) = (uint16_t)(x0 >> 16);
        *reinterpret_cast<uint16_t *>(d + PLANE) = (uint16_t)x1;
        *reinterpret_cast<uint16_t *>(d + PLANE + PB) = (uint16_t)(x1 >> 16);
        *reinterpret_cast<uint16_t *>(d + 2 * PLANE) = (uint16_t)x2;
        *reinterpret_cast<uint16_t *>(d + 2 * PLANE + PB) = (uint16_t)(x2 >> 16);
    };

    for (int t = p.L - 1; t >= 0; t--) {
        // (row numbers are re-derived from an opaque copy of the lane id in every step: as loop invariants the
        //  per-row offsets would occupy ~40 registers across the MFMA loops and spill)
        const int lane_t = fresh_lane();
        if (p.keep) {      // this step's keep bytes (MT rows x H/4) -> LDS, read by the scatter phase below
            const int tid_t = wave_u * 64 + lane_t;    // (offsets re-derived per step, see lane_t)
            for (int i = tid_t; i < MT * (H / 16); i += NT) {
                const int row = i / (H / 16), w = i - row * (H / 16);
                const uint32_t rc = (uint32_t)min(row, rows_here - 1);
                reinterpret_cast<uint32_t *>(s_keep + (t & 1) * MT * (H / 4))[i] =
                    at_bytes(reinterpret_cast<const uint32_t *>(keep_t), (rc * (uint32_t)p.L + t) * (uint32_t)(H / 4) + 4u * w);
            }
        }
        // ---- cell backward.  All loads of a batch of NB accumulator elements are issued together (unconditionally, padded rows read a
        //      clamped row and are zeroed afterwards) so the wave pays one memory round trip, not one per element.
        float ag[16], ao[16];      // (g, o) gate gradients wait here for the second pass
#pragma unroll
        for (int half = 0; half < 16 / NB; half++) {
            float vi[NB], vf[NB], vg[NB], vo[NB], vc[NB], vn[NB];
#pragma unroll
            for (int e = 0; e < NB; e++) {
                const int r = half * NB + e;
                const int rc = min(r0 + acc_row(r, lane_t), rows_here - 1);
                const float *sv = &at_bytes(saved_t, (((uint32_t)rc * (uint32_t)p.L + t) * (uint32_t)(SV * H) + col) * 4u);
                if (GRU) {
                    vi[e] = sv[0]; vf[e] = sv[H]; vg[e] = sv[2 * H];     // r, z, n
                    vo[e] = sv[3 * H];                                    // W_hn h + b_hn
                    vc[e] = sv[4 * H];                                    // h_{t-1}
                    vn[e] = 0.0f;
                } else if (G == 4) {
                    vi[e] = sv[0]; vf[e] = sv[H]; vg[e] = sv[2 * H];
                    vo[e] = sv[3 * H];
                    vc[e] = t > 0 ? sv[-H] : 0.0f;                  // c_{t-1} = slot 4 of step t-1
                    vn[e] = CARRY_C ? cnext[r] : sv[4 * H];         // c_t
                } else {
                    vi[e] = sv[0];                                   // h_t
                }
            }
            float ai[NB], af[NB];
#pragma unroll
            for (int e = 0; e < NB; e++) {
                const int r = half * NB + e;
                const int row = r0 + acc_row(r, lane_t);
                const bool ok = row < rows_here;
                float *d = &at_bytes(dG_t, (((uint32_t)min(row, rows_here - 1) * (uint32_t)p.L + t) * (uint32_t)GH + col) * 4u);
                if (GRU) {
                    // h = (1 - z) n + z h_prev,  n = tanh(nx + r nh):  gradients of the four slots r, z, nx, nh; the direct
                    // path d h_t / d h_{t-1} = z is carried in dc[] across the GEMM and added to its dh output
                    const float rg = vi[e], zg = vf[e], ng = vg[e], nh = vo[e], hp = vc[e];
                    const float dhv = dh[r];
                    const float dnp = dhv * (1.0f - zg) * (1.0f - ng * ng);
                    float a_r = dnp * nh * rg * (1.0f - rg);
                    float a_z = dhv * (hp - ng) * zg * (1.0f - zg);
                    float a_nx = dnp;
                    float a_nh = dnp * rg;
                    if (!ok) a_r = a_z = a_nx = a_nh = 0.0f;
                    dc[r] = ok ? dhv * zg : 0.0f;
                    ai[e] = a_r; af[e] = a_z; ag[r] = a_nx; ao[r] = a_nh;
                    if (ok) {
                        d[0] = a_r; d[H] = a_z; d[2 * (G > 1 ? H : 0)] = a_nx; d[3 * (G > 1 ? H : 0)] = a_nh;
                    }
                } else if (G == 4) {
                    const float ig = vi[e], fg = vf[e], gg = vg[e], og = vo[e], cprev = vc[e];
                    const float tc = tanhf_(vn[e]);
                    const float dhv = dh[r];
                    const float d_o = dhv * tc;
                    const float dct = dc[r] + dhv * og * (1.0f - tc * tc);
                    float a_i = dct * gg * ig * (1.0f - ig);
                    float a_f = dct * cprev * fg * (1.0f - fg);
                    float a_g = dct * ig * (1.0f - gg * gg);
                    float a_o = d_o * og * (1.0f - og);
                    if (!ok) a_i = a_f = a_g = a_o = 0.0f;
                    dc[r] = dct * fg;
                    if (CARRY_C) cnext[r] = cprev;
                    ai[e] = a_i; af[e] = a_f; ag[r] = a_g; ao[r] = a_o;
                    if (ok) {
                        d[0] = a_i; d[H] = a_f; d[2 * (G > 1 ? H : 0)] = a_g; d[3 * (G > 1 ? H : 0)] = a_o;
                    }
                } else {
                    const float h = vi[e];
                    const float a = ok ? dh[r] * (1.0f - h * h) : 0.0f;
                    ai[e] = a;
                    if (ok) d[0] = a;
                }
            }
#pragma unroll
            for (int e = 0; e < NB; e += 2) {
                put_pair(half * NB + e, lane_t, 0, ai[e], ai[e + 1]);
                if (G == 4) put_pair(half * NB + e, lane_t, 1, af[e], af[e + 1]);
            }
        }
        __syncthreads();

        // ---- [dx_t ; dh_{t-1}] = dG_t . [W_ih | W_hh]; the dh half is not needed at t = 0 ------------------------
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = 0.0f;
        const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.WpT) + (size_t)ws_u * (NU * 12 * 1024);
        const uint32_t voff = lane_t * 16;
        const unsigned char *arow = ldsb + (r0 + (lane_t & 31)) * PB + 16 * (lane_t >> 5);
        // The weight stream runs through both passes without a break: the last unit of the (i, f) pass prefetches the
        // first unit of the (g, o) pass, whose fragments are then in flight across the two barriers in between.  Same
        // fragment pipeline as seq_fwd3_kernel: vmcnt (in order) sees [P0(u) P1(u) P2(u) P0(u+1)] at the top of unit u.
        auto mfma_phase = [&](auto ntn_tag) {
            constexpr int NTN = decltype(ntn_tag)::value, NF = 2 * NTN;       // fragments per plane and unit
            u32x4 P0a[NF], P0b[NF], P1[NF], P2[NF];
            auto load = [&](u32x4 (&B)[NF], int u, int pl) {      // fragment kk*2 + nt of the unit's plane
                const unsigned char *sb = wb + (size_t)(u * 3 + pl) * 4096;
                if constexpr (NTN == 2) {
                    async_load_frags<4>(B, sb, voff);
                } else {
                    async_load_b128_s<0>(B[0], sb, voff);
                    async_load_b128_s<2048>(B[1], sb, voff);
                }
            };
            // A fragments pipelined in place, as in seq_fwd3_kernel: products run a2.P0, a1.P0, a0.P0 | a1.P1, a0.P1 |
            // a0.P2 and each plane of the A tile is re-read for the next unit right after its last use.  The last unit
            // of a pass does not read ahead (the tile is rewritten behind the barrier); the first one reads up front.
            u32x4 a[2][3];
            auto aread = [&](int ul, int pl) {          // ul = unit index within the pass
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
                    a[kk][pl] = *reinterpret_cast<const u32x4 *>(arow + pl * PLANE + 64 * ul + 32 * kk);
            };
            auto group = [&](const u32x4 (&B)[NF], int pl) {
#pragma unroll
                for (int f = 0; f < NF; f++) acc[f % NTN] = mfma_bf16(a[f / NTN][pl], B[f], acc[f % NTN]);
            };
            auto unit = [&](int u, int ub, u32x4 (&P0)[NF], u32x4 (&P0next)[NF]) {
                const int un = min(u + 1, NU - 1);
                const int ul = u - ub;
                const bool ahead = ul + 1 < NUP;        // block-uniform
                load(P0next, un, 0);
                if (ul == 0) {
                    aread(0, 0);
                    aread(0, 1);
                    aread(0, 2);
                }
                wait_frag<3 * NF, NF>(P0);
                group(P0, 2);
                if (ahead) aread(ul + 1, 2);
                group(P0, 1);
                group(P0, 0);
                wait_frag<2 * NF, NF>(P1);
                group(P1, 1);
                if (ahead) aread(ul + 1, 1);
                group(P1, 0);
                load(P1, un, 1);
                wait_frag<2 * NF, NF>(P2);
                group(P2, 0);
                if (ahead) aread(ul + 1, 0);
                load(P2, un, 2);
            };
            load(P0a, 0, 0);
            load(P1, 0, 1);
            load(P2, 0, 2);
#pragma unroll
            for (int pass = 0; pass < NPASS; pass++) {
                const int ub = pass * NUP, ue = ub + NUP;
                if (pass > 0) {
                    __syncthreads();             // every wave is done with the (i, f) planes
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        put_pair(r, lane_t, 0, ag[r], ag[r + 1]);
                        put_pair(r, lane_t, 1, ao[r], ao[r + 1]);
                    }
                    __syncthreads();
                }
                static_assert(NPASS == 1 || NUP % 2 == 0, "the plane-0 ping-pong must be in phase at the pass boundary");
#pragma unroll 1
                for (int u = ub; u < ue; u += 2) {
                    unit(u, ub, P0a, P0b);
                    if (NUP % 2 == 0 || u + 1 < ue) unit(u + 1, ub, P0b, P0a);
                }
            }
            wait_frag<0, NF>(P0a);       // drain (harmless re-loads of the last unit)
            wait_frag<0, NF>(P0b);
            wait_frag<0, NF>(P1);
            wait_frag<0, NF>(P2);
        };
        if (t > 0)
            mfma_phase(std::integral_constant<int, 2>{});
        else
            mfma_phase(std::integral_constant<int, 1>{});
        __syncthreads();

        // ---- gather backward: dZ[row(q, t)] += mask * dx.  Step 0 is the last one of the kernel and its rows are the
        //      paths' own start nodes: the W paths of a node all add to the same table row, 32 atomics per column on one
        //      address (0.033 of the kernel's 0.49 ms, by ablation).  There the wave parks its 32 x 32 block in the (now dead)
        //      plane region, and each half-wave walks 16 rows in order, adding up runs of equal table rows: one atomic per
        //      run and column -- two to four instead of thirty-two.
        // (hidden sizes that are not powers of two sit at the register limit already: they keep the plain scatter)
        constexpr bool MERGE_STEP0 = (H & (H - 1)) == 0;
        if (MERGE_STEP0 && t == 0 && p.merge0) {
            static_assert(3 * PLANE >= NW * RG * 32 * 33 * 4, "the scatter scratch fits the plane region");
            const int lane_s = fresh_lane(), li_s = lane_s & 31;       // (re-derived here: nothing of this block is hoisted)
            float *scr = reinterpret_cast<float *>(ldsb) + wave_u * (32 * 33);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rl = acc_row(r, lane_s), row = r0 + rl;
                float dx = acc[0][r];
                if (q0 + row < p.P) {
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
                const int rl = 16 * hk + i, row = r0 + rl;
                const int rid = q0 + row < p.P ? s_rowidx[row * p.L] : -1;       // (uniform over a half-wave)
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
                const int row = r0 + acc_row(r, lane_t);
                if (q0 + row < p.P) {
                    float dx = acc[0][r];
                    if (p.mask)
                        dx *= p.mask[((uint64_t)t * p.Pmask + s_slotof[row]) * H + col];
                    else if (p.keep)
                        dx = (s_keep[((t & 1) * MT + row) * (H / 4) + (col >> 2)] >> (col & 3)) & 1 ? dx * keep_scale : 0.0f;
                    atomicAdd(p.dZ + ((size_t)(uint32_t)s_rowidx[row * p.L + t] * (uint32_t)H + col), dx);
                }
                dh[r] = GRU ? acc[1][r] + dc[r] : acc[1][r];
            }
        }
    }
}

// ---- recurrent weight gradients:  [g_W_ih | g_W_hh]  [G*H, 2H] = dG^T [G*H, R] . XH [R, 2H],  R = P*L rows,
//      plus the bias gradient colsum(dG).  Both operands are row-major with the reduction dimension
//      outermost, i.e. already "K-major": tiles go global -> LDS with coalesced 16-byte loads and no
//      transposition.  128x128 output tile per workgroup (4 waves x (2x2) 32x32 MFMA tiles), the R rows
//      are split over blockIdx.z; partial tiles go to a [split][G*H][2H] buffer and are summed by
//      wgrad_reduce_kernel (deterministic, no atomics). -----------------------------------------------
constexpr int WG_THREADS = 512;

// (WgradParams: pn_seq.h)

// ---- the same GEMM on the bf16 matrix pipe (pn_kernels.h: six bf16 MFMAs = one fp32-accurate product) -----------
// Both operands have the reduction dimension (rows) outermost, the bf16 MFMA wants 8 consecutive k per lane.  A
// thread therefore fetches an 8-row x 4-column fp32 block (8 coalesced 16-byte loads), splits the 32 values into
// their three bf16 planes in registers and writes, per column, one 16-byte k-octet per plane: the in-register
// transposition costs nothing.  LDS image per (plane, operand, k-octet kb = 0..3 of the 32-row K tile): 256 columns,
// column c at 16-byte slot (c & 3) * 68 + (c >> 2) -- consecutive lanes write consecutive slots (conflict-free
// ds_write_b128) and the 16-lane groups of the fragment ds_read_b128 hit 16 distinct bank quads.
constexpr int W3_BLK = 4 * 68;                          // slots per k-octet block
constexpr int W3_PLANE = 2 * 4 * W3_BLK;                // slots per plane (2 operands x 4 k-octets)
constexpr int W3_LDS_BYTES = 3 * W3_PLANE * 16;         // 104 448 B

__global__ __launch_bounds__(WG_THREADS, 2) void wgrad3_kernel(WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds4[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * WG_BM, n0 = blockIdx.x * WG_BN;
    // split z takes the K tiles z, z + nz, z + 2 nz, ...: every workgroup starts on the low rows, which the (reversed)
    // BPTT kernel wrote last and which are still in the Infinity Cache
    const int64_t rbeg = (int64_t)blockIdx.z * WG_KT, rend = p.R, kstep = (int64_t)gridDim.z * WG_KT;
    if (rbeg >= rend) return;   // block-uniform
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // staging task of this thread: k-octet ro of the K tile, operand op, columns 4*cql .. +3
    const int ro = tid >> 7, cq = tid & 127, op = cq >> 6, cql = cq & 63;
    const float *src = op == 0 ? p.dG : p.xh;
    const int ld = op == 0 ? p.GH : p.H2;
    const int c0 = (op == 0 ? m0 : n0) + 4 * cql;
    const bool c_ok = c0 < ld;
    const float *srcc = src + (c_ok ? c0 : 0);
    f32x4 rg[8];
    auto issue = [&](int64_t k0) {
#pragma unroll
        for (int e = 0; e < 8; e++) async_load_b128(rg[e], srcc + min(k0 + 8 * ro + e, rend - 1) * ld);
    };
    float bs[4] = {0.f, 0.f, 0.f, 0.f};     // column sums of dG over this thread's rows (bias gradient)
    u32x4 *stage = lds4 + (op * 4 + ro) * W3_BLK + cql;
    const int sa = (li & 3) * 68 + (li >> 2) + wm * 16, sb = (li & 3) * 68 + (li >> 2) + wn * 32;

    issue(rbeg);
    [[maybe_unused]] int tile_i = 0;     // (tuning builds stamp tiles 8..19, five stamps per tile)
    [[maybe_unused]] const int wblk = blockIdx.z * gridDim.y + blockIdx.y;
    for (int64_t k0 = rbeg; k0 < rend; k0 += kstep, tile_i++) {
        wait_vm<0>(rg[0], rg[1], rg[2], rg[3], rg[4], rg[5], rg[6], rg[7]);
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (!(c_ok && k0 + 8 * ro + e < rend)) rg[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32x4 q0, q1, q2;
#pragma unroll
            for (int h = 0; h < 4; h++) {
                uint32_t x0, x1, x2;
                split3(rg[2 * h][j], rg[2 * h + 1][j], x0, x1, x2);
                q0[h] = x0; q1[h] = x1; q2[h] = x2;
                bs[j] += rg[2 * h][j] + rg[2 * h + 1][j];
            }
            stage[j * 68] = q0;
            stage[W3_PLANE + j * 68] = q1;
            stage[2 * W3_PLANE + j * 68] = q2;
        }
        __syncthreads();
        issue(min(k0 + kstep, rend - 1));   // next tile in flight under the MFMAs (last trip: harmless re-load)
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const u32x4 *fa = lds4 + (kk * 2 + hk) * W3_BLK + sa;             // operand 0 (dG^T)
            const u32x4 *fb = lds4 + (4 + kk * 2 + hk) * W3_BLK + sb;         // operand 1 ([x|h])
            u32x4 a0[2], a1[2], b0[4], b1[4];
#pragma unroll
            for (int i = 0; i < 2; i++) a0[i] = fa[i * 8];
#pragma unroll
            for (int j = 0; j < 4; j++) b0[j] = fb[j * 8];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = mfma_bf16(a0[i], b0[j], acc[i][j]);
#pragma unroll
            for (int j = 0; j < 4; j++) b1[j] = fb[W3_PLANE + j * 8];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = mfma_bf16(a0[i], b1[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; i++) a1[i] = fa[W3_PLANE + i * 8];
            // a1.b1 before a1.b0: the plane-2 fragments can then be fetched into the registers of b1 (and next of a1)
            // under the MFMAs that follow, instead of stalling the matrix pipe for two LDS round trips per k-step
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = mfma_bf16(a1[i], b1[j], acc[i][j]);
#pragma unroll
            for (int j = 0; j < 4; j++) b1[j] = fb[2 * W3_PLANE + j * 8];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = mfma_bf16(a1[i], b0[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; i++) a1[i] = fa[2 * W3_PLANE + i * 8];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = mfma_bf16(a0[i], b1[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = mfma_bf16(a1[i], b0[j], acc[i][j]);
        }
        __syncthreads();
    }
    wait_vm<0>(rg[0], rg[1], rg[2], rg[3], rg[4], rg[5], rg[6], rg[7]);   // drain the trailing prefetch
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
    // bias gradient: the four k-octet owners of a column add up through LDS
    if (blockIdx.x != 0) return;   // block-uniform
    float *fl = reinterpret_cast<float *>(lds4);
    if (op == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) fl[ro * WG_BM + 4 * cql + j] = bs[j];
    }
    __syncthreads();
    if (tid < WG_BM && m0 + tid < p.GH)
        p.part_b[(int64_t)blockIdx.z * p.GH + m0 + tid] =
            (fl[tid] + fl[WG_BM + tid]) + (fl[2 * WG_BM + tid] + fl[3 * WG_BM + tid]);
}

// sums the split partials and scatters them into the reference layouts g_W_ih [GH,H], g_W_hh [GH,H], g_b_*
// (accumulate != 0: added to what the previous micro-batches left there)
// (gru: the four slots r, z, nx, nh map to torch's [3H, H] layouts: W_i{r,z,n} = x halves of slots 0, 1, 2,
//  W_h{r,z,n} = h halves of slots 0, 1, 3; b_i{r,z,n} = slots 0, 1, 2, b_h{r,z,n} = slots 0, 1, 3)
// A block of 256 threads sums 64 float4 columns: thread (zg = tid >> 6, c = tid & 63) adds the splits z = zg, zg + 4, ...
// with eight loads in flight, the four partial sums meet in LDS (fixed order: deterministic).  (One thread per element
// walking all splits alone took 46 us for the 67 MB of partials of the bench workload.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part_w,
                                                           const float *__restrict__ part_b, int nsplit, int GH, int H,
                                                           int accumulate, int gru, float *__restrict__ g_w_ih,
                                                           float *__restrict__ g_w_hh, float *__restrict__ g_b_ih,
                                                           float *__restrict__ g_b_hh) {
    __shared__ float4 red[3][64];
    const int zg = threadIdx.x >> 6, c = threadIdx.x & 63;
    const int64_t nw = (int64_t)GH * 2 * H, ntot = nw + GH;      // weights [GH, 2H], then the bias sums [GH]
    const int64_t i0 = ((int64_t)blockIdx.x * 64 + c) * 4;       // (nw and GH are multiples of 4: no float4 straddles)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 < ntot) {
        const bool bias = i0 >= nw;
        const float *src = bias ? part_b + (i0 - nw) : part_w + i0;
        const int64_t pitch = bias ? GH : nw;
        float4 t[8];
        int z = zg;
        for (; z + 28 < nsplit; z += 32) {
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = *reinterpret_cast<const float4 *>(src + (int64_t)(z + 4 * u) * pitch);
#pragma unroll
            for (int u = 0; u < 8; u++) s.x += t[u].x, s.y += t[u].y, s.z += t[u].z, s.w += t[u].w;
        }
        for (; z < nsplit; z += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)z * pitch);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    }
    if (zg > 0) red[zg - 1][c] = s;
    __syncthreads();
    if (zg > 0 || i0 >= ntot) return;
#pragma unroll
    for (int k = 0; k < 3; k++) s.x += red[k][c].x, s.y += red[k][c].y, s.z += red[k][c].z, s.w += red[k][c].w;
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int64_t i = i0 + e;
        const float v = sv[e];
        if (i < nw) {
            int m = (int)(i / (2 * H));
            const int n = (int)(i - (int64_t)m * 2 * H);
            const bool xhalf = n < H;
            int row = m;
            if (gru) {
                const int slot = m / H, j = m - slot * H;
                if ((slot == 2 && !xhalf) || (slot == 3 && xhalf)) continue;        // products with the zero halves
                row = gru_weight_row(slot, j, H);
            }
            float *dst = xhalf ? g_w_ih : g_w_hh;
            if (!dst) continue;
            dst += (int64_t)row * H + (xhalf ? n : n - H);
            *dst = accumulate ? *dst + v : v;
        } else {
            int m = (int)(i - nw);
            if (gru) {
                const int slot = m / H, j = m - slot * H, wr = gru_weight_row(slot, j, H);
                if (g_b_ih && slot != 3) g_b_ih[wr] = accumulate ? g_b_ih[wr] + v : v;
                if (g_b_hh && slot != 2) g_b_hh[wr] = accumulate ? g_b_hh[wr] + v : v;
            } else {
                if (g_b_ih) g_b_ih[m] = accumulate ? g_b_ih[m] + v : v;
                if (g_b_hh) g_b_hh[m] = accumulate ? g_b_hh[m] + v : v;
            }
        }
    }
}


template <int H, int G>
int launch_seq_bwd(pn_context *ctx, hipStream_t stream, const SeqBwdParams &sp) {
    constexpr int RG = H <= 128 ? PN_SEQ_RG : 1;
    constexpr int MT = 32 * RG;
    const size_t lds_bytes = (size_t)3 * MT * (2 * (G >= 3 ? 2 * H : H) + 16) + (size_t)(MT * sp.L + MT) * 4 +
                             (size_t)2 * MT * (H / 4);      // (G = 3: GRU, on four gate slots)
    auto kern = seq_bwd3_kernel<H, G, RG>;
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), (int)lds_bytes)) return rc;
    const int blocks = (sp.P + MT - 1) / MT;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(H / 32 * 64 * RG), lds_bytes, stream, sp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

template <int G>
int dispatch_seq_bwd(pn_context *ctx, hipStream_t stream, int H, const SeqBwdParams &sp) {
    switch (H) {      // every multiple of 32 up to 256 (the LDS tile of H = 256 is 98 KB)
        case 32: return launch_seq_bwd<32, G>(ctx, stream, sp);
        case 64: return launch_seq_bwd<64, G>(ctx, stream, sp);
        case 96: return launch_seq_bwd<96, G>(ctx, stream, sp);
        case 128: return launch_seq_bwd<128, G>(ctx, stream, sp);
        case 160: return launch_seq_bwd<160, G>(ctx, stream, sp);
        case 192: return launch_seq_bwd<192, G>(ctx, stream, sp);
        case 224: return launch_seq_bwd<224, G>(ctx, stream, sp);
        case 256: return launch_seq_bwd<256, G>(ctx, stream, sp);
    }
    PN_FAIL(PN_ERR_ARG, "hidden size %d not supported", H);
}

template <int H, int G>
int launch_seq_fwd(pn_context *ctx, hipStream_t stream, const SeqFwdParams &sp) {
    constexpr int RG = H <= 128 ? PN_SEQ_RG : 1;      // H = 256: one row group already fills the LDS
    constexpr int RB = 1;       // row blocks of 32 paths per wave (2 measured slower: 0.411 vs 0.308 ms, profiles/HISTORY_r1_r4.md)
    constexpr int MT = 32 * RG * RB;
    const size_t lds_bytes = (size_t)3 * MT * (4 * H + 16) + (size_t)(MT * sp.L + MT) * 4;
    auto kern = seq_fwd3_kernel<H, G, RG, RB>;
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), (int)lds_bytes)) return rc;
    const int blocks = (sp.P + MT - 1) / MT;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(H / 32 * 64 * RG), lds_bytes, stream, sp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

template <int G>
int dispatch_seq_fwd(pn_context *ctx, hipStream_t stream, int H, const SeqFwdParams &sp) {
    switch (H) {
        case 32: return launch_seq_fwd<32, G>(ctx, stream, sp);
        case 64: return launch_seq_fwd<64, G>(ctx, stream, sp);
        case 96: return launch_seq_fwd<96, G>(ctx, stream, sp);
        case 128: return launch_seq_fwd<128, G>(ctx, stream, sp);
        case 160: return launch_seq_fwd<160, G>(ctx, stream, sp);
        case 192: return launch_seq_fwd<192, G>(ctx, stream, sp);
        case 224: return launch_seq_fwd<224, G>(ctx, stream, sp);
        case 256: return launch_seq_fwd<256, G>(ctx, stream, sp);
    }
    PN_FAIL(PN_ERR_ARG, "hidden size %d not supported", H);
}



}  // namespace

namespace pn {

int launch_pack_fwd3(void *stream, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, int H, int G, int gru,
                     void *Wp, float *biasc) {
    hipLaunchKernelGGL(pack_fwd3_kernel, dim3((unsigned)((G * H * H / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh,
                       b_ih, b_hh, H, G, gru, reinterpret_cast<u32x4 *>(Wp), biasc);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int launch_pack_bwd3(void *stream, const float *w_ih, const float *w_hh, int H, int G, int gru, void *WpT) {
    hipLaunchKernelGGL(pack_bwd3_kernel, dim3((unsigned)((G * H * H / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh,
                       H, G, gru, reinterpret_cast<u32x4 *>(WpT));
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int launch_seq_fwd3(pn_context *ctx, void *stream, int H, int gc, const SeqFwdParams &sp) {
    hipStream_t s = (hipStream_t)stream;
    return gc == 3 ? dispatch_seq_fwd<3>(ctx, s, H, sp) : gc == 4 ? dispatch_seq_fwd<4>(ctx, s, H, sp) : dispatch_seq_fwd<1>(ctx, s, H, sp);
}

int launch_seq_bwd3(pn_context *ctx, void *stream, int H, int gc, const SeqBwdParams &sp) {
    hipStream_t s = (hipStream_t)stream;
    return gc == 3 ? dispatch_seq_bwd<3>(ctx, s, H, sp) : gc == 4 ? dispatch_seq_bwd<4>(ctx, s, H, sp) : dispatch_seq_bwd<1>(ctx, s, H, sp);
}

int launch_wgrad3(pn_context *ctx, void *stream, const WgradParams &wp, int nsplit) {
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(wgrad3_kernel), W3_LDS_BYTES)) return rc;
    hipLaunchKernelGGL(wgrad3_kernel, dim3((wp.H2 + WG_BN - 1) / WG_BN, (wp.GH + WG_BM - 1) / WG_BM, nsplit), dim3(WG_THREADS),
                       W3_LDS_BYTES, (hipStream_t)stream, wp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int launch_wgrad_reduce(void *stream, const float *part_w, const float *part_b, int nsplit, int GH, int H, int accumulate, int gru,
                        float *g_w_ih, float *g_w_hh, float *g_b_ih, float *g_b_hh) {
    const int64_t nred = (int64_t)GH * 2 * H + GH;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nred + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part_w, part_b,
                       nsplit, GH, H, accumulate, gru, g_w_ih, g_w_hh, g_b_ih, g_b_hh);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

}  // namespace pn
