// pn_seq4.hip -- the three recurrent kernels of the aggregator for the headline shape (hidden size 128, four gate
// slots: nn.LSTM of PathNet / PathNet_homo, /root/reference/PathNet_run.py:164,195,265, and the GRU ablation), built
// around one idea: BOTH operands of every k-step go through LDS and a workgroup owns 128 paths.
//
// Why (DESIGN.md §2 finding 1): in seq_fwd3 / seq_bwd3 (pn_pagg.hip) every wave streams its own
// weight fragments L2 -> VGPR, 512 B per MFMA, once per 32-path tile and step -- at the matrix pipe's peak that is the
// whole 64 B/clk vector-memory return path of a CU, and the kernels sat at 26-41 % of the pipe.  Here a k-step's
// weight fragments are copied once per 128 paths into LDS by LDS-DMA (global_load_lds_dwordx4, no registers, 16 B/clk
// per CU at the pipe's peak) and read from there by the eight waves (ds_read_b128: 256 B/clk per CU); a wave owns 64
// paths x 32 hidden units x 4 gates, so every weight fragment it reads feeds two MFMAs.  The activations of a k-step
// (the gathered x_t rows / h_{t-1} / the gate gradients) are split into their three bf16 planes by four lanes per path
// and double-buffered in LDS in fragment order, one barrier per k-step.  What no longer fits in LDS travels through
// L2: h_t goes to the next step through the [x | h] rows the weight-gradient GEMM needs anyway, the BPTT re-reads the
// gate gradients it has just written for that GEMM.
//
// Arithmetic is unchanged (pn_kernels.h): fp32 products as six bf16 MFMAs over exact three-plane splits, in the same
// order of planes as the fused kernels.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "pn_kernels.h"
#include "pn_seq.h"

using namespace pn;

// tuning builds only (tools/seq4_variants.sh): ablations of the forward kernel (results are wrong, times are the point) and
// s_memtime stamps of waves 0 and 4 of every workgroup at the phase boundaries of each k-step
// The forward and BPTT kernels of this file measured SLOWER than the fused ones of pn_pagg.hip (DESIGN.md section 2, round-3
// finding 1) and are not part of the shipped library: they build only with -DPN_EXPERIMENTAL=1 (tools/seq4_variants.sh,
// tests/test_gpu_seq4.py runs them when PN_LIB_PATH names such a build).  The weight-gradient GEMM below is the bf16 x 3
// mode's default at hidden size 128.
#ifndef PN_EXPERIMENTAL
#define PN_EXPERIMENTAL 0
#endif
#ifndef PN_F4_NODMA
#define PN_F4_NODMA 0
#endif
#ifndef PN_F4_NOPHILOX
#define PN_F4_NOPHILOX 0
#endif
#ifndef PN_F4_NOSTORE
#define PN_F4_NOSTORE 0
#endif
#ifndef PN_F4_NOMFMA
#define PN_F4_NOMFMA 0
#endif
#ifndef PN_F4_NOALOAD
#define PN_F4_NOALOAD 0
#endif
#ifndef PN_F4_NOCELL
#define PN_F4_NOCELL 0
#endif
#ifndef PN_TRACE4
#define PN_TRACE4 0
#endif
#ifndef PN_B4_NOFRAG
#define PN_B4_NOFRAG 0      // ablations of the BPTT kernel (tuning builds; results are wrong)
#endif
#ifndef PN_B4_NOMFMA
#define PN_B4_NOMFMA 0
#endif
#ifndef PN_B4_NOLOAD
#define PN_B4_NOLOAD 0
#endif
#ifndef PN_B4_NOCELL
#define PN_B4_NOCELL 0
#endif
#ifndef PN_B4_NOSCAT
#define PN_B4_NOSCAT 0
#endif
#ifndef PN_W4_NOSPLIT
#define PN_W4_NOSPLIT 0     // ablations of the weight-gradient GEMM (tuning builds; results are wrong)
#endif
#ifndef PN_W4_NOMFMA
#define PN_W4_NOMFMA 0
#endif
#ifndef PN_W4_NOLOAD
#define PN_W4_NOLOAD 0
#endif
#ifndef PN_W4_PRIO
#define PN_W4_PRIO 0        // 1: s_setprio 1 for waves 4..7 before the main loop
#endif
#ifndef PN_W4_NOFRAG
#define PN_W4_NOFRAG 0
#endif
#if PN_TRACE4
__device__ long long *g_trace4 = nullptr;       // [blocks][2][T4_SLOTS]
constexpr int T4_SLOTS = 512;
#define T4_STAMP(slot)                                                                                              \
    do {                                                                                                            \
        if (g_trace4 && (threadIdx.x & 255) == 0 && (slot) < T4_SLOTS)                                              \
            g_trace4[((size_t)blockIdx.x * 2 + (threadIdx.x >> 8)) * T4_SLOTS + (slot)] = (long long)__builtin_readcyclecounter(); \
    } while (0)
// (the weight-gradient GEMM: workgroup = blockIdx.z * gridDim.y + blockIdx.y, waves 0 and 4 stamp)
#define W4_STAMP(slot)                                                                                              \
    do {                                                                                                            \
        if (g_trace4 && (threadIdx.x & 255) == 0 && (slot) >= 0 && (slot) < T4_SLOTS)                               \
            g_trace4[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * 2 + (threadIdx.x >> 8)) * T4_SLOTS + (slot)] =  \
                (long long)__builtin_readcyclecounter();                                                            \
    } while (0)
#else
#define T4_STAMP(slot) do { } while (0)
#define W4_STAMP(slot) do { } while (0)
#endif

namespace {

[[maybe_unused]] constexpr int H4 = 128, G4 = 4, NW4 = H4 / 32;      // hidden size, gate slots, unit blocks of 32 hidden units
[[maybe_unused]] constexpr int MT4 = 128, NT4 = 512;                 // paths and threads per workgroup (8 waves: 2 row groups x 4 unit blocks)
[[maybe_unused]] constexpr int SV4 = 5;                              // saved values per (path, step, unit): i f g o c / r z n nh h_prev

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses to `lds_wave_base + lane * 16` (wave-uniform base)
__device__ __forceinline__ void dma16(const void *gsrc_lane, unsigned char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

#if PN_EXPERIMENTAL      // the 128-path forward and the 64 / 128-path BPTT: measured experiments, not product (see above)
// =====================================================================================================================
// forward recurrence
//   k-step s of step t: gates[128, 4H] += A[128, 16] . B[16, 4H];  A = columns 16s.. of [x_t | h_{t-1}], B = rows of
//   [W_ih | W_hh]^T.  Step 0 (h_{-1} = 0) has the KX x k-steps only.
//   LDS per stage: B 48 KB = fragment ((ub * 3 + plane) * 4 + gate) of 1 KB (lane l: 16 bytes), A 12 KB = fragment
//   (row block * 3 + plane); lane l of an A fragment sits in 16-byte slot l ^ (4 * (l >> 5)) (the four lanes that
//   produce a path's 16 columns then write conflict-free).  Two stages.
// =====================================================================================================================
constexpr int F_KS = 2 * H4 / 16, F_KX = H4 / 16;
constexpr int F_BSTAGE = NW4 * G4 * 3 * 1024;       // 49 152
constexpr int F_ASTAGE = (MT4 / 32) * 3 * 1024;     // 12 288
constexpr int F_A_OFF = 2 * F_BSTAGE;
constexpr int F_IDX_OFF = F_A_OFF + 2 * F_ASTAGE;   // 122 880: row indices [128][L], slots [128]

// Wp4[((s * NW + ub) * 3 + plane) * G + g][lane] (16 bytes) =
//     plane of Wcat[g*H + 32 ub + (lane & 31)][16 s + 8 (lane >> 5) .. +7]:  the stage image of k-step s is 48 KB contiguous
__global__ void pack_fwd4_kernel(const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                 const float *__restrict__ b_ih, const float *__restrict__ b_hh, int gru,
                                 u32x4 *__restrict__ Wp, float *__restrict__ biasc) {
    constexpr int H = H4, G = G4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < G * H) {
        if (!gru) {
            biasc[idx] = b_ih[idx] + b_hh[idx];
        } else {    // slots r, z, nx = W_in x + b_in, nh = W_hn h + b_hn (pn_pagg.hip: pack_fwd3_kernel)
            const int slot = idx / H, j = idx - slot * H, wr = (slot < 2 ? slot : 2) * H + j;
            biasc[idx] = slot < 2 ? b_ih[wr] + b_hh[wr] : slot == 2 ? b_ih[wr] : b_hh[wr];
        }
    }
    if (idx >= F_KS * NW4 * G * 64) return;
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int g = rest % G;
    rest /= G;
    const int ub = rest % NW4, s = rest / NW4;
    const int j = 32 * ub + (lane & 31), k = 16 * s + 8 * (lane >> 5);
    const int row = gru ? (g < 2 ? g : 2) * H + j : g * H + j;
    const float *src = k < H ? w_ih + (int64_t)row * H + k : w_hh + (int64_t)row * H + (k - H);
    float4 v0 = reinterpret_cast<const float4 *>(src)[0], v1 = reinterpret_cast<const float4 *>(src)[1];
    if (gru && ((g == 2 && k >= H) || (g == 3 && k < H))) v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
    u32x4 q0, q1, q2;
    uint32_t x0, x1, x2;
    split3(v0.x, v0.y, x0, x1, x2); q0[0] = x0; q1[0] = x1; q2[0] = x2;
    split3(v0.z, v0.w, x0, x1, x2); q0[1] = x0; q1[1] = x1; q2[1] = x2;
    split3(v1.x, v1.y, x0, x1, x2); q0[2] = x0; q1[2] = x1; q2[2] = x2;
    split3(v1.z, v1.w, x0, x1, x2); q0[3] = x0; q1[3] = x1; q2[3] = x2;
    u32x4 *dst = Wp + ((int64_t)((s * NW4 + ub) * 3) * G + g) * 64 + lane;
    dst[0] = q0;
    dst[G * 64] = q1;
    dst[2 * G * 64] = q2;
}

// GC: 4 = LSTM, 3 = GRU on the LSTM's four gate slots
template <int GC>
__global__ __launch_bounds__(NT4, 2) void seq_fwd4_kernel(SeqFwdParams p) {
    constexpr bool GRU = GC == 3;
    constexpr int H = H4, G = G4, SV = SV4, KS = F_KS, KX = F_KX;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    int *s_rowidx = reinterpret_cast<int *>(ldsb + F_IDX_OFF);     // [MT][L] gather rows of this tile
    int *s_slotof = s_rowidx + MT4 * p.L;                           // [MT]
    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave_u >> 2, ub = wave_u & 3;                    // rows 64 rg .. +63, hidden units 32 ub .. +31 (x 4 gates)
    const int q0 = blockIdx.x * MT4;
    const int L = p.L;

    for (int i = tid; i < MT4 * L; i += NT4) s_rowidx[i] = q0 + i / L < p.P ? p.rowidx[(int64_t)q0 * L + i] : 0;
    for (int i = tid; i < MT4; i += NT4) s_slotof[i] = q0 + i < p.P ? p.slotof[q0 + i] : 0;

    const float keep_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (1.0f - p.p_drop))));
    const bool builtin_drop = !p.mask && p.p_drop > 0.0f;
    const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
    // 64-bit tile bases (wave-uniform) + 32-bit offsets inside the tile
    const size_t tile_row = (size_t)q0 * (size_t)L;
    const int rows_here = min(MT4, p.P - q0);               // >= 1
    uint8_t *keep_t = p.keep ? p.keep + tile_row * (H / 4) : nullptr;
    float *xh_t = p.xh + tile_row * (2 * H);
    float *saved_t = p.saved ? p.saved + tile_row * (SV * H) : nullptr;
    float *hn_t = p.hn + (size_t)q0 * H;
    const unsigned char *wp_b = reinterpret_cast<const unsigned char *>(p.Wp);

    // ---- the producer of the A stage: thread (path pr, quarter pp) owns columns 4 pp .. +3 of a k-step's 16 ---------
    const int pr = tid >> 2, pp = tid & 3;
    const bool row_ok = pr < rows_here;
    const uint32_t prc = (uint32_t)min(pr, rows_here - 1);
    // (row block, plane 0) fragment, swizzled slot of lane (pr & 31) + 32 (pp >> 1), low / high half of its 16 bytes
    const uint32_t a_wr = (uint32_t)((pr >> 5) * 3 * 1024 + ((((pr & 31) + 32 * (pp >> 1)) ^ ((pp >> 1) << 2)) << 4) + (pp & 1) * 8);
    f32x4 xr;                   // raw 16 bytes of the block in flight
    uint32_t kb_next = 0;       // keep bits drawn for the block in flight

    // k-step j of the kernel = (step, k-step of the step): step 0 has the KX x k-steps only
    const int NK = KX + (L - 1) * KS;
    auto step_of = [&](int j) { return j < KX ? 0 : 1 + (j - KX) / KS; };
    auto ks_of = [&](int j) { return j < KX ? j : (j - KX) % KS; };

    auto dma_stage = [&](int s, int stage) {        // this wave's eighth of k-step s's weight fragments
        const unsigned char *g = wp_b + (size_t)s * F_BSTAGE + wave_u * (F_BSTAGE / 8) + (threadIdx.x & 63) * 16;
        unsigned char *l = ldsb + stage * F_BSTAGE + wave_u * (F_BSTAGE / 8);
#pragma unroll
        for (int i = 0; i < F_BSTAGE / 8 / 1024; i++) dma16(g + i * 1024, l + i * 1024);
    };
    // (unconditional, the address alone depends on the block: no branch may sit between an asm load and its wait)
    auto a_issue = [&](int t, int s) {
        const float *px = p.Z + ((size_t)(uint32_t)s_rowidx[pr * L + t] * H + 16 * s + 4 * pp);
        const float *ph = xh_t + ((size_t)(prc * (uint32_t)L + t) * (2 * H) + H + 16 * (s - KX) + 4 * pp);
        if (PN_F4_NOALOAD)
            xr = f32x4{0.f, 0.f, 0.f, 0.f};
        else
            async_load_b128(xr, s < KX ? px : ph);
    };
    auto draw_keep = [&](int t, int s) {
        const float4 m = dropout4(seed, ((uint64_t)t * p.Pmask + s_slotof[pr]) * (H / 4) + 4 * s + pp, 1u, p.p_drop);
        uint32_t bits = (m.x != 0.f ? 1u : 0u) | (m.y != 0.f ? 2u : 0u) | (m.z != 0.f ? 4u : 0u) | (m.w != 0.f ? 8u : 0u);
        asm volatile("" : "+v"(bits));      // drawn here, not sunk to the commit
        kb_next = bits;
    };
    // block (t, s) has arrived (the caller has waited): mask, split into planes, write to `stage`; x blocks also go to
    // the [x | h] rows and their keep bits to the keep bytes (what the backward needs)
    auto a_commit = [&](int t, int s, int stage) {
        float4 v = make_float4(xr[0], xr[1], xr[2], xr[3]);
        if (!row_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < KX) {
            if (p.mask) {
                if (row_ok) {
                    const float4 m = reinterpret_cast<const float4 *>(
                        p.mask)[((int64_t)t * p.Pmask + s_slotof[pr]) * (H / 4) + 4 * s + pp];
                    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
                }
            } else if (builtin_drop) {
                const uint32_t b = kb_next;
                v.x = b & 1u ? v.x * keep_scale : 0.0f;
                v.y = b & 2u ? v.y * keep_scale : 0.0f;
                v.z = b & 4u ? v.z * keep_scale : 0.0f;
                v.w = b & 8u ? v.w * keep_scale : 0.0f;
            }
        }
        uint32_t a0, a1, a2, b0, b1, b2;
        split3(v.x, v.y, a0, a1, a2);
        split3(v.z, v.w, b0, b1, b2);
        unsigned char *d = ldsb + F_A_OFF + stage * F_ASTAGE + a_wr;
        *reinterpret_cast<uint2 *>(d) = make_uint2(a0, b0);
        *reinterpret_cast<uint2 *>(d + 1024) = make_uint2(a1, b1);
        *reinterpret_cast<uint2 *>(d + 2048) = make_uint2(a2, b2);
        if (s < KX && row_ok && !PN_F4_NOSTORE) {
            const uint32_t c4 = 4 * s + pp;
            if (p.store_x) {
                float4 *xo = &at_bytes(reinterpret_cast<float4 *>(xh_t), (((uint32_t)pr * (uint32_t)L + t) * (uint32_t)(2 * H / 4) + c4) * 16u);
                xo[0] = v;
                if (t == 0) xo[H / 4] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (keep_t && builtin_drop) keep_t[((uint32_t)pr * (uint32_t)L + t) * (uint32_t)(H / 4) + c4] = (uint8_t)(kb_next & 15u);
        }
    };
    // a slot boundary: nothing is scheduled across it
    auto slot_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto wait_vmem = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto wait_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

    f32x16 cst[2], acc[2][G];
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int r = 0; r < 16; r++) cst[rb][r] = 0.0f;
    auto acc_init = [&]() {
        const int col = 32 * ub + (fresh_lane() & 31);
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float bias = p.biasc[g * H + col];
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[rb][g][r] = bias;
        }
    };
    // ---- cell update in registers; h_t leaves for the next step through the [x | h] rows (L2) -------------------------
    auto cell_update = [&](int t) {
        const int lane_o = fresh_lane();
        const int col_o = 32 * ub + (lane_o & 31);
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = 64 * rg + 32 * rb + acc_row(r, lane_o);
                const bool ok = row < rows_here;
                float h;
                if (GRU) {
                    // saved: r, z, n, the pre-activation W_hn h + b_hn, h_{t-1}
                    const float rgt = sigmoidf_(acc[rb][0][r]);
                    const float zg = sigmoidf_(acc[rb][1][r]);
                    const float nh = acc[rb][3][r];
                    const float ng = tanhf_(acc[rb][2][r] + rgt * nh);
                    const float hp = cst[rb][r];
                    h = (1.0f - zg) * ng + zg * hp;
                    cst[rb][r] = h;
                    if (saved_t && ok && !PN_F4_NOSTORE) {
                        float *sv = &at_bytes(saved_t, (((uint32_t)row * (uint32_t)L + t) * (uint32_t)(SV * H) + col_o) * 4u);
                        sv[0] = rgt; sv[H] = zg; sv[2 * H] = ng; sv[3 * H] = nh; sv[4 * H] = hp;
                    }
                } else {
                    const float ig = sigmoidf_(acc[rb][0][r]);
                    const float fg = sigmoidf_(acc[rb][1][r]);
                    const float gg = tanhf_(acc[rb][2][r]);
                    const float og = sigmoidf_(acc[rb][3][r]);
                    const float c = fg * cst[rb][r] + ig * gg;
                    cst[rb][r] = c;
                    h = og * tanhf_(c);
                    if (saved_t && ok && !PN_F4_NOSTORE) {
                        float *sv = &at_bytes(saved_t, (((uint32_t)row * (uint32_t)L + t) * (uint32_t)(SV * H) + col_o) * 4u);
                        sv[0] = ig; sv[H] = fg; sv[2 * H] = gg; sv[3 * H] = og; sv[4 * H] = c;
                    }
                }
                if (ok) {
                    if (t == L - 1)
                        at_bytes(hn_t, ((uint32_t)row * (uint32_t)H + col_o) * 4u) = h;
                    else
                        at_bytes(xh_t, (((uint32_t)row * (uint32_t)L + t + 1) * (uint32_t)(2 * H) + H + col_o) * 4u) = h;
                }
            }
        }
        acc_init();
    };

    // fragments: fa = the A planes of this wave's two row blocks, fb / fb2 = B planes (four gates each)
    u32x4 fa[2][3], fb[G], fb2[G];
    auto read_c0 = [&](int stage) {          // what the first cluster of a k-step needs: all A planes, B plane 0
        const int lane_k = fresh_lane();
        const unsigned char *ab = ldsb + F_A_OFF + stage * F_ASTAGE + rg * (2 * 3 * 1024) + ((lane_k ^ ((lane_k >> 5) << 2)) << 4);
        const unsigned char *bb = ldsb + stage * F_BSTAGE + ub * (3 * G * 1024) + (lane_k << 4);
#pragma unroll
        for (int g = 0; g < G; g++) fb[g] = *reinterpret_cast<const u32x4 *>(bb + g * 1024);
#pragma unroll
        for (int pl = 2; pl >= 0; pl--)
#pragma unroll
            for (int rb = 0; rb < 2; rb++) fa[rb][pl] = *reinterpret_cast<const u32x4 *>(ab + (rb * 3 + pl) * 1024);
    };
    auto read_c1 = [&](int stage) {          // the second cluster: B planes 1 and 2 (A planes 0 and 1 stay)
        const int lane_k = fresh_lane();
        const unsigned char *bb = ldsb + stage * F_BSTAGE + ub * (3 * G * 1024) + (lane_k << 4);
#pragma unroll
        for (int g = 0; g < G; g++) fb[g] = *reinterpret_cast<const u32x4 *>(bb + (G + g) * 1024);
#pragma unroll
        for (int g = 0; g < G; g++) fb2[g] = *reinterpret_cast<const u32x4 *>(bb + (2 * G + g) * 1024);
    };
    auto prod = [&](int pa, u32x4 (&b)[G]) {
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int g = 0; g < G; g++) acc[rb][g] = mfma_bf16(fa[rb][pa], b[g], acc[rb][g]);
    };

    __syncthreads();        // the index arrays

    // ---- prologue: stages 0 and 1 <- k-steps 0 and 1 -------------------------------------------------------------------
    dma_stage(0, 0);
    dma_stage(1, 1);
    if (builtin_drop) draw_keep(0, 0);
    a_issue(0, 0);
    wait_vmem();
    a_commit(0, 0, 0);
    if (builtin_drop) draw_keep(0, 1);
    a_issue(0, 1);          // (KX >= 2: k-step 1 is an x k-step of step 0)
    wait_vmem();
    a_commit(0, 1, 1);
    acc_init();
    wait_lds();
    __syncthreads();
    read_c0(0);
    // The two row groups (the two waves of every SIMD) run the slots of a k-step one slot apart: while one group feeds
    // the matrix pipe (24 MFMAs), the other reads its next fragments, commits its activations and issues its copies;
    // s_barrier between the slots keeps them in this order.  Per group and k-step j:
    //   M0: products a2.b0 a1.b0 a0.b0   |  R1: fragments of the second cluster, load of block j+2, its keep bits
    //   M1: products a1.b1 a0.b1 a0.b2   |  R0: fragments of k-step j+1's first cluster, commit of block j+2, copy of
    //                                           weight stage j+2 (both into the buffer k-step j has just left)
    // vmcnt(0) stands at the head of a memory slot, before anything new is issued: what it waits for was issued two
    // slots earlier.
    if (rg == 1) slot_barrier();
#pragma unroll 1
    for (int j = 0; j < NK; j++) {
        const int t = step_of(j);
        const bool step_end = j + 1 == NK || step_of(j + 1) != t;
        const int j2 = min(j + 2, NK - 1), t2 = step_of(j2), s2 = ks_of(j2);
        T4_STAMP(8 + 5 * j + 0);
        // ---- M0 ----
        if (!PN_F4_NOMFMA) {
            __builtin_amdgcn_s_setprio(1);
            prod(2, fb);
            prod(1, fb);
            prod(0, fb);
            __builtin_amdgcn_s_setprio(0);
        }
        T4_STAMP(8 + 5 * j + 1);
        slot_barrier();
        // ---- R1 ----
        wait_vmem();
        read_c1(j & 1);
        a_issue(t2, s2);
        if (j + 2 < NK && builtin_drop && s2 < KX && !PN_F4_NOPHILOX) draw_keep(t2, s2);
        T4_STAMP(8 + 5 * j + 2);
        slot_barrier();
        // ---- M1 ----
        if (!PN_F4_NOMFMA) {
            __builtin_amdgcn_s_setprio(1);
            prod(1, fb);
            prod(0, fb);
            prod(0, fb2);
            __builtin_amdgcn_s_setprio(0);
        }
        T4_STAMP(8 + 5 * j + 3);
        slot_barrier();
        // ---- R0 (at the end of a step both groups update their cells at the same time: group 1 before this slot's
        //      work, group 0 after the barrier that ends it) ----
        if (step_end && rg == 1) cell_update(t);
        wait_vmem();
        if (j + 1 < NK) read_c0((j + 1) & 1);
        if (j + 2 < NK) {
            a_commit(t2, s2, j & 1);
            if (!PN_F4_NODMA) dma_stage(s2, j & 1);
        }
        wait_lds();
        T4_STAMP(8 + 5 * j + 4);
        slot_barrier();
        if (step_end && rg == 0) cell_update(t);
    }
    if (rg == 0) slot_barrier();
}


// =====================================================================================================================
// BPTT:  [dx_t | dh_{t-1}] [128, 2H] = dG_t [128, 4H] . [W_ih | W_hh],  K = 4H gate columns in stages of 32
//   Per step: (1) cell backward, element-wise in accumulator layout -- reads the saved gates, writes the gate gradients
//   dG_t (the weight-gradient GEMM's operand) to HBM; (2) the GEMM, whose A stages re-read dG_t from L2 (four lanes per
//   path: 32 bytes each = one fragment lane of 8 k, split into planes, one ds_write_b128 per plane); (3) the gather
//   backward scatter of dx.  A wave owns 64 paths x 32 columns of dx and the same 32 columns of dh.
//   LDS per stage: B 48 KB = fragment (((half * 2 + kk) * 4 + ub) * 3 + plane), half 0 = dx columns, 1 = dh columns
//   (step 0 needs dx only and copies the first 24 KB); A 24 KB = fragment ((kk * 4 + row block) * 3 + plane), lane l of
//   k-step kk in slot l ^ (2 (l >> 5)) ^ (4 kk).
// =====================================================================================================================
constexpr int B_NS = G4 * H4 / 32;                          // stages per step
constexpr int B_BSTAGE = 2 * 2 * NW4 * 3 * 1024;            // 49 152

// WpT4[(((u * 2 + half) * 2 + kk) * NW + ub) * 3 + plane][lane] (16 bytes) =
//     plane of Wcat[k = 32 u + 16 kk + 8 (lane >> 5) .. +7][n = half * H + 32 ub + (lane & 31)]
__global__ void pack_bwd4_kernel(const float *__restrict__ w_ih, const float *__restrict__ w_hh, int gru,
                                 u32x4 *__restrict__ WpT) {
    constexpr int H = H4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B_NS * 2 * 2 * NW4 * 64) return;
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int ub = rest % NW4;
    rest /= NW4;
    const int kk = rest & 1, half = (rest >> 1) & 1, u = rest >> 2;
    const int k = 32 * u + 16 * kk + 8 * (lane >> 5), n = 32 * ub + (lane & 31);
    float v[8];
    if (!gru) {
        const float *src = (half == 0 ? w_ih : w_hh) + (int64_t)k * H + n;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = src[(int64_t)e * H];
    } else {        // k .. k+7 lie inside one gate slot (8 | H)
        const int slot = k / H, j = k - slot * H;
        const bool zero = (slot == 2 && half == 1) || (slot == 3 && half == 0);
        const float *src = (half == 0 ? w_ih : w_hh) + (int64_t)((slot < 2 ? slot : 2) * H + j) * H + n;
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
    u32x4 *dst = WpT + ((int64_t)((((u * 2 + half) * 2 + kk) * NW4 + ub) * 3)) * 64 + lane;
    dst[0] = q0;
    dst[64] = q1;
    dst[128] = q2;
}

// NRG: row groups of 64 paths per workgroup (2: 128 paths, 8 waves, one workgroup per CU, the two groups share every weight
//      fragment in LDS; 1: 64 paths, 4 waves, two workgroups per CU whose phases overlap).  KK: k-steps per stage.
template <int GC, int NRG, int KK>
__global__ __launch_bounds__(256 * NRG, 2) void seq_bwd4_kernel(SeqBwdParams p) {
    constexpr bool GRU = GC == 3;
    constexpr int H = H4, G = G4, GH = G * H, SV = SV4;
    constexpr int MT = 64 * NRG, NT = 256 * NRG;
    constexpr int BST = KK * 2 * NW4 * 3 * 1024;            // weight stage: [half][kk][ub][plane] fragments
    constexpr int AST = KK * (MT / 32) * 3 * 1024;          // activation stage: [kk][row block][plane]
    constexpr int A_OFF = 2 * BST, IDX_OFF = A_OFF + 2 * AST;
    constexpr int NS = GH / (16 * KK);                      // stages per step
    static_assert((NRG == 2 && KK == 2) || (NRG == 1 && KK == 1), "three 1 KB loads per wave and half of a stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int L = p.L;
    int *s_rowidx = reinterpret_cast<int *>(ldsb + IDX_OFF);     // [MT][L]
    int *s_slotof = s_rowidx + MT * L;                             // [MT]
    uint8_t *s_keep = reinterpret_cast<uint8_t *>(s_slotof + MT);  // [2][MT][H/4] keep bits of step t (t & 1)
    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave_u >> 2, ub = wave_u & 3;
    // tiles in descending order: the forward wrote the saved tensors of the last tiles last (Infinity Cache)
    const int q0 = (int)(gridDim.x - 1 - blockIdx.x) * MT;

    for (int i = tid; i < MT * L; i += NT) s_rowidx[i] = q0 + i / L < p.P ? p.rowidx[(int64_t)q0 * L + i] : 0;
    for (int i = tid; i < MT; i += NT) s_slotof[i] = q0 + i < p.P ? p.slotof[q0 + i] : 0;

    const float keep_scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / (1.0f - p.p_drop))));
    const size_t tile_row = (size_t)q0 * (size_t)L;
    const float *saved_t = p.saved + tile_row * (SV * H);
    float *dG_t = p.dG + tile_row * GH;
    const uint8_t *keep_t = p.keep ? p.keep + tile_row * (H / 4) : nullptr;
    const float *dhn_t = p.dhn + (size_t)q0 * H;
    const int rows_here = min(MT, p.P - q0);            // >= 1
    const unsigned char *wp_b = reinterpret_cast<const unsigned char *>(p.WpT);

    f32x16 accx[2], acch[2];            // dx_t, dh_{t-1} of the step in the GEMM (accumulator layout)
    // The element-wise phases run in a second layout of the wave's 64 paths x 32 hidden units: lane (row8 = l >> 3,
    // chunk = l & 7) holds units 4 chunk .. +3 of the rows row8 + 8 i (i = 0..3) of each row block -- 16 bytes per lane
    // and whole 128-byte lines per 8 lanes for every global access (a wave instruction costs the CU's vector memory
    // pipe ~16 cycles whatever its width: one unit per lane, the accumulator layout, quadruples the instructions).
    // dh_t crosses from the accumulator layout through a wave-private LDS block.
    f32x4 dhq[2][4], dcq[2][4];         // d h_t, d c_t (GRU: the direct path d h_t / d h_{t-1})
    const int q_row8 = (tid & 63) >> 3, q_col = 32 * ub + 4 * (tid & 7);
    {
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = 64 * rg + 32 * rb + q_row8 + 8 * i;
                const f32x4 dh0 = *reinterpret_cast<const f32x4 *>(
                    &at_bytes(dhn_t, ((uint32_t)min(row, rows_here - 1) * (uint32_t)H + q_col) * 4u));
                dhq[rb][i] = row < rows_here ? dh0 : f32x4{0.f, 0.f, 0.f, 0.f};
                dcq[rb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    constexpr int XP = 36;              // row pitch (floats) of the 32 x 32 crossing block
    // accumulator layout -> the quad layout, through this wave's block at `scr` (the caller has made sure nobody reads
    // the stage buffers it lies in)
    auto cross = [&](const f32x16 &v, float *scr, f32x4 (&out)[4]) {
        const int lane = fresh_lane();
#pragma unroll
        for (int r = 0; r < 16; r++) scr[acc_row(r, lane) * XP + (lane & 31)] = v[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = *reinterpret_cast<const f32x4 *>(scr + ((lane >> 3) + 8 * i) * XP + 4 * (lane & 7));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // ---- the producer of the A stage: thread (path pr, lane quarter pp) owns k = 8 pp .. +7 of a stage's 32 -------------
    const int pr = tid >> 2, pp = tid & 3;
    const bool row_ok = pr < rows_here;
    const uint32_t prc = (uint32_t)min(pr, rows_here - 1);
    // (KK = 1: 16 columns per stage, two of a path's four lanes produce them)
    const int kkp = KK == 2 ? pp >> 1 : 0;
    const bool a_active = KK == 2 || pp < 2;
    const uint32_t a_wr = (uint32_t)(((kkp * (MT / 32) + (pr >> 5)) * 3) * 1024 +
                                     ((((pr & 31) + 32 * (pp & 1)) ^ ((pp & 1) << 1) ^ (kkp << 2)) << 4));
    f32x4 xr0, xr1;
    // this wave's eighth of a stage's weight fragments, global -> registers -> LDS.  (Plain 16-byte loads: an LDS-DMA
    // instruction of the same 1 KB costs the CU's vector-memory pipe ~60 cycles, ~4x a plain load -- profiles/README.md)
    f32x4 braw[6];
#pragma unroll
    for (int i = 0; i < 6; i++) braw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // stage v: its first k-step is s0 = v * KK; half h of it sits at stage image (s0 >> 1), piece (h * 2 + (s0 & 1)) of 12 KB
    // (KK = 2: the two k-steps of a half are adjacent, 24 KB); every wave moves 3 KB of each half
    auto b_issue = [&](int v, auto both_tag) {
        constexpr bool both = decltype(both_tag)::value;
        const int s0 = v * KK;
        const unsigned char *g = wp_b + (size_t)(s0 >> 1) * B_BSTAGE + (s0 & 1) * (NW4 * 3 * 1024) + wave_u * 3072 + (threadIdx.x & 63) * 16;
#pragma unroll
        for (int i = 0; i < 3; i++) async_load_b128(braw[i], g + i * 1024);
        if constexpr (both) {
#pragma unroll
            for (int i = 0; i < 3; i++) async_load_b128(braw[3 + i], g + B_BSTAGE / 2 + i * 1024);
        }
    };
    auto b_commit = [&](int stage, auto both_tag) {
        constexpr bool both = decltype(both_tag)::value;
        unsigned char *l = ldsb + stage * BST + wave_u * 3072 + (threadIdx.x & 63) * 16;
#pragma unroll
        for (int i = 0; i < 3; i++) *reinterpret_cast<f32x4 *>(l + i * 1024) = braw[i];
        if constexpr (both) {
#pragma unroll
            for (int i = 0; i < 3; i++) *reinterpret_cast<f32x4 *>(l + BST / 2 + i * 1024) = braw[3 + i];
        }
    };
    auto a_issue = [&](int t, int u) {
        const float *src = dG_t + ((size_t)(prc * (uint32_t)L + t) * GH + 16 * KK * u + 8 * (KK == 2 ? pp : (pp & 1)));
        async_load_b128(xr0, src);
        async_load_b128(xr1, src + 4);
    };
    auto a_commit = [&](int stage, bool write) {
        if (!write || !a_active) return;
        float v[8] = {xr0[0], xr0[1], xr0[2], xr0[3], xr1[0], xr1[1], xr1[2], xr1[3]};
        u32x4 q0v, q1v, q2v;
#pragma unroll
        for (int h = 0; h < 4; h++) {
            uint32_t x0, x1, x2;
            split3(row_ok ? v[2 * h] : 0.0f, row_ok ? v[2 * h + 1] : 0.0f, x0, x1, x2);
            q0v[h] = x0; q1v[h] = x1; q2v[h] = x2;
        }
        unsigned char *d = ldsb + A_OFF + stage * AST + a_wr;
        *reinterpret_cast<u32x4 *>(d) = q0v;
        *reinterpret_cast<u32x4 *>(d + 1024) = q1v;
        *reinterpret_cast<u32x4 *>(d + 2048) = q2v;
    };
    __syncthreads();        // the index arrays
#pragma unroll 1
    for (int t = L - 1; t >= 0; t--) {
        // the first weight stage of this step does not depend on anything: in flight under the cell backward
        [[maybe_unused]] const int ti = L - 1 - t;
        T4_STAMP(8 * ti + 0);
        const int lane_t = fresh_lane();
        if (p.keep) {       // this step's keep bytes (MT rows x H/4) -> LDS, read by the scatter phase below
            const int tid_t = wave_u * 64 + lane_t;
            for (int i = tid_t; i < MT * (H / 16); i += NT) {
                const int row = i / (H / 16), w = i - row * (H / 16);
                const uint32_t rc = (uint32_t)min(row, rows_here - 1);
                reinterpret_cast<uint32_t *>(s_keep + (t & 1) * MT * (H / 4))[i] =
                    at_bytes(reinterpret_cast<const uint32_t *>(keep_t), (rc * (uint32_t)L + t) * (uint32_t)(H / 4) + 4u * w);
            }
        }
        // ---- cell backward in the quad layout: the 24 loads of a row block are issued together (unconditionally: padded
        //      rows read a clamped row and are zeroed afterwards) -- one memory round trip per row block
#pragma unroll
        for (int rb = 0; rb < (PN_B4_NOCELL ? 0 : 2); rb++) {
            f32x4 vi[4], vf[4], vg[4], vo[4], vc[4], vn[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int rc = min(64 * rg + 32 * rb + q_row8 + 8 * i, rows_here - 1);
                const f32x4 *sv = reinterpret_cast<const f32x4 *>(
                    &at_bytes(saved_t, (((uint32_t)rc * (uint32_t)L + t) * (uint32_t)(SV * H) + q_col) * 4u));
                vi[i] = sv[0]; vf[i] = sv[H / 4]; vg[i] = sv[2 * H / 4];
                vo[i] = sv[3 * H / 4];
                if (GRU) {
                    vc[i] = sv[4 * H / 4];                                              // h_{t-1}
                    vn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    vc[i] = t > 0 ? sv[-(H / 4)] : f32x4{0.f, 0.f, 0.f, 0.f};           // c_{t-1} = slot 4 of step t-1
                    vn[i] = sv[4 * H / 4];                                              // c_t
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = 64 * rg + 32 * rb + q_row8 + 8 * i;
                const bool ok = row < rows_here;
                f32x4 *d = reinterpret_cast<f32x4 *>(
                    &at_bytes(dG_t, (((uint32_t)min(row, rows_here - 1) * (uint32_t)L + t) * (uint32_t)GH + q_col) * 4u));
                f32x4 a0, a1, a2, a3;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float dhv = dhq[rb][i][e];
                    if (GRU) {
                        // h = (1 - z) n + z h_prev,  n = tanh(nx + r nh): gradients of the slots r, z, nx, nh; the direct path
                        // d h_t / d h_{t-1} = z is carried in dcq across the GEMM and added to its dh output
                        const float rgt = vi[i][e], zg = vf[i][e], ng = vg[i][e], nh = vo[i][e], hp = vc[i][e];
                        const float dnp = dhv * (1.0f - zg) * (1.0f - ng * ng);
                        a0[e] = dnp * nh * rgt * (1.0f - rgt);
                        a1[e] = dhv * (hp - ng) * zg * (1.0f - zg);
                        a2[e] = dnp;
                        a3[e] = dnp * rgt;
                        dcq[rb][i][e] = ok ? dhv * zg : 0.0f;
                    } else {
                        const float ig = vi[i][e], fg = vf[i][e], gg = vg[i][e], og = vo[i][e], cprev = vc[i][e];
                        const float tc = tanhf_(vn[i][e]);
                        const float d_o = dhv * tc;
                        const float dct = dcq[rb][i][e] + dhv * og * (1.0f - tc * tc);
                        a0[e] = dct * gg * ig * (1.0f - ig);
                        a1[e] = dct * cprev * fg * (1.0f - fg);
                        a2[e] = dct * ig * (1.0f - gg * gg);
                        a3[e] = d_o * og * (1.0f - og);
                        dcq[rb][i][e] = dct * fg;
                    }
                }
                if (ok) {
                    d[0] = a0; d[H / 4] = a1; d[2 * H / 4] = a2; d[3 * H / 4] = a3;
                }
            }
        }
        T4_STAMP(8 * ti + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the gate gradients have left this wave
        T4_STAMP(8 * ti + 2);
        __syncthreads();

        // ---- the GEMM over NS stages --------------------------------------------------------------------------------
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int r = 0; r < 16; r++) accx[rb][r] = acch[rb][r] = 0.0f;
        T4_STAMP(8 * ti + 3);
        auto gemm = [&](auto ntn_tag) {
            constexpr int NTN = decltype(ntn_tag)::value;       // 2: dx and dh, 1: dx only (step 0)
            using Both = std::integral_constant<bool, NTN == 2>;
            // stage u (clamped) into flight / out of flight / into LDS buffer u & 1
            auto issue = [&](int u) {
                if (PN_B4_NOLOAD) return;
                const int uc = min(u, NS - 1);
                b_issue(uc, Both{});
                a_issue(t, uc);
            };
            auto wait_all = [&]() { wait_vm<0>(braw[0], braw[1], braw[2], braw[3], braw[4], braw[5], xr0, xr1); };
            auto commit = [&](int u) {
                if (u < NS && !PN_B4_NOLOAD) {
                    b_commit(u & 1, Both{});
                    a_commit(u & 1, true);
                }
            };
            // the 48 (24 at step 0) MFMAs of stage u
            auto products = [&](int u) {
                const int lane_k = fresh_lane();
                const unsigned char *ab = ldsb + A_OFF + (u & 1) * AST + rg * (2 * 3 * 1024);
                const uint32_t slot0 = (uint32_t)(lane_k ^ ((lane_k >> 5) << 1));
                const unsigned char *bb = ldsb + (u & 1) * BST + ub * (3 * 1024) + (lane_k << 4);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < KK; kk++) {
                    const unsigned char *abk = ab + kk * ((MT / 32) * 3 * 1024) + ((slot0 ^ (uint32_t)(kk << 2)) << 4);
                    auto afrag = [&](int rb, int pl) { return *reinterpret_cast<const u32x4 *>(abk + (rb * 3 + pl) * 1024); };
                    auto bfrag = [&](int half, int pl) {
                        return *reinterpret_cast<const u32x4 *>(bb + ((half * KK + kk) * NW4 * 3 + pl) * 1024);
                    };
                    u32x4 a[2][3], bx[3], bh[3];
                    auto prod = [&](int pa, int pb) {
                        if (PN_B4_NOMFMA) return;
#pragma unroll
                        for (int rb = 0; rb < 2; rb++) {
                            accx[rb] = mfma_bf16(a[rb][pa], bx[pb], accx[rb]);
                            if constexpr (NTN == 2) acch[rb] = mfma_bf16(a[rb][pa], bh[pb], acch[rb]);
                        }
                    };
                    if (PN_B4_NOFRAG) {     // operands from registers: no LDS traffic
#pragma unroll
                        for (int pl = 0; pl < 3; pl++) {
                            bx[pl] = u32x4{(uint32_t)lane_k, 1u, 2u, 3u};
                            bh[pl] = u32x4{(uint32_t)lane_k, 5u, 2u, 3u};
#pragma unroll
                            for (int rb = 0; rb < 2; rb++) a[rb][pl] = u32x4{(uint32_t)lane_k, 7u, (uint32_t)rb, 3u};
                        }
                    } else {
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) {
                        bx[pl] = bfrag(0, pl);
                        if constexpr (NTN == 2) bh[pl] = bfrag(1, pl);
                    }
#pragma unroll
                    for (int pl = 0; pl < 3; pl++)
#pragma unroll
                        for (int rb = 0; rb < 2; rb++) a[rb][pl] = afrag(rb, pl);
                    }
                    prod(2, 0);
                    prod(1, 0);
                    prod(0, 0);
                    prod(1, 1);
                    prod(0, 1);
                    prod(0, 2);
                }
                __builtin_amdgcn_s_setprio(0);
            };
            // lockstep, products first: the loads of stage u+2 are issued behind the products of stage u and stay in
            // flight across the barrier: [products u][wait, commit u+1][issue u+2][barrier]
            issue(0);
            wait_all();
            commit(0);
            issue(1);
            __syncthreads();
#pragma unroll 1
            for (int u = 0; u < NS; u++) {
                T4_STAMP(64 + (ti * NS + u) * 4 + 0);
                products(u);
                T4_STAMP(64 + (ti * NS + u) * 4 + 1);
                wait_all();
                commit(u + 1);
                issue(u + 2);
                T4_STAMP(64 + (ti * NS + u) * 4 + 2);
                __syncthreads();
                T4_STAMP(64 + (ti * NS + u) * 4 + 3);
            }
            wait_all();
        };
        if (t > 0)
            gemm(std::integral_constant<int, 2>{});
        else
            gemm(std::integral_constant<int, 1>{});

        // ---- gather backward: dZ[row(q, t)] += mask * dx.  Step 0 is the last one and its rows are the paths' own start
        //      nodes: the W paths of a node add to the same table row.  There the wave parks a 32 x 32 block in the (now
        //      dead) stage region and each half-wave walks 16 rows in order, adding up runs of equal table rows: one
        //      atomic per run and column instead of one per path.
        T4_STAMP(8 * ti + 4);
        // d h_{t-1} into the quad layout (the k loop ended with a barrier: the stage buffers are free; the scatter's
        // scratch below uses the same block afterwards -- both are wave-private)
        if (t > 0) {
            float *xs = reinterpret_cast<float *>(ldsb) + wave_u * (32 * XP);
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                cross(acch[rb], xs, dhq[rb]);
                if (GRU) {
#pragma unroll
                    for (int i = 0; i < 4; i++) dhq[rb][i] += dcq[rb][i];
                }
            }
        }
        const int lane_s = fresh_lane(), li_s = lane_s & 31;
        const int col_s = 32 * ub + li_s;
        if (PN_B4_NOSCAT) {
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int r = 0; r < 16; r++) asm volatile("" ::"v"(accx[rb][r]));
        } else if (t == 0 && p.merge0) {
            float *scr = reinterpret_cast<float *>(ldsb) + wave_u * (32 * XP);
#pragma unroll 1
            for (int rb = 0; rb < 2; rb++) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rl = acc_row(r, lane_s), row = 64 * rg + 32 * rb + rl;
                    float dx = rb == 0 ? accx[0][r] : accx[1][r];
                    if (row < rows_here) {
                        if (p.mask)
                            dx *= p.mask[((uint64_t)t * p.Pmask + s_slotof[row]) * H + col_s];
                        else if (p.keep)
                            dx = (s_keep[((t & 1) * MT + row) * (H / 4) + (col_s >> 2)] >> (col_s & 3)) & 1 ? dx * keep_scale : 0.0f;
                    }
                    scr[rl * 33 + li_s] = dx;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int hk = lane_s >> 5;
                int curid = -1;
                float run = 0.0f;
#pragma unroll 1
                for (int i = 0; i < 16; i++) {
                    const int rl = 16 * hk + i, row = 64 * rg + 32 * rb + rl;
                    const int rid = row < rows_here ? s_rowidx[row * L] : -1;       // (uniform over a half-wave)
                    if (rid != curid) {
                        if (curid >= 0) atomicAdd(p.dZ + ((size_t)(uint32_t)curid * (uint32_t)H + col_s), run);
                        curid = rid;
                        run = 0.0f;
                    }
                    run += scr[rl * 33 + li_s];
                }
                if (curid >= 0) atomicAdd(p.dZ + ((size_t)(uint32_t)curid * (uint32_t)H + col_s), run);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = 64 * rg + 32 * rb + acc_row(r, lane_s);
                    if (row < rows_here) {
                        float dx = accx[rb][r];
                        if (p.mask)
                            dx *= p.mask[((uint64_t)t * p.Pmask + s_slotof[row]) * H + col_s];
                        else if (p.keep)
                            dx = (s_keep[((t & 1) * MT + row) * (H / 4) + (col_s >> 2)] >> (col_s & 3)) & 1 ? dx * keep_scale : 0.0f;
                        atomicAdd(p.dZ + ((size_t)(uint32_t)s_rowidx[row * L + t] * (uint32_t)H + col_s), dx);
                    }
                }
        }
        T4_STAMP(8 * ti + 5);
    }
}


#endif  // PN_EXPERIMENTAL

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
            if (!PN_W4_NOLOAD) async_load_b128(rg[e], srcc + min(k0 + 4 * rq + e, p.R - 1) * ld);
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
        if (PN_W4_NOSPLIT) {
            x0 = __float_as_uint(v[0]); x1 = __float_as_uint(v[1]); x2 = x0 ^ x1;
            y0 = __float_as_uint(v[2]); y1 = __float_as_uint(v[3]); y2 = y0 ^ y1;
        } else {
            split3(v[0], v[1], x0, x1, x2);
            split3(v[2], v[3], y0, y1, y2);
        }
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
        if (!PN_W4_NOMFMA) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = mfma_bf16(a[i], b[j], acc[i][j]);
        }
        W4_FENCE();
    };
    auto step = [&](int buf, u32x4 (&A0)[2], u32x4 (&A1)[2], u32x4 (&B0)[4], u32x4 (&B1)[4], f32x4 (&rg)[4], int64_t inext) {
        [[maybe_unused]] const int ts = ((int)inext - 1 - 8) * 5;       // (tuning builds stamp tiles 8 .. 8 + 101)
        W4_STAMP(ts + 1);
        const u32x4 *fa = lds4 + (PN_W4_NOFRAG ? 0 : buf * W4_STAGE) + sa, *fb = lds4 + (PN_W4_NOFRAG ? 0 : buf * W4_STAGE) + sb;
        const u32x4 *na = lds4 + (PN_W4_NOFRAG ? 0 : (buf ^ 1) * W4_STAGE) + sa, *nb = lds4 + (PN_W4_NOFRAG ? 0 : (buf ^ 1) * W4_STAGE) + sb;
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
        W4_STAMP(ts + 2);
        __syncthreads();        // stage buf ^ 1 is complete; every read of stage buf has been issued and has landed
        W4_STAMP(ts + 3);
        group(A0, B1);                                  // a0.b2
#pragma unroll
        for (int i = 0; i < 2; i++) A0[i] = na[i * 8];                          // next tile: a plane 0
#pragma unroll
        for (int j = 0; j < 4; j++) B1[j] = nb[j * 8];                          //            b plane 0 (the next step's B0)
        group(A1, B0);                                  // a2.b0
#pragma unroll
        for (int j = 0; j < 4; j++) B0[j] = nb[W4_PLANE + j * 8];               //            b plane 1 (the next step's B1)
        W4_STAMP(ts + 4);
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
#if PN_W4_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);       // the second-dispatched half loses every arbitration otherwise
#endif
    // two tiles per trip (register sets, stages and the b fragment arrays alternate); an odd count runs one tile of zeros
#pragma unroll 1
    for (int64_t i = 0; i < my_tiles; i += 2) {
        W4_STAMP(((int)i - 8) * 5);
        issue(rgA, i + 2);
        wait_vm<4>(rgB[0], rgB[1], rgB[2], rgB[3]);         // tile i + 1 has arrived (the four loads just issued may be out)
        step(0, fA0, fA1, fB0, fB1, rgB, i + 1);
        W4_STAMP(((int)i + 1 - 8) * 5);
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

#if PN_TRACE4
extern "C" int pn_debug_set_trace4(long long *dev_buf) {     // tuning builds only; not part of the ABI
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace4), &dev_buf, sizeof dev_buf) == hipSuccess ? 0 : -4;
}
#endif

namespace pn {

int seq4_select(const pn_context *ctx, int H, int G, int L) {
    if (H != H4 || G != G4 || L < 1 || L > 8) return 0;      // (LDS: the index arrays of a 128-path tile)
    // default: the weight-gradient GEMM of this file (commit of the next K tile between the MFMA groups of the current
    // one: 0.278 vs 0.297 ms, A/B in one session); its forward and BPTT measured slower than the fused kernels and stay
    // opt-in (PN_SEQ4 = bit mask, 0 = every fused kernel)
    const int mask = knobs_of(ctx).seq4;
    return mask & (PN_EXPERIMENTAL ? (SEQ4_FWD | SEQ4_BWD | SEQ4_WGRAD) : SEQ4_WGRAD);
}

#if PN_EXPERIMENTAL
int launch_pack_fwd4(void *stream, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, int H, int G,
                     int gru, void *Wp, float *biasc) {
    if (H != H4 || G != G4) PN_FAIL(PN_ERR_ARG, "pack_fwd4: shape");
    const int n = F_KS * NW4 * G4 * 64;
    hipLaunchKernelGGL(pack_fwd4_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh, b_ih, b_hh,
                       gru, reinterpret_cast<u32x4 *>(Wp), biasc);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int launch_seq_fwd4(pn_context *ctx, void *stream, int gc, const SeqFwdParams &sp) {
    if (!sp.xh) PN_FAIL(PN_ERR_ARG, "seq_fwd4: the [x | h] rows are required");
    const size_t lds_bytes = (size_t)F_IDX_OFF + (size_t)MT4 * (sp.L + 1) * 4;
    const int blocks = (sp.P + MT4 - 1) / MT4;
    if (gc == 3) {
        auto kern = seq_fwd4_kernel<3>;
        if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), (int)lds_bytes)) return rc;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT4), lds_bytes, (hipStream_t)stream, sp);
    } else {
        auto kern = seq_fwd4_kernel<4>;
        if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), (int)lds_bytes)) return rc;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT4), lds_bytes, (hipStream_t)stream, sp);
    }
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}


int launch_pack_bwd4(void *stream, const float *w_ih, const float *w_hh, int H, int G, int gru, void *WpT) {
    if (H != H4 || G != G4) PN_FAIL(PN_ERR_ARG, "pack_bwd4: shape");
    const int n = B_NS * 2 * 2 * NW4 * 64;
    hipLaunchKernelGGL(pack_bwd4_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh, gru,
                       reinterpret_cast<u32x4 *>(WpT));
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

template <int GC, int NRG, int KK>
static int launch_seq_bwd4_t(pn_context *ctx, hipStream_t stream, const SeqBwdParams &sp) {
    constexpr int MT = 64 * NRG;
    const size_t lds_bytes = (size_t)2 * KK * (2 * NW4 * 3 * 1024 + (MT / 32) * 3 * 1024) + (size_t)MT * (sp.L + 1) * 4 +
                             (size_t)2 * MT * (H4 / 4);
    auto kern = seq_bwd4_kernel<GC, NRG, KK>;
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(kern), (int)lds_bytes)) return rc;
    hipLaunchKernelGGL(kern, dim3((sp.P + MT - 1) / MT), dim3(256 * NRG), lds_bytes, stream, sp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int launch_seq_bwd4(pn_context *ctx, void *stream, int gc, const SeqBwdParams &sp) {
    // PN_B4_WIDE=1: 128 paths per workgroup (one workgroup per CU); default: 64 paths, two workgroups per CU
    const bool wide = knobs_of(ctx).b4_wide != 0;
    hipStream_t st = (hipStream_t)stream;
    if (wide) return gc == 3 ? launch_seq_bwd4_t<3, 2, 2>(ctx, st, sp) : launch_seq_bwd4_t<4, 2, 2>(ctx, st, sp);
    return gc == 3 ? launch_seq_bwd4_t<3, 1, 1>(ctx, st, sp) : launch_seq_bwd4_t<4, 1, 1>(ctx, st, sp);
}


#else   // the shipped library: seq4_select never selects them
int launch_pack_fwd4(void *, const float *, const float *, const float *, const float *, int, int, int, void *, float *) {
    PN_FAIL(PN_ERR_ARG, "seq_fwd4: not in this build (PN_EXPERIMENTAL)");
}
int launch_seq_fwd4(pn_context *, void *, int, const SeqFwdParams &) { PN_FAIL(PN_ERR_ARG, "seq_fwd4: not in this build (PN_EXPERIMENTAL)"); }
int launch_pack_bwd4(void *, const float *, const float *, int, int, int, void *) {
    PN_FAIL(PN_ERR_ARG, "seq_bwd4: not in this build (PN_EXPERIMENTAL)");
}
int launch_seq_bwd4(pn_context *, void *, int, const SeqBwdParams &) { PN_FAIL(PN_ERR_ARG, "seq_bwd4: not in this build (PN_EXPERIMENTAL)"); }
#endif

int launch_wgrad4(pn_context *ctx, void *stream, const WgradParams &wp, int nsplit) {
    if (int rc = ensure_dynamic_lds(ctx, reinterpret_cast<const void *>(wgrad4_kernel), W4_LDS_BYTES)) return rc;
    hipLaunchKernelGGL(wgrad4_kernel, dim3((wp.H2 + W4_BN - 1) / W4_BN, (wp.GH + W4_BM - 1) / W4_BM, nsplit), dim3(W4_THREADS),
                       W4_LDS_BYTES, (hipStream_t)stream, wp);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

}  // namespace pn

// which of the 128-path kernels this build holds (bit mask of SEQ4_*): tests skip what is not there; not part of the ABI
extern "C" int pn_debug_seq4_kernels(void) { return PN_EXPERIMENTAL ? 7 : 4; }
