// pn_sampler.hip -- the MERW walker on gfx950: one walk per lane.
//
// Replaces the hot loop of /root/reference/preprocess/gen_merw.cpp:182-209 (and the per-epoch file
// variant gen_epoch_merw.cpp:164-206).  Per step a lane does what AliasTable::roll() (:81-91) does:
//   slot = r0 % table_len;  next = (r1 >= thr[slot]) ? A[slot] : B[slot]
// where thr is the exact integer form of `1.0*r1/RAND_MAX > S[slot]` computed on the host
// (pn_alias_build), and emits the node id and the distance code dis[st][u]-1 (:192-193, :201).
//
// Draw sources
//   PN_DRAW_GLIBC_REPLAY: the sequential glibc rand() stream after srand(seed), regenerated ON THE
//     DEVICE.  The TYPE_3 generator is a linear recurrence over Z/2^32 (r[j+31] = r[j] + r[j+28]),
//     so the state at any stream position is a polynomial jump x^d mod (x^31 - x^28 - 1) applied to
//     the seed state.  The host prepares three tiny tables (one state per epoch of the window, one
//     jump per block, one jump per thread); each thread then jumps to its own position and emits
//     248 consecutive draws.  The walker reads draw 2*(walk*L + t) (+1) from that buffer, which
//     reproduces the reference binary's output bit for bit.
//   PN_DRAW_PHILOX: rocRAND's Philox4x32-10 device generator, subsequence = global walk index;
//     no stream buffer, any window of any epoch is independent (throughput mode).
//
// Memory: alias triples are packed {A, B, thr, 0} = one 16-byte load per roll; the dense hop table
// is n*n bytes in HBM (Pubmed-size 389 MB, the reference's cap n = 100050 is 10 GB of 288 GB).
// The first hop of every walk starts at its source node, so each workgroup stages the alias tables
// of the few source nodes it covers in LDS; later hops hit L2.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <rocrand/rocrand_kernel.h>

#include <cstring>
#include <type_traits>
#include <vector>

#include "pn_internal.h"

namespace {

constexpr int kFillThreads = 256;               // threads per block of the stream generator
constexpr int kFillBatches = 8;                 // 31-draw batches per thread
constexpr int kDrawsPerThread = 31 * kFillBatches;          // 248
constexpr int kDrawsPerBlock = kDrawsPerThread * kFillThreads;  // 63488
constexpr int kWalkThreads = 256;
constexpr int kStageTriples = 1024;             // 16 KB of LDS for first-hop tables

// new_state[m] = sum_k c[k] * w[m + k], w = the 31-word state extended by 30 recurrence steps
__device__ __forceinline__ void glibc_jump(const uint32_t *__restrict__ c, uint32_t (&s)[31]) {
    uint32_t w[61];
#pragma unroll
    for (int j = 0; j < 31; j++) w[j] = s[j];
#pragma unroll
    for (int j = 31; j < 61; j++) w[j] = w[j - 31] + w[j - 3];
    uint32_t coef[31];
#pragma unroll
    for (int k = 0; k < 31; k++) coef[k] = c[k];
#pragma unroll
    for (int m = 0; m < 31; m++) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 31; k++) acc += coef[k] * w[m + k];
        s[m] = acc;
    }
}

// grid = (blocks_per_epoch, epoch_count).  Segment e holds seg_len draws that start at the stream
// position encoded in epoch_state[e].
__global__ __launch_bounds__(kFillThreads) void glibc_fill_kernel(const uint32_t *__restrict__ epoch_state,
                                                                   const uint32_t *__restrict__ block_jump,
                                                                   const uint32_t *__restrict__ thread_jump,
                                                                   int64_t seg_len, int32_t *__restrict__ draws) {
    const int e = blockIdx.y;
    const int64_t first = (int64_t)blockIdx.x * kDrawsPerBlock + (int64_t)threadIdx.x * kDrawsPerThread;
    if (first >= seg_len) return;
    uint32_t s[31];
#pragma unroll
    for (int j = 0; j < 31; j++) s[j] = epoch_state[e * 31 + j];
    glibc_jump(block_jump + (size_t)blockIdx.x * 31, s);
    glibc_jump(thread_jump + (size_t)threadIdx.x * 31, s);
    int32_t *out = draws + (int64_t)e * seg_len;
#pragma unroll 1
    for (int b = 0; b < kFillBatches; b++) {
        const int64_t at = first + 31 * b;
#pragma unroll
        for (int j = 0; j < 31; j++)
            if (at + j < seg_len) out[at + j] = (int32_t)(s[j] >> 1);
        // next 31 words: n[j] = s[j] + s[j+28] (j < 3), n[j] = s[j] + n[j-3] (j >= 3)
        uint32_t nx[31];
#pragma unroll
        for (int j = 0; j < 31; j++) nx[j] = s[j] + (j < 3 ? s[j + 28] : nx[j - 3]);
#pragma unroll
        for (int j = 0; j < 31; j++) s[j] = nx[j];
    }
}

struct WalkParams {
    int32_t n;
    const int64_t *off;
    const int4 *triples;
    const uint8_t *dis;
    int32_t W, L;
    uint64_t seed;
    int64_t epoch_begin, epoch_count;
    int32_t node_begin, node_count;
    int32_t *ids;
    uint8_t *codes;
    const int32_t *draws;  // GLIBC only: [epoch_count][node_count*W*dps*L]
    int32_t *status;
    int32_t dps;           // draws per step: 2 = alias roll (gen_merw.cpp:81-91), 1 = uniform rand() % deg (gen.cpp:113-114)
    const pn_step_state *dyn;   // Philox: seed / epoch_begin read from device memory when set (hipGraph replay)
    const uint2 *node_ref;      // [n] {first triple, count} (one 8-byte load per roll) or null: two words of off[]
    const int32_t *node_list;   // source node of window slot i (Philox) or null: node_begin + i
    int32_t stage;              // stage the first-hop tables in LDS (launches with several workgroups per CU; a launch of one
                                // round of workgroups only pays the staging's extra memory round trips: 10.5 -> 14 us at 52 k walks)
};

// table position of node x
__device__ __forceinline__ void node_table(const WalkParams &p, int32_t x, int64_t &o0, int32_t &len) {
    if (p.node_ref) {
        const uint2 r = p.node_ref[x];
        o0 = r.x;
        len = (int32_t)r.y;
    } else {
        o0 = p.off[x];
        len = (int32_t)(p.off[x + 1] - o0);
    }
}

// The draws of step t of a walk.  Alias roll: draws 2t, 2t+1 of the walk's stream; uniform: draw t.  Philox words
// come four at a time: draw q is word q & 3 of block q >> 2.
template <int DRAW>
__device__ __forceinline__ void step_draws(const int dps, const int32_t *my_draws, rocrand_state_philox4x32_10 &rng,
                                           uint4 &word, int32_t t, uint32_t &r0, uint32_t &r1) {
    if (DRAW == PN_DRAW_GLIBC_REPLAY) {
        r0 = (uint32_t)my_draws[dps * t];
        r1 = dps == 1 ? 0u : (uint32_t)my_draws[2 * t + 1];
    } else {
        const int q0 = dps * t;                                   // index of the step's first draw in the walk's stream
        if ((q0 & 3) == 0) word = rocrand4(&rng);                 // (one call site: a new block of four words)
        const int k = q0 & 3;                                     // alias roll: k is 0 or 2
        r0 = (k == 0 ? word.x : k == 1 ? word.y : k == 2 ? word.z : word.w) >> 1;
        r1 = dps == 1 ? 0u : (k == 0 ? word.y : word.w) >> 1;
    }
}

// L4: path length 4 (the reference's default): ids and codes of a path leave as one 16-byte and one 4-byte store
template <int DRAW, bool L4>
__global__ __launch_bounds__(kWalkThreads) void merw_walk_kernel(WalkParams p) {
    __shared__ int4 s_tab[kStageTriples];
    const int64_t per_epoch = (int64_t)p.node_count * p.W;
    const int64_t total = per_epoch * p.epoch_count;
    const int64_t g0 = (int64_t)blockIdx.x * kWalkThreads;
    const int64_t g = g0 + threadIdx.x;

    // ---- stage the first-hop tables of the source nodes this block covers (block-uniform): a block of 256 walks starts at
    //      1 + 255 / W source nodes, consecutive in the window -- a node range, or a list of nodes (what a training step
    //      samples: its masked nodes), whose tables are scattered over the triple array and are staged one by one ------
    constexpr int kStageNodes = 16;
    __shared__ int64_t s_o0[kStageNodes];
    __shared__ int32_t s_len[kStageNodes];
    const int64_t g_last = (g0 + kWalkThreads - 1 < total ? g0 + kWalkThreads - 1 : total - 1);
    const int64_t e_first = g0 / per_epoch, e_last = g_last / per_epoch;
    const int32_t slot_lo = (int32_t)((g0 % per_epoch) / p.W);
    int32_t nstage = (p.stage && e_first == e_last) ? (int32_t)((g_last % per_epoch) / p.W) - slot_lo + 1 : 0;
    if (nstage > kStageNodes) nstage = 0;
    if ((int)threadIdx.x < nstage) {
        const int32_t sl = slot_lo + (int32_t)threadIdx.x;
        const int32_t nd = p.node_list ? min(max(p.node_list[sl], 0), p.n - 1) : p.node_begin + sl;
        int64_t o0;
        int32_t len;
        node_table(p, nd, o0, len);
        s_o0[threadIdx.x] = o0;
        s_len[threadIdx.x] = len > 0 ? len : 0;
    }
    __syncthreads();
    int32_t my_stage = -1;          // where this walk's first-hop table starts in s_tab (-1: not staged)
    {
        // (all tables in one sweep -- every thread's loads are independent and in flight together; a loop over the nodes
        //  paid one memory round trip per node and made the 52 k-walk launch of a training step 5 us slower than not
        //  staging at all)
        int32_t tot = 0;
        for (int j = 0; j < nstage; j++) tot += s_len[j];
        if (tot > kStageTriples) nstage = 0, tot = 0;
        for (int32_t i = threadIdx.x; i < tot; i += kWalkThreads) {
            int32_t pre = 0, j = 0;
            while (i >= pre + s_len[j]) pre += s_len[j++];
            s_tab[i] = p.triples[s_o0[j] + (i - pre)];
        }
        if (nstage > 0 && g < total) {
            const int32_t mine = (int32_t)((g - e_first * per_epoch) / p.W) - slot_lo;
            int32_t pre = 0;
            for (int j = 0; j < mine; j++) pre += s_len[j];
            my_stage = pre;
        }
    }
    __syncthreads();
    if (g >= total) return;

    const int64_t e_l = g / per_epoch;
    const int64_t rem = g - e_l * per_epoch;
    const int32_t st_i = (int32_t)(rem / p.W);
    // (a listed source node outside the graph is clamped: the host checks ranges, not lists in device memory)
    const int32_t st = p.node_list ? min(max(p.node_list[st_i], 0), p.n - 1) : p.node_begin + st_i;
    const int32_t wi = (int32_t)(rem % p.W);
    const int64_t epoch_begin = p.dyn ? p.dyn->epoch : p.epoch_begin;
    const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
    const uint64_t walk = ((uint64_t)(epoch_begin + e_l) * (uint64_t)p.n + (uint64_t)st) * (uint64_t)p.W + wi;

    rocrand_state_philox4x32_10 rng;
    uint4 word = {0, 0, 0, 0};
    if (DRAW == PN_DRAW_PHILOX) rocrand_init(seed, walk, 0, &rng);
    const int32_t *my_draws = DRAW == PN_DRAW_GLIBC_REPLAY ? p.draws + g * p.dps * (int64_t)p.L : nullptr;

    const uint8_t *dis_row = p.dis + (size_t)st * (size_t)p.n;
    int32_t *out_ids = p.ids + g * p.L;
    uint8_t *out_codes = p.codes + g * p.L;
    int32_t x = st;
    int32_t idv[4] = {0, 0, 0, 0};          // L4: the path's ids / codes, stored once at the end
    uint32_t cdv = 0;
    int32_t dead = 1 << 30, fill_x = 0;     // first step after an empty alias table, and what fills the rest
    uint8_t fill_c = 0;
    const int32_t steps = L4 ? 4 : p.L;
    // one path node (no lambda: the Philox state must stay in registers); `break`s out of the enclosing loop on an
    // empty alias table
#define PN_WALK_STEP(t)                                                                            \
    {                                                                                              \
        const uint8_t code = (uint8_t)(dis_row[x] - 1);                                            \
        if constexpr (L4) {                                                                        \
            idv[(t) & 3] = x;                                                                      \
            cdv |= (uint32_t)code << (8 * ((t) & 3));                                              \
        } else {                                                                                   \
            out_ids[t] = x;                                                                        \
            out_codes[t] = code;                                                                   \
        }                                                                                          \
        int64_t o0;                                                                                \
        int32_t len;                                                                               \
        node_table(p, x, o0, len);                                                                 \
        uint32_t r0, r1;                                                                           \
        step_draws<DRAW>(p.dps, my_draws, rng, word, t, r0, r1);                                   \
        if (len <= 0) {                                                                            \
            if (p.status) atomicExch(p.status, PN_ERR_EMPTY_TABLE);                                \
            dead = (t) + 1; /* the rest of the path repeats this node (the reference exits here) */ \
            fill_x = x;                                                                            \
            fill_c = code;                                                                         \
            break;                                                                                 \
        }                                                                                          \
        const int64_t slot = o0 + (int64_t)(r0 % (uint32_t)len);                                   \
        int4 tr;                                                                                   \
        if ((t) == 0 && my_stage >= 0)                                                             \
            tr = s_tab[my_stage + (int32_t)(slot - o0)];                                           \
        else                                                                                       \
            tr = p.triples[slot];                                                                  \
        x = (r1 >= (uint32_t)tr.z) ? tr.x : tr.y;                                                  \
    }
    if constexpr (L4) {
#pragma unroll
        for (int32_t t = 0; t < 4; t++) PN_WALK_STEP(t)
    } else {
        for (int32_t t = 0; t < steps; t++) PN_WALK_STEP(t)
    }
#undef PN_WALK_STEP
    if constexpr (L4) {
#pragma unroll
        for (int k = 1; k < 4; k++)
            if (k >= dead) {
                idv[k] = fill_x;
                cdv |= (uint32_t)fill_c << (8 * k);
            }
        *reinterpret_cast<int4 *>(out_ids) = make_int4(idv[0], idv[1], idv[2], idv[3]);
        *reinterpret_cast<uint32_t *>(out_codes) = cdv;
    } else {
        for (int32_t k = dead; k < steps; k++) {
            out_ids[k] = fill_x;
            out_codes[k] = fill_c;
        }
    }
}

// =================================================================================================
// On-the-fly hop codes (dis == NULL): the walker for graphs whose dense n*n hop table does not fit.
//
// One wavefront per source node st.  It first builds the out-ball of st of radius RF <= 2 with exact
// distances in a private LDS hash table (level-synchronous, insert-if-absent keeps the smaller level).
// hops(st -> x) for a walk node x at step t (known to be <= t) is then
//     min( t,  df(x),  min over in-neighbour chains  x <- c1 <- ... <- ck  of  k + df(ck) )
// with df = distance in the ball.  This is exact: a shortest path of length d has its node at position
// min(d, RF) inside the ball, at backward depth d - min(d, RF) from x; every other candidate is the
// length of a real path, i.e. an upper bound.  The backward search is depth-limited by the best bound found so
// far (best - 1 - RF), which for L <= RF + 2 (L = 4) means no search at all and for L = 6 two levels.
// A hub whose 2-ball overflows the table falls back to RF = 1 or 0 (deeper backward search, still exact).
//
// Round 4 -- the LAST level of that search is where the time went at L = 6 (configs[4]: a walk node at step 5 is hardly
// ever inside the 2-ball, so all 16 in-neighbours and all their 256 in-neighbours were looked up, for nothing: 195 M
// paths/s against 17 G at L = 4).  Scanning the in-neighbours cc of a node c at the last allowed level can only succeed if
// some cc lies in the ball, i.e. if c is an out-neighbour of a ball node; for c outside the ball that means c is an
// out-neighbour of a LEVEL-2 node.  Those are recorded once per source node in a Bloom filter (32 Kbit, two hashes: no false
// negatives) and a last-level scan is entered only on "maybe" -- a false positive costs one wasted scan, never a wrong code.
// A node c found inside the ball is not expanded either: the best path through it is k + df(c), already recorded.
// =================================================================================================
#ifndef PN_OTF_ABL
#define PN_OTF_ABL 0     // tuning builds only: 1 = no Bloom filter (round 3's search), 2 = no backward search at all (wrong codes)
#endif
constexpr int kOtfWaves = 2;                  // source nodes per workgroup (16 KB of LDS each: five workgroups per CU)
constexpr int kOtfCap = 2048;                 // hash slots per wave (8 KB)
constexpr uint32_t kOtfEmpty = 0xFFFFFFFFu;
constexpr int kOtfMaxDepth = 8;
constexpr int kOtfBloomWords = 1024;          // 32 Kbit per wave
__device__ __forceinline__ void otf_bloom_bits(int32_t node, uint32_t &b0, uint32_t &b1, uint32_t &b2) {
    b0 = ((uint32_t)node * 0x9E3779B1u) >> 17;         // 15 bits each
    b1 = ((uint32_t)node * 0x85EBCA6Bu + 0x27D4EB2Fu) >> 17;
    b2 = ((uint32_t)node * 0xC2B2AE35u + 0x165667B1u) >> 17;
}
__device__ __forceinline__ bool otf_bloom_maybe(const uint32_t *bloom, int32_t node) {
    uint32_t b0, b1, b2;
    otf_bloom_bits(node, b0, b1, b2);
    return ((bloom[b0 >> 5] >> (b0 & 31)) & (bloom[b1 >> 5] >> (b1 & 31)) & (bloom[b2 >> 5] >> (b2 & 31)) & 1u) != 0;
}

struct OtfParams {
    WalkParams w;
    const int64_t *adj_off;
    const int32_t *adj;
    const int64_t *radj_off;
    const int32_t *radj;
};

__device__ __forceinline__ uint32_t otf_hash(uint32_t x) {
    x *= 0x9E3779B1u;
    return (x ^ (x >> 15)) & (kOtfCap - 1);
}
// returns true if (node, d) was newly inserted
__device__ __forceinline__ bool otf_insert(uint32_t *tab, int32_t node, int d) {
    uint32_t h = otf_hash((uint32_t)node);
    const uint32_t packed = ((uint32_t)node << 3) | (uint32_t)d;
    for (int probe = 0; probe < kOtfCap; probe++) {
        const uint32_t old = atomicCAS(&tab[h], kOtfEmpty, packed);
        if (old == kOtfEmpty) return true;
        if ((old >> 3) == (uint32_t)node) return false;     // already there with a level <= d
        h = (h + 1) & (kOtfCap - 1);
    }
    return false;
}
__device__ __forceinline__ int otf_lookup(const uint32_t *tab, int32_t node) {
    uint32_t h = otf_hash((uint32_t)node);
    for (int probe = 0; probe < kOtfCap; probe++) {
        const uint32_t v = tab[h];
        if (v == kOtfEmpty) return -1;
        if ((v >> 3) == (uint32_t)node) return (int)(v & 7u);
        h = (h + 1) & (kOtfCap - 1);
    }
    return -1;
}

template <int DRAW>
__global__ __launch_bounds__(kOtfWaves * 64) void merw_walk_otf_kernel(OtfParams op) {
    __shared__ uint32_t s_tab[kOtfWaves][kOtfCap];
    __shared__ int s_cnt[kOtfWaves];
    __shared__ uint32_t s_stack[kOtfWaves][2][kOtfMaxDepth][64];   // per lane DFS cursor / end per level (CSR positions);
                                                                    // before the walks: the list of the ball's level-2 nodes
    __shared__ uint32_t s_bloom[kOtfWaves][kOtfBloomWords];
    const WalkParams &p = op.w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int st_l = blockIdx.x * kOtfWaves + wave;
#define CUR(d_) s_stack[wave][0][d_][lane]
#define END(d_) s_stack[wave][1][d_][lane]
    if (st_l >= p.node_count) return;                       // wave-uniform; no block barriers below
    const int32_t st = p.node_list ? min(max(p.node_list[st_l], 0), p.n - 1) : p.node_begin + st_l;
    uint32_t *tab = s_tab[wave];

    // ---- out-ball of st, radius rf (largest of 2, 1, 0 that keeps the table at most half full) ----------
    int rf = 2;
    for (;; rf--) {
        for (int i = lane; i < kOtfCap; i += 64) tab[i] = kOtfEmpty;
        if (lane == 0) s_cnt[wave] = 0;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            otf_insert(tab, st, 0);
            s_cnt[wave] = 1;
        }
        __builtin_amdgcn_wave_barrier();
        const int64_t b0 = op.adj_off[st], e0 = op.adj_off[st + 1];
        bool overflow = false;
        if (rf >= 1) {
            if (e0 - b0 > kOtfCap / 2) overflow = true;
            if (!overflow) {
                for (int64_t j = b0 + lane; j < e0; j += 64)
                    if (otf_insert(tab, op.adj[j], 1)) atomicAdd(&s_cnt[wave], 1);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (rf >= 2 && !overflow) {
            for (int64_t j = b0; j < e0 && !overflow; j++) {
                const int32_t a = op.adj[j];
                const int64_t b1 = op.adj_off[a], e1 = op.adj_off[a + 1];
                if (s_cnt[wave] + (e1 - b1) > kOtfCap / 2) {    // could exceed half load: give up this radius
                    overflow = true;
                    break;
                }
                for (int64_t k = b1 + lane; k < e1; k += 64)
                    if (otf_insert(tab, op.adj[k], 2)) atomicAdd(&s_cnt[wave], 1);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!overflow || rf == 0) break;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- out-neighbours of the ball's level-2 nodes -> Bloom filter (only when a search can happen: L >= rf + 3) ----------
    uint32_t *bloom = s_bloom[wave];
    bool bloom_ok = false;
    if (rf == 2 && p.L >= 5 && PN_OTF_ABL != 1) {
        uint32_t *list = &s_stack[wave][0][0][0];             // 2 * kOtfMaxDepth * 64 = 1024 entries: the table is at most half full
        for (int i = lane; i < kOtfBloomWords; i += 64) bloom[i] = 0u;
        if (lane == 0) s_cnt[wave] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < kOtfCap; i += 64) {
            const uint32_t v = tab[i];
            if (v != kOtfEmpty && (v & 7u) == 2u) list[atomicAdd(&s_cnt[wave], 1)] = v >> 3;
        }
        __builtin_amdgcn_wave_barrier();
        const int n2 = s_cnt[wave];
        // a lane per level-2 node: its out-neighbours, three bits each; the list bounds of a lane's NEXT node are fetched
        // while it walks the current one
        int64_t kb_n = 0, ke_n = 0;
        if (lane < n2) kb_n = op.adj_off[(int32_t)list[lane]], ke_n = op.adj_off[(int32_t)list[lane] + 1];
        for (int i = lane; i < n2; i += 64) {
            const int64_t kb = kb_n, ke = ke_n;
            if (i + 64 < n2) kb_n = op.adj_off[(int32_t)list[i + 64]], ke_n = op.adj_off[(int32_t)list[i + 64] + 1];
            for (int64_t k = kb; k < ke; k += 4) {       // four loads in flight (the tail repeats the last entry)
                int32_t nb[4];
#pragma unroll
                for (int u = 0; u < 4; u++) nb[u] = op.adj[min(k + u, ke - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t b0, b1, b2;
                    otf_bloom_bits(nb[u], b0, b1, b2);
                    atomicOr(&bloom[b0 >> 5], 1u << (b0 & 31));
                    atomicOr(&bloom[b1 >> 5], 1u << (b1 & 31));
                    atomicOr(&bloom[b2 >> 5], 1u << (b2 & 31));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        bloom_ok = true;
    }

    // ---- walks: lane = walk index (loop when W > 64), all epochs of the window -------------------------
    const int64_t epoch_begin = p.dyn ? p.dyn->epoch : p.epoch_begin;
    const uint64_t seed = p.dyn ? p.dyn->seed : p.seed;
    for (int64_t e_l = 0; e_l < p.epoch_count; e_l++) {
        for (int wi = lane; wi < p.W; wi += 64) {
            const int64_t g = (e_l * p.node_count + st_l) * p.W + wi;
            const uint64_t walk = ((uint64_t)(epoch_begin + e_l) * (uint64_t)p.n + (uint64_t)st) * (uint64_t)p.W + wi;
            rocrand_state_philox4x32_10 rng;
            uint4 word = {0, 0, 0, 0};
            if (DRAW == PN_DRAW_PHILOX) rocrand_init(seed, walk, 0, &rng);
            const int32_t *my_draws = DRAW == PN_DRAW_GLIBC_REPLAY ? p.draws + g * p.dps * (int64_t)p.L : nullptr;
            int32_t *out_ids = p.ids + g * p.L;
            uint8_t *out_codes = p.codes + g * p.L;
            int32_t x = st;
            for (int32_t t = 0; t < p.L; t++) {
                // ---- exact hop code of x ------------------------------------------------------------------
                int best = t;
                if (x == st) best = 0;
                if (best > 0) {
                    const int d0 = otf_lookup(tab, x);
                    if (d0 >= 0 && d0 < best) best = d0;
                    // depth-limited DFS over in-neighbour chains; a level k can only help while k <= best - 1 - rf
                    if (PN_OTF_ABL != 2 && bloom_ok && best - 1 - rf <= 2) {
                        // the common case (L <= 7 with the full 2-ball): at most two levels, written as two loops with the
                        // in-neighbour ids fetched four at a time -- same candidates as the generic search below
                        // Level 1 for every lane first; the in-neighbours that pass the filter are only NOTED (a lane's
                        // 16 stack slots), and expanded afterwards, each lane walking its own short list: a wavefront runs
                        // a level-2 scan whenever ANY of its lanes has one, so expanding inside the level-1 loop cost the
                        // wave one scan per level-1 position (1 - 0.95^40 = 87 % of them) instead of one per list entry.
                        int n_cand = 0;
                        if (best - 1 - rf == 2 || (best - 1 - rf == 1 && otf_bloom_maybe(bloom, x))) {
                            const uint32_t e1 = (uint32_t)op.radj_off[x + 1];
                            for (uint32_t j = (uint32_t)op.radj_off[x]; j < e1; j += 4) {
                                int32_t c[4];
#pragma unroll
                                for (int u = 0; u < 4; u++) c[u] = op.radj[min(j + u, e1 - 1)];
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const int dc = otf_lookup(tab, c[u]);
                                    if (dc >= 0) {
                                        if (1 + dc < best) best = 1 + dc;
                                    } else if (best - 1 - rf == 2 && j + u < e1 && otf_bloom_maybe(bloom, c[u])) {
                                        if (n_cand < 2 * kOtfMaxDepth) {
                                            s_stack[wave][n_cand >> 3][n_cand & 7][lane] = (uint32_t)c[u];
                                            n_cand++;
                                        } else {        // (list full: expand right here)
                                            const uint32_t e2 = (uint32_t)op.radj_off[c[u] + 1];
                                            for (uint32_t i2 = (uint32_t)op.radj_off[c[u]]; i2 < e2 && best - 1 - rf == 2; i2++) {
                                                const int d2 = otf_lookup(tab, op.radj[i2]);
                                                if (d2 >= 0 && 2 + d2 < best) best = 2 + d2;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                        // c may be an out-neighbour of a level-2 ball node: dist(st, c) = 3 iff one of ITS in-neighbours is one
                        for (int q = 0; q < n_cand && best - 1 - rf == 2; q++) {
                            const int32_t c = (int32_t)s_stack[wave][q >> 3][q & 7][lane];
                            const uint32_t e2 = (uint32_t)op.radj_off[c + 1];
                            for (uint32_t i2 = (uint32_t)op.radj_off[c]; i2 < e2 && best - 1 - rf == 2; i2 += 4) {
                                int32_t cc[4];
#pragma unroll
                                for (int w2 = 0; w2 < 4; w2++) cc[w2] = op.radj[min(i2 + w2, e2 - 1)];
#pragma unroll
                                for (int w2 = 0; w2 < 4; w2++) {
                                    const int d2 = otf_lookup(tab, cc[w2]);
                                    if (d2 >= 0 && 2 + d2 < best) best = 2 + d2;
                                }
                            }
                        }
                    } else if (PN_OTF_ABL != 2 && best - 1 - rf >= 1 && !(bloom_ok && best - 1 - rf == 1 && !otf_bloom_maybe(bloom, x))) {
                        int d = 0;
                        CUR(0) = (uint32_t)op.radj_off[x];
                        END(0) = (uint32_t)op.radj_off[x + 1];
                        while (d >= 0) {
                            const uint32_t cur = CUR(d);
                            if (cur >= END(d) || d + 1 > best - 1 - rf) {
                                d--;
                                continue;
                            }
                            CUR(d) = cur + 1;
                            const int32_t c = op.radj[cur];
                            const int k = d + 1;
                            const int dc = otf_lookup(tab, c);
                            if (dc >= 0 && k + dc < best) best = k + dc;
                            // expand c (scan ITS in-neighbours at level k + 1)?  Not when c is in the ball (k + df(c) is the
                            // best any path through c can do); at the last allowed level only when c can be an out-
                            // neighbour of a level-2 ball node at all
                            if (dc < 0 && k + 1 <= best - 1 - rf && d + 1 < kOtfMaxDepth &&
                                !(bloom_ok && k + 1 == best - 1 - rf && !otf_bloom_maybe(bloom, c))) {
                                d++;
                                CUR(d) = (uint32_t)op.radj_off[c];
                                END(d) = (uint32_t)op.radj_off[c + 1];
                            }
                        }
                    }
                }
                out_ids[t] = x;
                out_codes[t] = (uint8_t)best;
                // ---- roll (same as the dense-table walker) ------------------------------------------------
                int64_t o0;
                int32_t len;
                node_table(p, x, o0, len);
                uint32_t r0, r1;
                step_draws<DRAW>(p.dps, my_draws, rng, word, t, r0, r1);
                if (len <= 0) {
                    if (p.status) atomicExch(p.status, PN_ERR_EMPTY_TABLE);
                    for (int32_t k = t + 1; k < p.L; k++) {
                        out_ids[k] = x;
                        out_codes[k] = out_codes[t];
                    }
                    break;
                }
                const int4 tr = p.triples[o0 + (int64_t)(r0 % (uint32_t)len)];
                x = (r1 >= (uint32_t)tr.z) ? tr.x : tr.y;
            }
        }
    }
#undef CUR
#undef END
}

}  // namespace

extern "C" {

static int64_t fill_blocks(int64_t seg_len) { return (seg_len + kDrawsPerBlock - 1) / kDrawsPerBlock; }

int pn_sample_workspace_bytes(int32_t W, int32_t L, int32_t draw_source, int64_t epoch_count, int32_t node_count,
                              int64_t *bytes) {
    if (!bytes || W < 1 || L < 1 || epoch_count < 0 || node_count < 0) PN_FAIL(PN_ERR_ARG, "bad sampler window");
    if (draw_source == PN_DRAW_PHILOX) {
        *bytes = 0;
        return PN_OK;
    }
    if (draw_source != PN_DRAW_GLIBC_REPLAY) PN_FAIL(PN_ERR_ARG, "unknown draw source %d", draw_source);
    const int64_t seg_len = (int64_t)node_count * W * 2 * L;
    const int64_t tables = (epoch_count + fill_blocks(seg_len) + kFillThreads) * 31 * 4;
    *bytes = ((tables + 255) / 256) * 256 + epoch_count * seg_len * 4;
    return PN_OK;
}

int pn_sample_paths(pn_context *ctx, const pn_sampler_tables *tb, int32_t W, int32_t L, int32_t draw_source, uint64_t seed,
                    int64_t epoch_begin, int64_t epoch_count, int32_t node_begin, int32_t node_count, int32_t *ids,
                    uint8_t *codes, void *workspace, int64_t workspace_bytes, int32_t *status_flag,
                    const pn_step_state *step_state, const int32_t *node_list, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (node_list && draw_source != PN_DRAW_PHILOX)
        PN_FAIL(PN_ERR_ARG, "pn_sample_paths: a node list needs PN_DRAW_PHILOX (the glibc replay walks the reference's "
                "contiguous draw stream)");
    if (step_state && draw_source != PN_DRAW_PHILOX)
        PN_FAIL(PN_ERR_ARG, "pn_sample_paths: a device step state needs PN_DRAW_PHILOX (the glibc replay prepares its "
                "jump tables on the host)");
    if (int rc = pn::context_check_device(ctx)) return rc;
    if (!tb || !tb->off || !tb->triples || !ids || !codes) PN_FAIL(PN_ERR_ARG, "pn_sample_paths: null table or output");
    const bool otf = tb->dis == nullptr;
    if (otf && (!tb->adj_off || !tb->adj || !tb->radj_off || !tb->radj))
        PN_FAIL(PN_ERR_ARG, "pn_sample_paths: neither a dense hop table nor the CSR lists for on-the-fly hop codes");
    if (otf && (tb->n >= (1 << 28) || L > 8))
        PN_FAIL(PN_ERR_ARG, "on-the-fly hop codes support n < 2^28 and L <= 8 (n=%d L=%d)", tb->n, L);
    if (node_list) node_begin = 0;
    if (W < 1 || L < 1 || epoch_begin < 0 || epoch_count < 0 || node_begin < 0 || node_count < 0 ||
        (!node_list && (int64_t)node_begin + node_count > tb->n))
        PN_FAIL(PN_ERR_ARG, "pn_sample_paths: bad window (n=%d W=%d L=%d nodes [%d,+%d))", tb->n, W, L, node_begin,
                node_count);
    const int64_t total = epoch_count * node_count * W;
    if (total == 0) return PN_OK;
    if (total / kWalkThreads + 1 > 2147483647LL) PN_FAIL(PN_ERR_ARG, "window too large for one launch");

    WalkParams wp{};
    wp.n = tb->n;
    wp.off = tb->off;
    wp.triples = reinterpret_cast<const int4 *>(tb->triples);
    wp.dis = tb->dis;
    wp.W = W;
    wp.L = L;
    wp.seed = seed;
    wp.epoch_begin = epoch_begin;
    wp.epoch_count = epoch_count;
    wp.node_begin = node_begin;
    wp.node_count = node_count;
    wp.ids = ids;
    wp.codes = codes;
    wp.status = status_flag;
    const int dps = tb->draws_per_step == 1 ? 1 : 2;
    if (tb->draws_per_step != 0 && tb->draws_per_step != 1 && tb->draws_per_step != 2)
        PN_FAIL(PN_ERR_ARG, "pn_sample_paths: draws_per_step = %d (0, 1 or 2)", tb->draws_per_step);
    wp.dps = dps;
    wp.dyn = step_state;
    wp.node_ref = reinterpret_cast<const uint2 *>(tb->node_ref);
    wp.node_list = node_list;
    {
        // a node range was always staged (round 2's rates); a node list only when the launch has workgroups to overlap with
        const int forced = pn::knobs_of(ctx).sampler_stage;
        const int64_t walks = (int64_t)epoch_count * node_count * W;
        wp.stage = forced >= 0 ? forced != 0 : (!node_list || walks >= (int64_t)kWalkThreads * 1024);
    }

    if (draw_source == PN_DRAW_GLIBC_REPLAY) {
        int64_t need = 0;
        pn_sample_workspace_bytes(W, L, draw_source, epoch_count, node_count, &need);
        if (!workspace || workspace_bytes < need)
            PN_FAIL(PN_ERR_CAPACITY, "sampler workspace holds %lld bytes, need %lld", (long long)workspace_bytes,
                    (long long)need);
        const int64_t seg_len = (int64_t)node_count * W * dps * L;     // (the workspace is sized for dps = 2)
        const int64_t nblk = fill_blocks(seg_len);
        // host: one state per epoch of the window + the block / thread jump polynomials
        std::vector<uint32_t> host((size_t)(epoch_count + nblk + kFillThreads) * 31);
        const uint64_t stride = (uint64_t)dps * L * W * (uint64_t)tb->n;   // draws per epoch
        const uint64_t first = stride * (uint64_t)epoch_begin + (uint64_t)dps * L * W * (uint64_t)node_begin;
        pn::GlibcState base = pn::glibc_seed_state((uint32_t)seed);
        pn::GlibcPoly pos = pn::glibc_poly_xpow(first);
        const pn::GlibcPoly step = pn::glibc_poly_xpow(stride);
        for (int64_t e = 0; e < epoch_count; e++) {
            pn::GlibcState st = pn::glibc_apply(pos, base);
            memcpy(&host[(size_t)e * 31], st.s, sizeof st.s);
            pos = pn::glibc_poly_mul(pos, step);
        }
        {
            pn::GlibcPoly acc = pn::glibc_poly_one();
            const pn::GlibcPoly jb = pn::glibc_poly_xpow(kDrawsPerBlock);
            for (int64_t b = 0; b < nblk; b++) {
                memcpy(&host[(size_t)(epoch_count + b) * 31], acc.c, sizeof acc.c);
                acc = pn::glibc_poly_mul(acc, jb);
            }
            acc = pn::glibc_poly_one();
            const pn::GlibcPoly jt = pn::glibc_poly_xpow(kDrawsPerThread);
            for (int t = 0; t < kFillThreads; t++) {
                memcpy(&host[(size_t)(epoch_count + nblk + t) * 31], acc.c, sizeof acc.c);
                acc = pn::glibc_poly_mul(acc, jt);
            }
        }
        uint32_t *d_tables = reinterpret_cast<uint32_t *>(workspace);
        const int64_t table_bytes = (int64_t)host.size() * 4;
        int32_t *d_draws = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(workspace) +
                                                       ((table_bytes + 255) / 256) * 256);
        PN_CHECK_HIP(hipMemcpyAsync(d_tables, host.data(), (size_t)table_bytes, hipMemcpyHostToDevice, stream));
        PN_CHECK_HIP(hipStreamSynchronize(stream));  // `host` dies at scope exit (parity mode, not the fast path)
        dim3 grid((unsigned)nblk, (unsigned)epoch_count);
        {
        pn::StageTimer tm(ctx, pn::ST_SAMPLER_FILL, stream);
        hipLaunchKernelGGL(glibc_fill_kernel, grid, dim3(kFillThreads), 0, stream, d_tables,
                           d_tables + (size_t)epoch_count * 31, d_tables + (size_t)(epoch_count + nblk) * 31, seg_len,
                           d_draws);
        }
        PN_CHECK_HIP(hipGetLastError());
        wp.draws = d_draws;
        pn::StageTimer tm(ctx, pn::ST_SAMPLER_WALK, stream);
        if (otf) {
            OtfParams op{wp, tb->adj_off, tb->adj, tb->radj_off, tb->radj};
            hipLaunchKernelGGL(merw_walk_otf_kernel<PN_DRAW_GLIBC_REPLAY>, dim3((node_count + kOtfWaves - 1) / kOtfWaves),
                               dim3(kOtfWaves * 64), 0, stream, op);
        } else {
            const unsigned blocks = (unsigned)((total + kWalkThreads - 1) / kWalkThreads);
            if (L == 4)
                hipLaunchKernelGGL((merw_walk_kernel<PN_DRAW_GLIBC_REPLAY, true>), dim3(blocks), dim3(kWalkThreads), 0, stream, wp);
            else
                hipLaunchKernelGGL((merw_walk_kernel<PN_DRAW_GLIBC_REPLAY, false>), dim3(blocks), dim3(kWalkThreads), 0, stream, wp);
        }
        PN_CHECK_HIP(hipGetLastError());
    } else if (draw_source == PN_DRAW_PHILOX) {
        pn::StageTimer tm(ctx, pn::ST_SAMPLER_WALK, stream);
        if (otf) {
            OtfParams op{wp, tb->adj_off, tb->adj, tb->radj_off, tb->radj};
            hipLaunchKernelGGL(merw_walk_otf_kernel<PN_DRAW_PHILOX>, dim3((node_count + kOtfWaves - 1) / kOtfWaves),
                               dim3(kOtfWaves * 64), 0, stream, op);
        } else {
            const unsigned blocks = (unsigned)((total + kWalkThreads - 1) / kWalkThreads);
            if (L == 4)
                hipLaunchKernelGGL((merw_walk_kernel<PN_DRAW_PHILOX, true>), dim3(blocks), dim3(kWalkThreads), 0, stream, wp);
            else
                hipLaunchKernelGGL((merw_walk_kernel<PN_DRAW_PHILOX, false>), dim3(blocks), dim3(kWalkThreads), 0, stream, wp);
        }
        PN_CHECK_HIP(hipGetLastError());
    } else {
        PN_FAIL(PN_ERR_ARG, "unknown draw source %d", draw_source);
    }
    return PN_OK;
}

}  // extern "C"
