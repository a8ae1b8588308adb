// pn_internal.h -- shared by the translation units of libpathnet_hip.so (not part of the ABI).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/pathnet_hip.h"

namespace pn {

// thread-local error text behind pn_last_error()
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define PN_FAIL(code, ...)          \
    do {                            \
        ::pn::set_error(__VA_ARGS__); \
        return (code);              \
    } while (0)

#define PN_CHECK_HIP(expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) PN_FAIL(PN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- glibc rand() (TYPE_3 additive feedback) as polynomial algebra over Z/2^32 -------------------
// The word sequence obeys r[j+31] = r[j] + r[j+28]; a "state" is 31 consecutive words and
// x^d mod (x^31 - x^28 - 1) maps the state at position 0 to the state at position d.
struct GlibcPoly {
    uint32_t c[31];
};
struct GlibcState {
    uint32_t s[31];
};
GlibcPoly glibc_poly_one();
GlibcPoly glibc_poly_mul(const GlibcPoly &a, const GlibcPoly &b);
GlibcPoly glibc_poly_xpow(uint64_t d);
GlibcState glibc_apply(const GlibcPoly &p, const GlibcState &st);
// state whose s[0] >> 1 is the first rand() after srand(seed)
GlibcState glibc_seed_state(uint32_t seed);

// ---- kernel-selection knobs (A/B runs, tests): part of the context, read from the environment ONCE when the context is
// created (pn_context_create) and changed afterwards only through pn_context_set_knob -- no call reads the environment,
// so a concurrent setenv cannot race a launch and a captured step replays the selection it was captured with.
// A NULL context runs the defaults.
#ifndef PN_SEQH_TAIL_DEFAULT
#define PN_SEQH_TAIL_DEFAULT 0      // measured neutral at the headline shape (profiles/r05_tune_scatter_tiling.txt): off
#endif
struct Knobs {
    int node_gemm3 = 7;         // PN_NODE_GEMM3: bit mask -- 1 fc0, 2 distance bank, 4 dense bank dX on the bf16 x 3 GEMM when its
                                //                tiles fill the GPU; 8: the compact bank dX as well (measured slower)
    int eval_zw = 1;            // PN_EVAL_ZW: inference forwards apply W_ih to the bank rows before the gather
    int pool_bwd_wg = 1;        // PN_POOL_BWD_WG: pooling backward as a workgroup per node
    int small_side = 1;         // PN_SMALL_SIDE: loss sum / classifier gradient / attention reduce on the second stream under the BPTT
    int event_dev = 1;          // PN_EVENT_DEVICE_SCOPE: fork / join / timing events without the system-scope release of a recorded event (a change makes the events again)
    int zero_early = 1;         // PN_ZERO_EARLY: pn_pagg_train_step zero-fills the backward's accumulators on the second stream, under fc0 / bank
    int pool_step = 1;          // PN_POOL_STEP: pn_pagg_train_step runs pooling forward, loss and pooling backward of a node in one launch
    int node_rgrad = 1;         // PN_NODE_RGRAD: row-reduction kernel for the node-level weight gradients of large graphs
    int sampler_stage = -1;     // PN_SAMPLER_STAGE: first-hop tables in LDS (-1: by launch size)
    int seq4 = 4;               // PN_SEQ4: bit 2 = the two-stage weight-gradient GEMM of pn_seq4.hip serves the bf16 mode at hid 128
    int seqh_tail = PN_SEQH_TAIL_DEFAULT;   // PN_SEQH_TAIL: 0 = 32-path tiles only (default); 1 = the remainder round of the fp16
                                //                recurrent launches in smaller tiles, one per CU; 8 / 16 / 24 = that size, always
};
const Knobs &knobs_of(const pn_context *ctx);

// ---- pn_context: the only state that outlives a call (pn_context.hip) ------------------------------------
enum Stage {
    ST_SAMPLER_FILL = 0, ST_SAMPLER_WALK, ST_GATHER, ST_FC0, ST_BANK, ST_PLAN_PACK, ST_SEQ_FWD, ST_POOL_FWD,
    ST_FC2_GRAD, ST_POOL_BWD, ST_SEQ_BWD, ST_WGRAD, ST_BIAS_GRAD, ST_BANK_BWD, ST_FC0_BWD, ST_ZERO_FILL, ST_COUNT
};
// true while pn_profile_configure(ctx, 1, ...) brackets every stage: stages then run back to back on one stream
bool profiling_every_stage(const pn_context *ctx);
// ctx == nullptr is fine everywhere: no timing, no second stream.  Fails when the context belongs to another device.
int context_check_device(const pn_context *ctx);
// the context's second stream, forked from `stream` (event) -- nullptr when there is no context
void *context_fork(pn_context *ctx, void *stream);
void *context_side_stream(pn_context *ctx);      // the second stream itself (or null)
// records the join event on the second stream; context_join makes `stream` wait for it
int context_record_join(pn_context *ctx);
int context_join(pn_context *ctx, void *stream);
// hipFuncSetAttribute(kernel, MaxDynamicSharedMemorySize, bytes), remembered per context (= per device) so that the
// steady state issues no runtime call besides the launches (a captured step must not)
int ensure_dynamic_lds(pn_context *ctx, const void *kernel, int bytes);
// resident workgroups of `kernel` on the whole device (occupancy x compute units), remembered per context like the above
int resident_slots(pn_context *ctx, const void *kernel, int threads, size_t lds_bytes, int *slots, int *cus);
// RAII bracket: records a start event now and a stop event at scope exit when the context's profiling selects `stage`
struct StageTimer {
    StageTimer(pn_context *ctx, int stage, void *stream);
    ~StageTimer();
    pn_context *ctx;
    int slot;
    void *stream;
};

// softmax cross entropy of `rows` rows (pn_train.hip): loss[0] += scale * sum of the rows' losses, g_logits (or null) =
// (softmax - onehot) * scale.  pn_cross_entropy = zero-fill + this with scale = 1 / rows.
// overwrite: loss[0] = ... instead of += (the zero fill, when one is needed at all, is issued here)
int launch_cross_entropy(const float *logits, const int64_t *target, int rows, int classes, float scale, float *loss,
                         float *g_logits, void *stream, bool overwrite = false);

// ---- stable radix sort of (int32 key, int32 value) pairs (pn_sort.hip; the deterministic backward) ----------------
size_t sort_temp_reserve(int64_t n);        // bytes of temporary storage to reserve for n pairs (no device needed)
int sort_pairs_i32(void *tmp, size_t tmp_bytes, const int32_t *keys_in, int32_t *keys_out, const int32_t *vals_in,
                   int32_t *vals_out, int64_t n, int key_bits, void *stream);

}  // namespace pn
