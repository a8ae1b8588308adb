/*
 * pathnet_hip.h -- C ABI of libpathnet_hip.so, the MI355X (gfx950) implementation of PathNet's
 * path-aggregation hot path: the MERW path sampler and the PAGG aggregator forward/backward.
 *
 * The reference (Sunefei/PathNet) has no FFI for this path: its two seams are a text file on disk
 * between the C++ sampler and the Python trainer, and a Python nn.Module surface.
 * Every entry point below names the reference code it stands in for.  The Python host side in
 * pathnet_amd/ binds these with ctypes; INTEGRATION.md shows the stubs.
 *
 * Conventions
 *   - plain C: pointers, sizes, PODs.  No exceptions cross the boundary.
 *   - every function returns an int status: PN_OK (0) or a negative PN_ERR_*; pn_last_error()
 *     returns a thread-local, human readable description of the last failure.
 *   - the caller owns every buffer (inputs, outputs, workspaces).  The library never frees or
 *     retains a pointer past the call.  "dev" in a comment = device (HBM) pointer, "host" = host.
 *   - device work is enqueued on the hipStream_t passed as `stream` (void* here so the header
 *     needs no HIP include; pass torch.cuda.current_stream().cuda_stream).  Device entry points
 *     are asynchronous with respect to the host.
 *   - the library has NO process-global state.  What outlives a call -- the second stream
 *     pn_pagg_forward/backward fork a few independent launches onto (joined back with events before
 *     they return, so all their work is ordered before whatever the caller enqueues on `stream`
 *     next) and the per-stage timing records -- lives in an opaque pn_context the caller creates for
 *     one device and destroys when done.  A context serialises nothing: two host threads may drive
 *     the same device through two contexts; one context must not be used by two threads at once.
 *     Every device entry point accepts ctx = NULL: everything then runs on `stream` alone, untimed.
 *   - the host-only entry points (files, tables) are re-entrant and use up to PN_HOST_THREADS threads.
 *   - sizes: n nodes, m edge rows, W walks per node (path_num), L path length, S masked nodes,
 *     P = S*W paths, H hidden size, F input features, C classes.
 */
#ifndef PATHNET_HIP_H_
#define PATHNET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_ABI_VERSION 10 /* 2: pn_sampler_tables gained draws_per_step; pn_pairs_*, pn_uniform_*, pn_merw_*, pn_cross_entropy, pn_adam_step
                          * 4: pn_context (no process-global state); pn_pagg_shape gained S_total / group_begin / batch_groups
                          *    (micro-batches, exact sharding of the hetero class); pn_pagg_args gained reuse_tables; 64-bit
                          *    offsets throughout; pn_clock_probe
                          * 5: pn_pagg_shape gained deterministic; pn_linear_backward gained workspace / workspace_bytes;
                          *    pn_pagg_train_step
                          * 6: pn_pagg_shape gained compact / seq_math (decisions that shape the workspace travel with the
                          *    shape, not with the environment); pn_pagg_args.reuse_tables = 2; pn_pagg_shape_info; pn_pagg_args gained
                          *    Xh_ready / g_Xh_ready (events that let the node-sharded path overlap its collectives)
                          * 7: pn_context_set_knob / pn_context_get_knob: the kernel-selection knobs are part of the context (read
                          *    from the environment once, when it is created); no call reads the environment any more
                          * 8: pn_seq_range / pn_pagg_range_offset (the fp16 recurrence's operand range and the spread of the gathered
                          *    rows' magnitudes, for callers that want to fall back to bf16x3 on pathological inputs)
                          * 9: pn_adam_step_advance (the optimizer's last launch moves the step state on)
                          * 10: pn_pagg_paths_stream (where a call reads its path arrays: a sampler enqueued there needs no event) */

#define PN_OK 0
#define PN_ERR_ARG (-1)          /* bad argument / unsupported shape */
#define PN_ERR_IO (-2)           /* file could not be opened / read / written */
#define PN_ERR_FORMAT (-3)       /* malformed text */
#define PN_ERR_HIP (-4)          /* a HIP runtime call or kernel launch failed */
#define PN_ERR_EMPTY_TABLE (-5)  /* a walk reached a node with no outgoing rows (gen_merw.cpp:84-87) */
#define PN_ERR_NOMEM (-6)
#define PN_ERR_CAPACITY (-7)     /* caller buffer too small; the required size is reported */

int pn_abi_version(void);
const char *pn_last_error(void);

/* GPU the library will launch on (current HIP device): name/arch and CU count, for bench reports. */
typedef struct pn_device_info {
    char name[128];
    char arch[64];
    int32_t compute_units;
    int32_t lds_bytes_per_block;
    int64_t hbm_bytes;
    int32_t clock_khz;
} pn_device_info;
int pn_device_query(pn_device_info *out);

/* Per-device, per-caller state: one internal non-blocking stream with its fork/join events and the records of
 * pn_profile_*.  Created on the current HIP device; every call that is handed the context must run with that
 * device current (checked).  pn_context_destroy waits for the context's own stream and releases it. */
typedef struct pn_context pn_context;
int pn_context_create(pn_context **out);
int pn_context_destroy(pn_context *ctx);
/* Kernel-selection knobs for A/B measurements and tests (none changes a result beyond rounding): PN_NODE_GEMM3, PN_EVAL_ZW,
 * PN_POOL_BWD_WG, PN_POOL_STEP, PN_ZERO_EARLY, PN_SMALL_SIDE, PN_EVENT_DEVICE_SCOPE, PN_NODE_RGRAD, PN_SAMPLER_STAGE, PN_SEQ4, PN_SEQH_TAIL (pn_internal.h: struct
 * Knobs).  A context
 * takes its values from the environment variables of the same names ONCE, in pn_context_create; afterwards only these two
 * calls read or change them -- device entry points never call getenv, so a captured step keeps its selection and a setenv
 * on another thread cannot race a launch.  get accepts ctx = NULL (the defaults a NULL context runs with). */
int pn_context_set_knob(pn_context *ctx, const char *name, int32_t value);
int pn_context_get_knob(const pn_context *ctx, const char *name, int32_t *value);

/* Per-step values kept in DEVICE memory, so that a whole training step captured into a hipGraph (every launch below is
 * capturable: no allocation, no synchronisation, the second stream joins the capture through its fork / join events) can
 * be replayed with fresh values: the sampler's epoch, the dropout seed and Adam's step count are read by the kernels
 * when they run, not baked into the launch.  pn_step_state_advance (one tiny launch, the first node of the graph) moves
 * to the next step: epoch += 1, adam_step += 1, seed = splitmix64(seed). */
typedef struct pn_step_state {
    int64_t epoch;      /* pn_sample_paths: epoch_begin */
    uint64_t seed;      /* pn_sample_paths (Philox seed) and pn_pagg_* (dropout seed) */
    int64_t adam_step;  /* pn_adam_step: the step count (from 1) */
    int64_t reserved;
} pn_step_state;
int pn_step_state_advance(pn_step_state *dev_state, void *stream);

/* Shader clock the device sustains right now (MHz), measured by a short kernel that reads the shader-cycle counter
 * (s_memtime) against the 100 MHz wall clock -- bench reports carry it so that box-to-box differences can be told
 * from code differences.  Synchronises `stream`. */
int pn_clock_probe(double *mhz, void *stream);

/* ================================================================================================
 * Sampler, host side.  Replaces the set-up half of preprocess/gen_merw.cpp main() (:162-179).
 * ============================================================================================== */

/* Edge file "<n> <m>\n" then m rows "u v p"  (gen_merw.cpp:162-172, gen_epoch_merw.cpp:145-155).
 * Call with cap = 0 to read only n and m; then with cap >= m to fill u, v, p (host, file order). */
int pn_edges_read_text(const char *path, int32_t *n, int64_t *m, int32_t *u, int32_t *v, double *p, int64_t cap);

/* ---- the uniform random-walk sampler ("RW-PathNet" ablation): preprocess/gen.cpp, gen_epoch.cpp ------------------
 * Pair file "<n> <m>" then m pairs "u v", read the way gen.cpp:80-92 reads it (scanf("%d%d")).  cap = 0 reads only
 * n and m; cap >= m fills u, v (host, file order). */
int pn_pairs_read_text(const char *path, int32_t *n, int64_t *m, int32_t *u, int32_t *v, int64_t cap);

/* The graph gen.cpp:83-94 walks on: list i starts with the self loop i, every pair with u != v then appends v to list
 * u and u to list v, in file order, repeated pairs kept.  off[n+1] prefixes the lists; packed[total*4] is the list
 * in the walker's 16-byte triple layout {nbr, nbr, 0, 0}; src/nbr (each [total], may be NULL) is the same graph as a
 * directed edge list for pn_hops_dense / pn_csr_build.  cap = 0 only sizes (*total). */
int pn_uniform_build(int32_t n, int64_t m, const int32_t *u, const int32_t *v, int64_t *off, int32_t *packed,
                     int32_t *src, int32_t *nbr, int64_t cap, int64_t *total);

/* Alias tables of every node, bit-identical to AliasTable::init (gen_merw.cpp:23-79) run on the
 * per-node lists that link() (:95-99) builds in file order.  off[n+1] is the per-node prefix of the
 * triple arrays.  A/B are the two candidate next nodes, S the fp64 split value, and
 * thr = the smallest 31-bit draw r for which (1.0 * r / RAND_MAX > S), so that roll() (:81-91)
 * becomes the integer test  r >= thr ? A : B  (thr = 2^31 when no draw qualifies).
 * Call with cap = 0 to get *total (and off[]); then with cap >= *total.  S may be NULL. */
int pn_alias_build(int32_t n, int64_t m, const int32_t *u, const int32_t *v, const double *p, int64_t *off,
                   int32_t *A, int32_t *B, double *S, uint32_t *thr, int64_t cap, int64_t *total);

/* Dense hop table dis[n*n] (host, uint8): dis[s*n + x] = 1 + (hops from s to x) for every x within
 * seq_len-1 hops of s, 0 otherwise -- the values bfs() (gen_merw.cpp:101-123) leaves in dis[][] for
 * every node a walk of length seq_len can visit.  The emitted distance code is dis - 1. */
int pn_hops_dense(int32_t n, int64_t m, const int32_t *u, const int32_t *v, int32_t seq_len, uint8_t *dis);

/* Sorted, duplicate-free adjacency in CSR form (host): for reverse = 0 the out-neighbours of every node
 * (row "u v p" puts v into list u), for reverse = 1 the in-neighbours.  Used instead of the dense hop
 * table when n*n bytes is too much (the reference's dis[N][N], gen_merw.cpp:10, caps n at 100050):
 * the walker then derives dis[st][x] - 1 exactly from these lists (see pn_sampler_tables).
 * Call with cap = 0 to get *count (and off[]); then with cap >= *count. */
int pn_csr_build(int32_t n, int64_t m, const int32_t *u, const int32_t *v, int32_t reverse, int64_t *off,
                 int32_t *adj, int64_t cap, int64_t *count);

/* The first `count` values rand() returns after srand(seed), starting at stream position `first`
 * (host, int32).  Uses the same jump algebra as the device generator (glibc TYPE_3 recurrence as
 * a polynomial over Z/2^32), so any window of the reference's draw stream (gen_merw.cpp:88-89) can
 * be produced without replaying the stream from the start. */
int pn_glibc_draws(uint32_t seed, uint64_t first, int64_t count, int32_t *out);

/* ================================================================================================
 * Sampler, device side.  Replaces the walk loop gen_merw.cpp:182-209 (and the per-epoch variant
 * gen_epoch_merw.cpp:164-206): one walk per GPU lane.
 * ============================================================================================== */

#define PN_DRAW_GLIBC_REPLAY 0 /* draws = glibc rand() after srand(seed): bit-exact to the reference binary */
#define PN_DRAW_PHILOX 1       /* draws = rocRAND Philox4x32-10, subsequence = global walk index (throughput) */

typedef struct pn_sampler_tables {
    int32_t n;               /* nodes */
    int64_t total;           /* alias triples */
    const int64_t *off;      /* dev [n+1]  prefix into the triple array */
    const int32_t *triples;  /* dev [total*4] packed {A, B, thr, 0} (one 16-byte load per roll) */
    const uint8_t *dis;      /* dev [n*n]  dense hop table from pn_hops_dense, or NULL: */
    /* dis == NULL selects on-the-fly hop codes (any n): exact hops(st -> x) by meeting in the middle of the
     * <=2-hop out-ball of the source (a hash table in LDS, built once per source node) and a bounded
     * search over the in-neighbours of x.  Needs both CSR lists from pn_csr_build (device copies). */
    const int64_t *adj_off;  /* dev [n+1] */
    const int32_t *adj;      /* dev out-neighbours, sorted per node */
    const int64_t *radj_off; /* dev [n+1] */
    const int32_t *radj;     /* dev in-neighbours, sorted per node */
    /* draws per walk step: 0 or 2 = the MERW alias roll (slot draw + probability draw, gen_merw.cpp:81-91);
     * 1 = the uniform sampler's single rand() % deg (gen.cpp:113-114) over tables from pn_uniform_build. */
    int32_t draws_per_step;
    /* dev [n][2] uint32 {first triple, number of triples} of every node (pn_node_ref_pack), or NULL: with it a roll
     * fetches a node's table position with ONE 8-byte load instead of two words of off[] (total < 2^32 triples). */
    const uint32_t *node_ref;
} pn_sampler_tables;

/* host: off[n+1] -> ref[n][2] = {off[i], off[i+1] - off[i]}.  Fails when the table holds 2^32 triples or more. */
int pn_node_ref_pack(int32_t n, const int64_t *off, uint32_t *ref);

/* Pack host A/B/thr arrays into the 16-byte device layout (host helper, dst is a host buffer of
 * total*4 int32 that the caller then copies to the device). */
int pn_alias_pack(int64_t total, const int32_t *A, const int32_t *B, const uint32_t *thr, int32_t *dst);

/* Bytes of device scratch pn_sample_paths needs for this window (0 for PN_DRAW_PHILOX). */
int pn_sample_workspace_bytes(int32_t W, int32_t L, int32_t draw_source, int64_t epoch_count, int32_t node_count,
                              int64_t *bytes);

/* Sample paths for epochs [epoch_begin, epoch_begin+epoch_count) and source nodes
 * [node_begin, node_begin+node_count) -- or, with node_list (dev int32 [node_count], PN_DRAW_PHILOX only: a walk's draws
 * are a function of (epoch, source node, walk index), whatever else is sampled beside it), the listed source nodes in
 * list order: a training step samples the paths of its masked nodes only (PathNet_run.py:345 indexes them out of the
 * epoch's file).  ids/codes are dev [epoch_count, node_count, W, L]
 * (int32 node ids, uint8 distance codes = dis - 1), i.e. exactly the numbers the reference prints,
 * in the reference's order.  The draw index of (epoch e, node st, walk i, step t) is
 * 2*(((e*n + st)*W + i)*L + t) (+1 for the probability draw), as in the reference where the last
 * roll of each walk is drawn and discarded (:195-196).
 * status_flag (dev int32, may be NULL) is set to PN_ERR_EMPTY_TABLE if any walk reaches a node with
 * an empty table (the reference exits there). */
int pn_sample_paths(pn_context *ctx, const pn_sampler_tables *tables, int32_t W, int32_t L, int32_t draw_source, uint64_t seed,
                    int64_t epoch_begin, int64_t epoch_count, int32_t node_begin, int32_t node_count, int32_t *ids,
                    uint8_t *codes, void *workspace, int64_t workspace_bytes, int32_t *status_flag,
                    const pn_step_state *step_state /* dev or NULL: PN_DRAW_PHILOX only; replaces seed / epoch_begin */,
                    const int32_t *node_list /* dev [node_count] or NULL */, void *stream);

/* ================================================================================================
 * Path file.  The on-disk interface between sampler and trainer: one line per path,
 * "[v0, v1, ..., v_{L-1}, d0, ..., d_{L-1}]\n"  (writer gen_merw.cpp:189-206; reader
 * PathNet_run.py:418-423 and :325-334, which requires the trailing "]\n").
 * ============================================================================================== */
int pn_paths_write_text(const char *path, const int32_t *ids, const uint8_t *codes, int64_t npaths, int32_t L,
                        int32_t append);
/* cap = 0: count lines only (ids/codes may be NULL).  Otherwise fills up to cap paths. */
int pn_paths_read_text(const char *path, int32_t L, int32_t *ids, uint8_t *codes, int64_t cap, int64_t *npaths);

/* Binary sidecar of the same content (the reference re-parses 10^7..10^8 text lines per run,
 * PathNet_run.py:418-423 / :325-334).  Layout: 32-byte header {char magic[8] = "PNPATHS1", int32 L, int32 reserved,
 * int64 npaths, int64 reserved}, then npaths*L int32 ids, then npaths*L uint8 codes.  Lossless w.r.t. the text. */
int pn_paths_write_bin(const char *path, const int32_t *ids, const uint8_t *codes, int64_t npaths, int32_t L);
/* cap = 0: read only the header (*npaths, *L_out).  Otherwise fills ids/codes (cap >= npaths). */
int pn_paths_read_bin(const char *path, int32_t *L_out, int32_t *ids, uint8_t *codes, int64_t cap, int64_t *npaths);

/* ================================================================================================
 * Aggregator ("PAGG"): the forward() of PathNet (PathNet_run.py:172-211), PathNet_homo (:239-278)
 * and PAGG (baseline/GPRGNN/src/copy.py:327-359), and its backward (autograd in the reference,
 * PathNet_run.py:351).
 * ============================================================================================== */
#define PN_VARIANT_HETERO 0 /* class PathNet      : LSTM, flip/time-major quirk, softmax(LeakyReLU) attention */
#define PN_VARIANT_HOMO 1   /* class PathNet_homo : LSTM, ReLU after fc0 and after the bank, (1+att) attention */
#define PN_VARIANT_PAGG 2   /* class PAGG         : tanh RNN, plain mean over paths */

typedef struct pn_pagg_shape {
    int32_t variant;
    int32_t N, F, H, C; /* nodes, input features, hidden (a multiple of 32: fused recurrent kernels up to 256, a
                         * step-by-step recurrence on the fp32 MFMA GEMM up to 1024), classes */
    int32_t S, W, L;    /* masked nodes this call aggregates (rows of out), paths per node, path length */
    /* A call may aggregate a slice of a larger batch: S_total masked nodes in the batch (0 means S), of which this
     * call computes the pooling groups [group_begin, group_begin + S).  ids / codes / sel / the explicit masks always
     * describe the WHOLE batch ([S_total, ...]); out / g_out are the S rows of the slice.  For HOMO / PAGG a group
     * only reads its own rows; HETERO's [W, S] re-view (PathNet_run.py:196-197) makes group g read paths of other
     * masked nodes of the batch -- with the whole batch's index arrays at hand a slice is still exactly the rows the
     * reference computes for the whole batch.  This is what node sharding across GPUs (pathnet_amd/dist.py) and the
     * micro-batches below rely on.  Dropout counters are functions of the position in the whole batch. */
    int32_t S_total, group_begin;
    /* > 0: the library walks the S groups in micro-batches of at most batch_groups groups, so that the per-path
     * tensors of the workspace are sized by batch_groups * W paths instead of S * W (configs with more paths than
     * HBM holds saved tensors for).  With more than one micro-batch the forward keeps no per-path tensors and the
     * backward re-runs each micro-batch's recurrence before its BPTT (same seed, same masks); gradients accumulate
     * across micro-batches and the node-level backward (bank, fc0) runs once at the end.  0: one batch. */
    int32_t batch_groups;
    /* The path encoder between the distance bank and the pooling.  0: the variant's own (LSTM for PathNet / PathNet_homo,
     * tanh RNN for PAGG).  The others are the ablation rows of the paper's table ("Changing the PAGG class can deliver
     * other variants", README.md:118; no code in the reference): GRU = torch.nn.GRU's cell (weights [3H, H], gate order
     * r, z, n), MEAN / SUM = the mean / the sum of a path's (dropped-out) step rows, no recurrent weights (w_* / b_* NULL). */
    int32_t cell;
    /* Non-zero: the backward adds in a fixed order -- bitwise identical gradients from run to run for the same inputs,
     * seed and shape (torch.use_deterministic_algorithms for this path).  The default backward scatters the gather's
     * gradient and splits the weight-gradient reductions with fp32 atomics, whose order the hardware picks; here the
     * scatter contributions are stored, sorted by destination row (stable radix sort) and summed in the order of the
     * path steps, and every split reduction stores its chunk sums and adds them in chunk order.  Costs workspace
     * (pn_pagg_workspace_bytes accounts for it: ~600 B per path step at H = 128) and time; the
     * forward computes the same values either way. */
    int32_t deterministic;
    /* Touched-row compaction of the distance bank (Z / dZ hold only the (node, code) rows this call's paths gather,
     * DESIGN.md section 3): 0 = the library decides from the shape (when the path steps cannot touch half of the N * L
     * rows), 1 = always, 2 = never.  The decision is part of the workspace layout, so pn_pagg_workspace_bytes, the
     * forward and the backward of one call must be given the same value. */
    int32_t compact;
    /* Arithmetic of the three recurrent GEMMs (hidden size <= 256; inputs, outputs and accumulation are fp32 either way):
     *   0 / PN_SEQ_MATH_F16X2: every fp32 product as THREE fp16 MFMAs over two-plane splits of both operands
     *       (a = a_hi + a_lo, |a_lo| <= 2^-12 |a|; a.b = a_hi b_hi + a_hi b_lo + a_lo b_hi + O(2^-24 |a||b|)), each operand
     *       scaled by a power of two taken from its largest magnitude so that the planes sit at the top of fp16's range;
     *   PN_SEQ_MATH_BF16X3: SIX bf16 MFMAs over three-plane splits, no scaling (rounds 1-3).
     * Both are as accurate as an fp32 FMA chain for operands whose magnitudes span less than ~2^16 below the largest
     * (DESIGN.md section 4); the bf16 mode keeps that for any span.  Explicit dropout masks must satisfy |m| <= 16. */
    int32_t seq_math;
} pn_pagg_shape;
#define PN_SEQ_MATH_DEFAULT 0
#define PN_SEQ_MATH_BF16X3 1
#define PN_SEQ_MATH_F16X2 2
#define PN_COMPACT_AUTO 0
#define PN_COMPACT_ON 1
#define PN_COMPACT_OFF 2
#define PN_CELL_DEFAULT 0
#define PN_CELL_LSTM 1
#define PN_CELL_RNN 2
#define PN_CELL_GRU 3
#define PN_CELL_MEAN 4
#define PN_CELL_SUM 5

/* All pointers are device pointers.  Weights use the reference state_dict layout
 * (nn.Linear weight [out, in]; LSTM/RNN weight_ih/hh [G*H, H] with torch gate order i,f,g,o).
 * Gradient pointers (g_*) may be NULL in forward; in backward a NULL gradient is skipped. */
typedef struct pn_pagg_args {
    pn_pagg_shape shape;
    /* inputs */
    const float *X;       /* [N, F] */
    const int32_t *ids;   /* [S_total, W, L] path node ids of the whole batch, path-major as in the path file */
    const uint8_t *codes; /* [S_total, W, L] distance codes */
    const int32_t *sel;   /* [S_total] node index of each masked node (nonzero(indices));
                           * with index_rows_local: the slice's rows only ([S, ...]) */
    /* parameters */
    const float *fc0_w, *fc0_b;   /* [H, F], [H] */
    const float *bank_w, *bank_b; /* [L, H, H], [L, H]   nets.<d> / nei<d> stacked by d */
    const float *w_ih, *w_hh;     /* [G*H, H]            G = 4 (LSTM), 1 (RNN), 3 (GRU); NULL for the mean / sum cells */
    const float *b_ih, *b_hh;     /* [G*H] */
    const float *att_w, *att_b;   /* [2H], [1]           unused for PN_VARIANT_PAGG */
    const float *fc2_w, *fc2_b;   /* [C, 2H], [C] */
    /* dropout (F.dropout(training=True) on the recurrent input [L,P,H] and on the classifier input
     * [S,2H]): p = 0 disables.  With mask pointers set, those multiplicative masks (already scaled by
     * 1/(1-p)) are used instead of the built-in Philox masks -- this is how parity tests inject the
     * reference's mask. */
    float p_seq, p_cls;
    uint64_t seed;
    const float *mask_seq; /* [L, S_total*W, H] (positions in the whole batch) or NULL */
    const float *mask_cls; /* [S_total, 2H] or NULL */
    /* outputs */
    float *out; /* [S, C] logits of the slice's masked nodes */
    /* saved-for-backward + scratch, sized by pn_pagg_workspace_bytes */
    void *workspace;
    int64_t workspace_bytes;
    /* backward only */
    const float *g_out; /* [S, C] upstream gradient */
    float *g_X;         /* [N, F] or NULL */
    float *g_fc0_w, *g_fc0_b, *g_bank_w, *g_bank_b, *g_w_ih, *g_w_hh, *g_b_ih, *g_b_hh, *g_att_w, *g_att_b,
        *g_fc2_w, *g_fc2_b;
    /* node-sharded multi-GPU use (pathnet_amd/dist.py): when Xh_in is set the projected feature matrix
     * Xh [N, H] (fc0 output, after ReLU for HOMO) is taken from the caller -- every rank projects its
     * own rows with pn_gemm_f32 and the rows are all-gathered over RCCL -- and X / fc0_* are ignored.
     * In backward g_Xh [N, H] then receives d loss / d Xh (to be reduce-scattered to the row owners and
     * finished with pn_linear_backward) instead of g_fc0_* / g_X. */
    const float *Xh_in;
    float *g_Xh;
    /* Overlap of those two collectives with the library's own work (both may be NULL; hipEvent_t handles, not captured into
     * graphs).  Xh_ready: recorded by the caller on ITS communication stream after the all-gather -- the forward makes `stream`
     * wait for it only where Xh_in is first read (the distance bank), so the touched-row marking, the index plan and the
     * weight packing run under the all-gather.  g_Xh_ready: recorded by the backward on `stream` as soon as g_Xh is complete
     * -- before the distance bank's and the recurrent weight gradients, which then run under the caller's reduce-scatter. */
    void *Xh_ready;
    void *g_Xh_ready;
    /* inference: non-zero skips writing the saved-for-backward tensors (gates, cell states, [x|h] rows --
     * ~3.6 KB per path step); pn_pagg_backward must not follow such a forward. */
    int32_t no_save;
    /* 1: Xh AND the dense Z = bank(Xh) [N * L, H] of the previous pn_pagg_forward on this workspace are still valid (same
     * X, same fc0 / bank weights, same N, H, L, and that forward was NOT compact): skip fc0 and the bank.
     * 2: only Xh is valid (the previous forward ran over compact rows, whose Z belongs to its own batch): skip fc0,
     * compute the bank.  A compact call always computes its own rows of the bank, whatever the value.
     * The validation and test forwards of an epoch (PathNet_run.py:362, :378) share the tables this way; the caller
     * vouches for the precondition -- pn_pagg_shape_info tells whether a shape runs compact. */
    int32_t reuse_tables;
    /* non-zero: ids / codes / sel hold only the rows of this call's slice ([S, W, L] / [S]: rows group_begin ..
     * group_begin + S of the batch) instead of the whole batch.  HOMO / PAGG only (a group reads nothing but its own
     * rows there); positions in the batch -- dropout counters, explicit masks -- stay batch-wide.  This is what a rank
     * of the node-sharded path passes: its own paths, no exchange of index arrays. */
    int32_t index_rows_local;
    /* dev or NULL: the dropout seed is step_state->seed, read when the kernels run (hipGraph replay), instead of `seed` */
    const pn_step_state *step_state;
} pn_pagg_args;

int pn_pagg_workspace_bytes(const pn_pagg_shape *shape, int64_t *bytes);
/* What the library decides from a shape: out[0] = 1 when the call runs over compact rows of the distance bank, out[1] =
 * rows of Z / dZ, out[2] = micro-batches, out[3] = PN_SEQ_MATH_* the recurrent GEMMs will use (0: the shape has none). */
int pn_pagg_shape_info(const pn_pagg_shape *shape, int64_t out[4]);
/* The operand ranges of the fp16 x 2 recurrent kernels (seq_math = PN_SEQ_MATH_F16X2) live in the workspace, in device memory,
 * written by the forward's kernels.  x_bits: bit pattern of max |Z| over the distance-bank rows the call gathers from (the
 * power-of-two scale of the gathered rows is taken from it).  x_esum / x_cnt: sum and count of the biased fp32 exponents of the
 * non-zero maxima of a sample of tiles of Z, so that  ((x_bits >> 23) & 255) - x_esum / x_cnt  is how far, in bits, the largest
 * value sits above the typical tile.  The two-plane fp16 split keeps 22 bits of a row within ~2^18 of the maximum and loses one
 * bit per factor of two below that: a caller that sees a spread beyond ~18 bits (a single spike row 2^20 times the typical one)
 * should run the call with seq_math = PN_SEQ_MATH_BF16X3, whose planes carry fp32's own exponent range.  The library itself never
 * reads x_esum / x_cnt (no host round trip in a call); pathnet_amd/modules.py reads them back asynchronously. */
typedef struct pn_seq_range {
    uint32_t x_bits;
    int32_t x_esum;
    uint32_t x_cnt;
    uint32_t w_ih_bits, w_hh_bits, dg_bits;     /* max |W_ih|, max |W_hh|, max |dG| of the last BPTT */
} pn_seq_range;
/* *offset = byte offset of the call's pn_seq_range in its workspace, or -1 when the shape runs no fp16 recurrent kernel */
int pn_pagg_range_offset(const pn_pagg_shape *shape, int64_t *offset);
/* The stream on which a call with this shape, made on `stream`, READS its path arrays (ids / codes: the index plan and the
 * touched-row marks): the context's second stream when the call forks them off (a context, one micro-batch, stages not timed one
 * by one), else `stream` itself.  A producer of the paths enqueued THERE -- the sampler of a training loop: pn_sample_paths with
 * *out as its stream, right before the call -- is ordered before its readers and behind those of the previous call without an
 * event on `stream`, and runs under whatever that stream still holds of the previous step (the recurrent weight gradient):
 * the walk leaves the step's critical path.  Ask before every call (the answer follows pn_profile_configure). */
int pn_pagg_paths_stream(pn_context *ctx, const pn_pagg_shape *shape, void *stream, void **out);
/* out = forward(...).  Leaves what backward needs in the workspace. */
int pn_pagg_forward(pn_context *ctx, const pn_pagg_args *args, void *stream);
/* Gradients of sum(out * g_out) w.r.t. every parameter (overwritten, not accumulated) and X.
 * Must follow a pn_pagg_forward (no_save = 0) on the same args/workspace with the same weights: the saved tensors, the index
 * plan and -- fp16 mode -- the weights' operand ranges and their packed planes for the BPTT (written by the forward's packing
 * stream, off the backward's critical path) live there. */
int pn_pagg_backward(pn_context *ctx, const pn_pagg_args *args, void *stream);
/* One call for the aggregator's part of a training step (PathNet_run.py:343-351: forward, CrossEntropyLoss, backward):
 *   out [S, C] = forward(...);  loss[0] = grad_scale * sum over the S rows of softmax-cross-entropy(out[r], target[r]);
 *   every g_* of args = d loss / d parameter (g_out is not read).
 * grad_scale = 1 / S gives torch.nn.CrossEntropyLoss()'s mean (a rank of the node-sharded step passes 1 / S_total).
 * target: dev int64 [S] class indices of the slice's rows; loss: dev float.  Same values as pn_pagg_forward +
 * pn_cross_entropy + pn_pagg_backward; what it saves is the second forward of every micro-batch: with batch_groups > 0
 * pn_pagg_forward keeps no per-path tensors and pn_pagg_backward has to re-run each micro-batch's recurrence, here a
 * micro-batch's forward, loss gradient and backward follow each other on the tensors still in the workspace. */
int pn_pagg_train_step(pn_context *ctx, const pn_pagg_args *args, const int64_t *target, float grad_scale, float *loss,
                       void *stream);

/* Stand-alone stages of the same path, exposed for measurement and tests. */
/* rows[q, t, :] = table[(node(q,t) * L + code(q,t)), :] for the variant's index plan: the
 * [P, L, H] path-feature gather with the distance code fused in (PathNet_run.py:179 / :246). */
int pn_pagg_gather(pn_context *ctx, const pn_pagg_shape *shape, const float *table /* [N, L, H] */, const int32_t *ids,
                   const uint8_t *codes, float *rows /* [P, L, H] */, void *stream);

/* C[m*ldc + n] = act( sum_k A[m*sAm + k*sAk] * B[n*sBn + k*sBk] + bias[n] ): the fp32 MFMA GEMM
 * every dense layer of the path uses (nn.Linear: sAk = sBk = 1).  One stride of each operand must
 * be 1.  bias may be NULL; relu != 0 applies max(0, .). */
int pn_gemm_f32(const float *A, int64_t sAm, int64_t sAk, const float *B, int64_t sBn, int64_t sBk, float *C,
                int64_t ldc, const float *bias, int32_t M, int32_t N, int32_t K, int32_t relu, void *stream);

/* Y [rows, out_f] = act(X [rows, in_f] . W^T + b): nn.Linear (+ReLU when relu != 0), fc0 of the path for a block of
 * rows (PathNet_run.py:175 / :242) -- what a rank computes before the all-gather of the node-sharded path.  With a
 * device workspace of at least PN_LINEAR_SPLIT_MAX * rows * out_f * 4 bytes the reduction dimension is split over
 * workgroups (deterministic: chunk sums + a fixed-order finish), which is what fills the GPU when rows * out_f is
 * small; workspace may be NULL (one workgroup per output tile). */
#define PN_LINEAR_SPLIT_MAX 8
int pn_linear_forward(pn_context *ctx, const float *X, const float *W, const float *b, int32_t rows, int32_t in_f, int32_t out_f,
                      int32_t relu, float *Y, void *workspace, int64_t workspace_bytes, void *stream);

/* Backward of Y = act(X . W^T + b) for `rows` rows (nn.Linear, fc0 of the path: PathNet_run.py:175 / :242):
 * dY [rows, out_f] is gated by [gate > 0] when gate != NULL (ReLU backward, gate = Y), then
 * g_W [out_f, in_f] = dY^T . X,  g_b [out_f] = colsum(dY),  g_X [rows, in_f] = dY . W.  Any output may be NULL.
 * The weight gradient splits the `rows` reduction over workgroups.  With a device workspace of at least
 * PN_LINEAR_BWD_SPLIT_MAX * (out_f * in_f + out_f) * 4 bytes the chunk sums are stored and added in a fixed order --
 * bitwise reproducible, what the deterministic mode of the aggregator (pn_pagg_shape.deterministic) pairs with on the
 * node-sharded path; with workspace NULL the chunks are added with fp32 atomics (last bits vary from run to run). */
#define PN_LINEAR_BWD_SPLIT_MAX 32
int pn_linear_backward(pn_context *ctx, const float *dY, const float *gate, const float *X, const float *W, int32_t rows, int32_t in_f,
                       int32_t out_f, float *g_W, float *g_b, float *g_X, void *workspace, int64_t workspace_bytes, void *stream);

/* ---- the MERW transition probabilities: what writes edge_input/<name>.in ----------------------
 * preprocess/compute_merw.py:107-121 compute_merw(A), as init_rw.py:76 calls it: (lambda, psi) = dominant eigenpair of
 * the symmetric adjacency matrix, p[k] = A[i,j] psi[j] / (lambda psi[i]) for every stored entry k = (i, j).
 * A is CSR on the device (row_off [n+1], col [nnz], val [nnz] or NULL for all ones).  Power iteration on A + I in fp64,
 * deterministic; converged when |A x - lambda x| <= tol (lambda + 1) (tol <= 0: 1e-13; max_iter < 1: 100000).
 * Outputs: p dev [nnz], psi dev [n] (unit 2-norm, positive), *lambda and *iters on the host.  Fails with the residual in
 * pn_last_error() when it does not converge.  Call it per CONNECTED component: on a disconnected graph the reference's one
 * eigenpair is exact on the dominant component only and eigensolver noise elsewhere (the negative and > 1 "probabilities" of
 * the shipped cora.in / citeseer.in); pathnet_amd/merw_init.py splits the graph and documents what it writes there. */
int pn_merw_workspace_bytes(int32_t n, int64_t *bytes);
int pn_merw_probabilities(int32_t n, int64_t nnz, const int64_t *row_off, const int32_t *col, const double *val,
                          double *p, double *psi, double *lambda, int32_t max_iter, double tol, int32_t *iters,
                          void *workspace, int64_t workspace_bytes, void *stream);

/* ---- training-step glue: the loss and the optimizer of PathNet_run.py:295-297, :346-352 ------
 * Mean softmax cross entropy over `rows` rows of `classes` logits (torch.nn.CrossEntropyLoss(), :297; int64 class
 * targets, no ignore_index / label smoothing -- the reference uses neither): *loss (device float) receives the mean,
 * g_logits [rows, classes] (may be NULL) d loss / d logits = (softmax - onehot) / rows. */
int pn_cross_entropy(const float *logits, const int64_t *target, int32_t rows, int32_t classes, float *loss,
                     float *g_logits, void *stream);

/* One Adam update of a list of tensors in a single launch, with the semantics of torch.optim.Adam(lr, betas, eps,
 * weight_decay) (:295-296): g += weight_decay * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 * p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).  `step` counts from 1.  All pointers are device
 * fp32 arrays of `count` elements; exp_avg / exp_avg_sq are the caller-owned optimizer state (zero before step 1). */
#define PN_ADAM_MAX_TENSORS 32
typedef struct pn_adam_tensor {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t count;
} pn_adam_tensor;
int pn_adam_step(const pn_adam_tensor *tensors, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int64_t step, const pn_step_state *step_state /* dev or NULL: replaces step */,
                 void *stream);
/* The same update with step_state's step count, after which the state is moved to the NEXT step by the update's own last
 * launch (the workgroup that finishes last does what pn_step_state_advance does): a training loop whose steps end in this call
 * needs no pn_step_state_advance launch in front of the next one -- ~7 us of a 0.93 ms step at the headline shape.  Uses the
 * low 32 bits of step_state->reserved as a ticket counter (zero between calls). */
int pn_adam_step_advance(const pn_adam_tensor *tensors, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                         float weight_decay, pn_step_state *step_state /* dev */, void *stream);

/* Byte offsets inside the aggregator workspace of the intermediates tests look at:
 * out[0] Xh [N,H], out[1] Z [N,L,H], out[2] hn [P,H] (pooling-group order), out[3] layer1 [S,2H]. */
int pn_pagg_debug_offsets(const pn_pagg_shape *shape, int64_t out[4]);

/* ================================================================================================
 * Per-stage timing with HIP events recorded on the caller's stream (measurement only; state in the context).
 * mode 0: off (default).  mode 1: bracket every stage.  mode 2: bracket only stage `stage`.
 * pn_profile_read waits for the recorded events, adds the elapsed times into ms_sum[i] / count[i]
 * (arrays of pn_profile_stage_count() entries) and clears the recording.
 * ============================================================================================== */
int pn_profile_configure(pn_context *ctx, int32_t mode, int32_t stage);
int pn_profile_stage_count(void);
const char *pn_profile_stage_name(int32_t stage);
int pn_profile_read(pn_context *ctx, double *ms_sum, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* PATHNET_HIP_H_ */
