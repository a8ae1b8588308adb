/*
 * merw_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * CPU restatement of the reference MERW path sampler, function by function, used as the
 * checker for the HIP sampler in pathnet_amd/csrc/.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.
 *
 * Parity pinning: the reference repo has no numeric golden vectors for this path
 * (SURVEY.md §8c).  This restatement is pinned instead against the *unmodified* reference
 * program compiled from /root/reference/preprocess/gen_merw.cpp into oracle/_ref/ and run
 * under a fixed-seed time() shim (tests/test_oracle_sampler.py), and against the fixtures in
 * tests/golden/ that were produced by that same binary (tests/golden/make_golden_sampler.py).
 *
 * Reference lines restated (all in /root/reference/preprocess/gen_merw.cpp):
 *   :95-99,  :162-172  edge list -> per-node neighbour / probability lists, file order
 *   :23-79             AliasTable::init  (two-queue construction, fp64, order sensitive)
 *   :81-91             AliasTable::roll  (two rand() draws per step)
 *   :101-123, :178-179 bfs() into dis[S][*] = 1 + hops, abandoned past seq_len
 *   :182-209           walk loop and the "[v0, .., d0, ..]\n" text line
 *   :161               srand(time(0)) + glibc rand()  (TYPE_3 additive feedback generator)
 * gen_epoch_merw.cpp:164-206 is the same stream split into one file per epoch.
 *
 * The Philox4x32-10 draw source is not in the reference; it restates the published
 * Random123 algorithm (Salmon et al., SC'11) with rocRAND's counter convention
 * (counter = {offset/4 lo, hi, subsequence lo, hi}, key = seed lo, hi) so the device
 * sampler's throughput mode has a CPU checker too.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define MO_DRAW_GLIBC 0
#define MO_DRAW_PHILOX 1
#define MO_RAND_MAX 2147483647

/* ------------------------------------------------------------------------------------------
 * glibc rand(): TYPE_3 generator (degree 31, separation 3).  srandom_r fills r[0..30] with a
 * Lehmer chain (16807 mod 2^31-1, Schrage form), then 310 outputs are discarded.  As one
 * infinite sequence: r[i] = r[i-31] for i in 31..33, r[i] = r[i-31] + r[i-3] (mod 2^32)
 * for i >= 34, and the k-th rand() result is r[k+344] >> 1.
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t r[34]; int pos; } mo_glibc_t;   /* ring of the last 34 words */

static void mo_glibc_seed(mo_glibc_t *g, uint32_t seed)
{
    uint32_t init[344 > 34 ? 34 : 34];
    int32_t word;
    int i;
    if (seed == 0) seed = 1;
    init[0] = seed;
    word = (int32_t)seed;
    for (i = 1; i < 31; i++) {
        long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        init[i] = (uint32_t)word;
    }
    for (i = 31; i < 34; i++) init[i] = init[i - 31];
    memcpy(g->r, init, sizeof init);
    g->pos = 0;                      /* r[pos] is the oldest word (index i-34) */
    /* words 34..343 are produced and thrown away */
    for (i = 34; i < 344; i++) {
        uint32_t v = g->r[(g->pos + 3) % 34] + g->r[(g->pos + 31) % 34];
        g->r[g->pos] = v;
        g->pos = (g->pos + 1) % 34;
    }
}

static inline int32_t mo_glibc_next(mo_glibc_t *g)
{
    /* new word i = word(i-31) + word(i-3); ring slot pos holds word(i-34) */
    uint32_t v = g->r[(g->pos + 3) % 34] + g->r[(g->pos + 31) % 34];
    g->r[g->pos] = v;
    g->pos = (g->pos + 1) % 34;
    return (int32_t)(v >> 1);
}

/* first `count` rand() values after srand(seed), skipping `skip` of them */
void mo_glibc_stream(uint32_t seed, uint64_t skip, uint64_t count, int32_t *out)
{
    mo_glibc_t g;
    uint64_t k;
    mo_glibc_seed(&g, seed);
    for (k = 0; k < skip; k++) (void)mo_glibc_next(&g);
    for (k = 0; k < count; k++) out[k] = mo_glibc_next(&g);
}

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Random123).  One call yields four 32-bit words for (key, counter).
 * ---------------------------------------------------------------------------------------- */
static void mo_philox4x32_10(uint32_t k0, uint32_t k1, const uint32_t ctr[4], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    int round;
    for (round = 0; round < 10; round++) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)m1;
        uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* draw q (0-based) of subsequence `sub` under `seed`: word (q & 3) of block q >> 2 */
uint32_t mo_philox_draw(uint64_t seed, uint64_t sub, uint64_t q)
{
    uint32_t ctr[4], out[4];
    ctr[0] = (uint32_t)(q >> 2); ctr[1] = (uint32_t)((q >> 2) >> 32);
    ctr[2] = (uint32_t)sub;      ctr[3] = (uint32_t)(sub >> 32);
    mo_philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), ctr, out);
    return out[q & 3];
}

/* ------------------------------------------------------------------------------------------
 * Alias tables, gen_merw.cpp:23-79.  For node i with k listed neighbours the probabilities are
 * scaled by k, split in input order into a "large" (> 1.0) and a "small" FIFO, and paired off.
 * Emits triples (A = large id, B = small id, S = small prob); when the leftover of the large
 * entry is within 1e-5 of 1 it is emitted as its own (id, id, leftover) triple and dropped,
 * otherwise it is re-queued at the tail of whichever FIFO its leftover belongs to.  Whatever
 * remains in either FIFO becomes (id, id, 1.0).  The triple count is NOT always k.
 *
 * off[n+1] receives the per-node prefix; A/B/S are filled up to `cap` triples.  Returns the
 * total triple count (call with cap = 0 to size the buffers), or -1 on bad input.
 * ---------------------------------------------------------------------------------------- */
int64_t mo_alias_build(int32_t n, int64_t m, const int32_t *u, const int32_t *v, const double *p,
                       int64_t *off, int32_t *A, int32_t *B, double *S, int64_t cap)
{
    int64_t *deg = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    int64_t *start = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    int32_t *nb = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m ? m : 1));
    double *pr = (double *)malloc(sizeof(double) * (size_t)(m ? m : 1));
    int64_t e, total = 0, maxdeg = 0;
    int32_t i;
    if (!deg || !start || !nb || !pr) return -1;
    for (e = 0; e < m; e++) {
        if (u[e] < 0 || u[e] >= n) { free(deg); free(start); free(nb); free(pr); return -1; }
        deg[u[e]]++;
    }
    for (i = 0; i < n; i++) { start[i + 1] = start[i] + deg[i]; if (deg[i] > maxdeg) maxdeg = deg[i]; }
    memset(deg, 0, sizeof(int64_t) * ((size_t)n + 1));
    for (e = 0; e < m; e++) {               /* stable bucket fill keeps file order per node */
        int64_t at = start[u[e]] + deg[u[e]]++;
        nb[at] = v[e]; pr[at] = p[e];
    }
    {
        /* FIFO storage: every pop re-pushes at most one entry, so 2*k+2 slots suffice */
        size_t qcap = (size_t)(2 * maxdeg + 4);
        int32_t *ida = (int32_t *)malloc(sizeof(int32_t) * qcap), *idb = (int32_t *)malloc(sizeof(int32_t) * qcap);
        double *pa = (double *)malloc(sizeof(double) * qcap), *pb = (double *)malloc(sizeof(double) * qcap);
        for (i = 0; i < n; i++) {
            int64_t k = start[i + 1] - start[i], j;
            size_t ha = 0, ta = 0, hb = 0, tb = 0;
            off[i] = total;
            for (j = 0; j < k; j++) {
                double s = pr[start[i] + j] * (double)k;       /* p[i] = p[i] * n */
                if (s > 1.0) { ida[ta] = nb[start[i] + j]; pa[ta++] = s; }
                else         { idb[tb] = nb[start[i] + j]; pb[tb++] = s; }
            }
#define MO_EMIT(a_, b_, s_) do { if (total < cap) { A[total] = (a_); B[total] = (b_); S[total] = (s_); } total++; } while (0)
            while (ha < ta && hb < tb) {
                int32_t big = ida[ha]; double pbig = pa[ha++];
                int32_t sml = idb[hb]; double psml = pb[hb++];
                double left = pbig - (1.0 - psml);
                MO_EMIT(big, sml, psml);
                if (fabs(left - 1.0) < 1e-5) { MO_EMIT(big, big, left); continue; }
                if (left > 1.0) { ida[ta] = big; pa[ta++] = left; }
                else            { idb[tb] = big; pb[tb++] = left; }
            }
            while (ha < ta) { int32_t id = ida[ha++]; MO_EMIT(id, id, 1.0); }
            while (hb < tb) { int32_t id = idb[hb++]; MO_EMIT(id, id, 1.0); }
#undef MO_EMIT
        }
        off[n] = total;
        free(ida); free(idb); free(pa); free(pb);
    }
    free(deg); free(start); free(nb); free(pr);
    return total;
}

/* ------------------------------------------------------------------------------------------
 * Distance codes, gen_merw.cpp:101-123.  Row S of `dis` (n x n bytes, zero-initialised by the
 * caller) gets 1 + hop count for every node the bounded BFS labels; the search stops as soon
 * as it pops a node whose label exceeds seq_len, so labels 1..seq_len+1 can appear.
 * ---------------------------------------------------------------------------------------- */
int mo_bfs_dense(int32_t n, int64_t m, const int32_t *u, const int32_t *v, int32_t seq_len, uint8_t *dis)
{
    int64_t *start = (int64_t *)calloc((size_t)n + 2, sizeof(int64_t));
    int64_t *fill = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    int32_t *nb = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m ? m : 1));
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    int64_t e;
    int32_t s;
    if (!start || !fill || !nb || !queue) return -1;
    for (e = 0; e < m; e++) start[u[e] + 1]++;
    for (s = 0; s < n; s++) start[s + 1] += start[s];
    for (e = 0; e < m; e++) nb[start[u[e]] + fill[u[e]]++] = v[e];
    for (s = 0; s < n; s++) {
        uint8_t *row = dis + (size_t)s * (size_t)n;
        int64_t head = 0, tail = 0;
        queue[tail++] = s;
        row[s] = 1;
        while (head < tail) {
            int32_t x = queue[head++];
            int64_t j;
            if (row[x] > seq_len) break;
            for (j = start[x]; j < start[x + 1]; j++) {
                int32_t y = nb[j];
                if (row[y] == 0) { row[y] = (uint8_t)(row[x] + 1); queue[tail++] = y; }
            }
        }
    }
    free(start); free(fill); free(nb); free(queue);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Walks, gen_merw.cpp:182-209 with roll() :81-91 inlined.
 *   draw index of (epoch e, source st, walk i, step t) = 2*(((e*n + st)*W + i)*L + t), +1 for
 *   the probability draw; the last roll of every walk is computed and discarded but still
 *   consumes its two draws.
 * MO_DRAW_GLIBC : draws are successive rand() values after srand(seed)  (bit-exact to _ref)
 * MO_DRAW_PHILOX: draw q of walk w is (philox(seed, sub = w, q) >> 1), w the global walk index
 * Outputs ids[e][st][i][t] (int32) and codes[...] = dis[st][u] - 1 (uint8).
 * Returns 0, or -2 if a walk reaches a node with an empty table (the reference exits there).
 * ---------------------------------------------------------------------------------------- */
int mo_walk(int32_t n, const int64_t *off, const int32_t *A, const int32_t *B, const double *S,
            const uint8_t *dis, int32_t W, int32_t L, int draw_source, uint64_t seed,
            int64_t epoch_begin, int64_t epoch_count, int32_t node_begin, int32_t node_count,
            int32_t *ids, uint8_t *codes)
{
    mo_glibc_t g;
    int64_t e, out = 0;
    uint64_t consumed = 0;
    if (draw_source == MO_DRAW_GLIBC) mo_glibc_seed(&g, (uint32_t)seed);
    for (e = epoch_begin; e < epoch_begin + epoch_count; e++) {
        int32_t st;
        for (st = node_begin; st < node_begin + node_count; st++) {
            int32_t i;
            for (i = 0; i < W; i++) {
                uint64_t walk = ((uint64_t)e * (uint64_t)n + (uint64_t)st) * (uint64_t)W + (uint64_t)i;
                int32_t x = st, t;
                if (draw_source == MO_DRAW_GLIBC) {      /* seek the sequential stream */
                    uint64_t want = walk * 2u * (uint64_t)L;
                    while (consumed < want) { (void)mo_glibc_next(&g); consumed++; }
                }
                for (t = 0; t < L; t++) {
                    int64_t len = off[x + 1] - off[x];
                    int32_t r0, r1;
                    int64_t slot;
                    double pp;
                    ids[out] = x;
                    codes[out] = (uint8_t)(dis[(size_t)st * (size_t)n + (size_t)x] - 1);
                    out++;
                    if (len == 0) return -2;
                    if (draw_source == MO_DRAW_GLIBC) {
                        r0 = mo_glibc_next(&g); r1 = mo_glibc_next(&g); consumed += 2;
                    } else {
                        r0 = (int32_t)(mo_philox_draw(seed, walk, 2u * (uint64_t)t) >> 1);
                        r1 = (int32_t)(mo_philox_draw(seed, walk, 2u * (uint64_t)t + 1u) >> 1);
                    }
                    slot = off[x] + (int64_t)(r0 % (int32_t)len);
                    pp = 1.0 * r1 / MO_RAND_MAX;
                    x = pp > S[slot] ? A[slot] : B[slot];
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * The uniform random-walk sampler ("RW-PathNet" ablation), /root/reference/preprocess/gen.cpp and
 * gen_epoch.cpp.  Same program as gen_merw.cpp except for the graph and the step:
 *   graph, gen.cpp:83-94: E[i] starts with the self loop i; every input pair (u, v) with u != v
 *     appends v to E[u] and u to E[v], in file order, duplicates kept.
 *     mo_uniform_build writes the lists as CSR (off[n+1], nbr[]); cap = 0 only sizes.  Returns the
 *     total length, or -1 on a node id outside [0, n).
 *   hop table, gen.cpp:18-40: the BFS of gen_merw.cpp over E  ->  mo_bfs_dense on the expanded lists.
 *   step, gen.cpp:113-114: ONE rand() per step, u = E[u][rand() % E[u].size()]; the last step of a
 *     walk still draws.  Draw index of (epoch e, source st, walk i, step t) = ((e*n + st)*W + i)*L + t.
 * ---------------------------------------------------------------------------------------- */
int64_t mo_uniform_build(int32_t n, int64_t m, const int32_t *u, const int32_t *v, int64_t *off,
                         int32_t *nbr, int64_t cap)
{
    int64_t i, total = 0;
    int64_t *cnt = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    if (!cnt) return -1;
    for (i = 0; i < n; i++) cnt[i] = 1;
    for (i = 0; i < m; i++) {
        if (u[i] < 0 || u[i] >= n || v[i] < 0 || v[i] >= n) { free(cnt); return -1; }
        if (u[i] == v[i]) continue;
        cnt[u[i]]++; cnt[v[i]]++;
    }
    for (i = 0; i < n; i++) total += cnt[i];
    if (cap >= total && off && nbr) {
        int64_t at = 0;
        for (i = 0; i < n; i++) { off[i] = at; at += cnt[i]; cnt[i] = off[i]; }
        off[n] = at;
        for (i = 0; i < n; i++) nbr[cnt[i]++] = (int32_t)i;
        for (i = 0; i < m; i++) {
            if (u[i] == v[i]) continue;
            nbr[cnt[u[i]]++] = v[i];
            nbr[cnt[v[i]]++] = u[i];
        }
    }
    free(cnt);
    return total;
}

int mo_walk_uniform(int32_t n, const int64_t *off, const int32_t *nbr, const uint8_t *dis, int32_t W,
                    int32_t L, int draw_source, uint64_t seed, int64_t epoch_begin, int64_t epoch_count,
                    int32_t node_begin, int32_t node_count, int32_t *ids, uint8_t *codes)
{
    mo_glibc_t g;
    int64_t e, out = 0;
    uint64_t consumed = 0;
    if (draw_source == MO_DRAW_GLIBC) mo_glibc_seed(&g, (uint32_t)seed);
    for (e = epoch_begin; e < epoch_begin + epoch_count; e++) {
        int32_t st;
        for (st = node_begin; st < node_begin + node_count; st++) {
            int32_t i;
            for (i = 0; i < W; i++) {
                uint64_t walk = ((uint64_t)e * (uint64_t)n + (uint64_t)st) * (uint64_t)W + (uint64_t)i;
                int32_t x = st, t;
                if (draw_source == MO_DRAW_GLIBC) {
                    uint64_t want = walk * (uint64_t)L;
                    while (consumed < want) { (void)mo_glibc_next(&g); consumed++; }
                }
                for (t = 0; t < L; t++) {
                    int64_t len = off[x + 1] - off[x];
                    int32_t r0;
                    ids[out] = x;
                    codes[out] = (uint8_t)(dis[(size_t)st * (size_t)n + (size_t)x] - 1);
                    out++;
                    if (len == 0) return -2;          /* cannot happen: every list holds its self loop */
                    if (draw_source == MO_DRAW_GLIBC) { r0 = mo_glibc_next(&g); consumed++; }
                    else r0 = (int32_t)(mo_philox_draw(seed, walk, (uint64_t)t) >> 1);
                    x = nbr[off[x] + (int64_t)(r0 % (int32_t)len)];
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Text line, gen_merw.cpp:189-206:  "[" v0 ", " v1 ", " ... v_{L-1} ", " d0 ", " ... d_{L-1} "]\n"
 * Writes npaths lines into buf (cap bytes); returns bytes written or -1 if cap is too small.
 * ---------------------------------------------------------------------------------------- */
int64_t mo_format_text(const int32_t *ids, const uint8_t *codes, int64_t npaths, int32_t L,
                       char *buf, int64_t cap)
{
    int64_t w = 0, p;
    for (p = 0; p < npaths; p++) {
        int32_t t;
        if (cap - w < 16 * 2 * (int64_t)L + 8) return -1;
        buf[w++] = '[';
        for (t = 0; t < L; t++) w += sprintf(buf + w, "%d, ", ids[p * L + t]);
        for (t = 0; t < L; t++) w += sprintf(buf + w, t + 1 < L ? "%d, " : "%d", (int)codes[p * L + t]);
        buf[w++] = ']'; buf[w++] = '\n';
    }
    return w;
}
