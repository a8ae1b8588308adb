// pn_host.cpp -- host half of libpathnet_hip.so: everything the sampler needs before the GPU walks
// (edge file, alias tables, hop table, glibc stream algebra) and the text path-file reader/writer.
//
// Reference behaviour reproduced (file:line in /root/reference):
//   preprocess/gen_merw.cpp:162-172, :95-99   edge rows -> per-node lists in file order
//   preprocess/gen_merw.cpp:23-79            AliasTable::init
//   preprocess/gen_merw.cpp:81-91            AliasTable::roll  (here: its fp64 compare turned into
//                                            an exact integer threshold per triple)
//   preprocess/gen_merw.cpp:101-123          bfs() / dis[][]
//   preprocess/gen_merw.cpp:189-206          text line format;  PathNet_run.py:418-423 reader
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "pn_internal.h"

namespace pn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------------
// glibc TYPE_3 generator algebra
// ------------------------------------------------------------------------------------------------
GlibcPoly glibc_poly_one() {
    GlibcPoly p{};
    p.c[0] = 1;
    return p;
}

GlibcPoly glibc_poly_mul(const GlibcPoly &a, const GlibcPoly &b) {
    uint32_t t[61] = {0};
    for (int i = 0; i < 31; i++)
        for (int j = 0; j < 31; j++) t[i + j] += a.c[i] * b.c[j];
    // x^k = x^(k-3) + x^(k-31) for k >= 31
    for (int k = 60; k >= 31; k--) {
        t[k - 3] += t[k];
        t[k - 31] += t[k];
    }
    GlibcPoly r;
    std::memcpy(r.c, t, sizeof r.c);
    return r;
}

GlibcPoly glibc_poly_xpow(uint64_t d) {
    GlibcPoly result = glibc_poly_one();
    GlibcPoly base{};
    base.c[1] = 1;
    while (d) {
        if (d & 1) result = glibc_poly_mul(result, base);
        base = glibc_poly_mul(base, base);
        d >>= 1;
    }
    return result;
}

GlibcState glibc_apply(const GlibcPoly &p, const GlibcState &st) {
    uint32_t w[61];
    std::memcpy(w, st.s, sizeof st.s);
    for (int j = 31; j < 61; j++) w[j] = w[j - 31] + w[j - 3];
    GlibcState out;
    for (int m = 0; m < 31; m++) {
        uint32_t acc = 0;
        for (int k = 0; k < 31; k++) acc += p.c[k] * w[m + k];
        out.s[m] = acc;
    }
    return out;
}

GlibcState glibc_seed_state(uint32_t seed) {
    // srandom_r: r[0] = seed (0 -> 1), r[i] = 16807 * r[i-1] mod (2^31 - 1) via Schrage, i < 31;
    // r[31..33] = r[0..2]; from i = 34 on r[i] = r[i-31] + r[i-3].  rand() number k is r[344+k] >> 1.
    uint32_t r[34];
    if (seed == 0) seed = 1;
    r[0] = seed;
    int32_t word = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    GlibcState base;  // window r[3..33]: its successor is r[34] = r[3] + r[31] = s[0] + s[28]
    for (int i = 0; i < 31; i++) base.s[i] = r[3 + i];
    // advance the window so that s[0] = r[344]
    return glibc_apply(glibc_poly_xpow(341), base);
}

}  // namespace pn

using namespace pn;

// Host threads for the host-side loops (PN_HOST_THREADS overrides; never more than one per `grain` items).
static int host_threads(int64_t items, int64_t grain) {
    const char *e = std::getenv("PN_HOST_THREADS");
    const int env = e ? std::atoi(e) : 0;
    int64_t t = env > 0 ? env : (int64_t)std::thread::hardware_concurrency();
    t = std::max<int64_t>(1, std::min<int64_t>(t, 32));
    return (int)std::max<int64_t>(1, std::min<int64_t>(t, items / std::max<int64_t>(grain, 1)));
}

// fn(0..T-1) on T threads (the caller is thread 0); a thread that cannot be created runs inline.  An exception in
// any of them (std::bad_alloc, in practice) is re-thrown here once all have finished.
template <class F>
static void run_threads(int T, F &&fn) {
    std::exception_ptr failure;
    std::mutex mu;
    auto guarded = [&](int i) {
        try {
            fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> lock(mu);
            if (!failure) failure = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(T > 1 ? T - 1 : 0);
    for (int i = 1; i < T; i++) {
        try {
            pool.emplace_back([&guarded, i] { guarded(i); });
        } catch (const std::system_error &) {
            guarded(i);
        }
    }
    guarded(0);
    for (auto &th : pool) th.join();
    if (failure) std::rethrow_exception(failure);
}

// No exception crosses the C boundary: every entry point below is a function-try-block ending here.
static int host_exception(const char *where) {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        PN_FAIL(PN_ERR_NOMEM, "%s: out of host memory", where);
    } catch (const std::exception &e) {
        PN_FAIL(PN_ERR_IO, "%s: %s", where, e.what());
    } catch (...) {
        PN_FAIL(PN_ERR_IO, "%s: unknown failure", where);
    }
}

// read-only view of a file through the page cache (no copy); an empty file maps to an empty view
struct MappedFile {
    const char *data = nullptr;
    size_t size = 0;
    bool open(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (::fstat(fd, &st) != 0) {
            const int e = errno;
            ::close(fd);
            errno = e;
            return false;
        }
        if (st.st_size > 0) {
            void *m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                const int e = errno;
                ::close(fd);
                errno = e;
                return false;
            }
            data = static_cast<const char *>(m);
            size = (size_t)st.st_size;
        }
        ::close(fd);
        return true;
    }
    ~MappedFile() {
        if (data) ::munmap(const_cast<char *>(data), size);
    }
};

// ---- whitespace-delimited number files (edge_input/<name>.in) on all host threads ---------------------------------
// The reference reads them with scanf("%d%d%lf") (gen_merw.cpp:162-172), i.e. as a stream of tokens.  The fast
// path cuts the mapped file into byte ranges at token boundaries, counts the tokens of each range, and parses the
// ranges concurrently (token k is field k % NF of row k / NF).  Anything it does not expect -- a token that is not
// entirely one number, too few tokens -- makes it give up, and the sequential scanf-equivalent reader below decides
// (and words the error), so results and failures are those of the sequential reader.
static inline bool is_ws(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

static int64_t count_tokens(const char *s, const char *e) {
    int64_t n = 0;
    bool in = false;
    for (; s < e; s++) {
        const bool w = is_ws((unsigned char)*s);
        n += (!w && !in);
        in = !w;
    }
    return n;
}

static bool parse_int_token(const char *t, const char *e, int64_t *out) {
    bool neg = false;
    if (t < e && (*t == '-' || *t == '+')) neg = *t++ == '-';
    if (t >= e || e - t > 18) return false;
    int64_t val = 0;
    for (; t < e; t++) {
        if (*t < '0' || *t > '9') return false;
        val = val * 10 + (*t - '0');
    }
    *out = neg ? -val : val;
    return true;
}

// tokens of [s, e); the first one has global index k0.  NF = 3: int int double rows, NF = 2: int int rows.
template <int NF>
static bool parse_number_rows(const char *s, const char *e, int64_t k0, int64_t rows, int32_t *u, int32_t *v, double *p) {
    int64_t k = k0;
    const int64_t kend = rows * NF;
    while (k < kend) {
        while (s < e && is_ws((unsigned char)*s)) s++;
        if (s >= e) break;
        const char *t = s;
        while (s < e && !is_ws((unsigned char)*s)) s++;
        const int f = (int)(k % NF);
        const int64_t row = k / NF;
        if (NF == 3 && f == 2) {
            char tmp[64];
            const size_t len = (size_t)(s - t);
            if (len >= sizeof tmp) return false;
            std::memcpy(tmp, t, len);
            tmp[len] = 0;
            char *end = nullptr;
            const double d = std::strtod(tmp, &end);     // same conversion scanf("%lf") performs
            if (end != tmp + len) return false;
            p[row] = d;
        } else {
            int64_t val;
            if (!parse_int_token(t, s, &val)) return false;
            (f == 0 ? u : v)[row] = (int32_t)val;
        }
        k++;
    }
    return true;
}

// -> 1 header parsed (n, m set; rows filled when cap > 0), 0 give up (caller runs the sequential reader)
template <int NF>
static int read_number_file_fast(const char *path, int32_t *n, int64_t *m, int32_t *u, int32_t *v, double *p, int64_t cap) {
    MappedFile f;
    if (!f.open(path)) return 0;
    const char *s = f.data, *e = s + f.size;
    int64_t hdr[2];
    for (int h = 0; h < 2; h++) {
        while (s < e && is_ws((unsigned char)*s)) s++;
        const char *t = s;
        while (s < e && !is_ws((unsigned char)*s)) s++;
        if (!parse_int_token(t, s, &hdr[h]) || hdr[h] < 0) return 0;
    }
    if (hdr[0] > 2147483647LL) return 0;
    if (cap == 0) {
        *n = (int32_t)hdr[0];
        *m = hdr[1];
        return 1;
    }
    if (cap < hdr[1] || !u || !v || (NF == 3 && !p)) return 0;
    const int T = host_threads(e - s, 1 << 20);
    std::vector<const char *> cut((size_t)T + 1, e);
    cut[0] = s;
    for (int i = 1; i < T; i++) {
        const char *c = std::max(cut[i - 1], s + (e - s) / T * i);
        while (c < e && !is_ws((unsigned char)*c)) c++;
        cut[i] = c;
    }
    std::vector<int64_t> first((size_t)T + 1, 0);
    run_threads(T, [&](int i) { first[i + 1] = count_tokens(cut[i], cut[i + 1]); });
    for (int i = 0; i < T; i++) first[i + 1] += first[i];
    if (first[T] < hdr[1] * NF) return 0;      // truncated: the sequential reader names the row
    std::atomic<bool> ok(true);
    run_threads(T, [&](int i) {
        if (!parse_number_rows<NF>(cut[i], cut[i + 1], first[i], hdr[1], u, v, p)) ok = false;
    });
    if (!ok) return 0;
    *n = (int32_t)hdr[0];
    *m = hdr[1];
    return 1;
}

extern "C" {

int pn_abi_version(void) { return PN_ABI_VERSION; }
const char *pn_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// edge file
// ------------------------------------------------------------------------------------------------
static bool slurp(const char *path, std::string &out) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = sz > 0 ? std::fread(&out[0], 1, (size_t)sz, f) : 0;
    std::fclose(f);
    return got == out.size();
}

int pn_edges_read_text(const char *path, int32_t *n, int64_t *m, int32_t *u, int32_t *v, double *p, int64_t cap) try {
    if (!path || !n || !m) PN_FAIL(PN_ERR_ARG, "pn_edges_read_text: null argument");
    if (read_number_file_fast<3>(path, n, m, u, v, p, cap)) return PN_OK;
    std::string txt;
    if (!slurp(path, txt)) PN_FAIL(PN_ERR_IO, "cannot read edge file %s: %s", path, std::strerror(errno));
    const char *s = txt.c_str();
    char *end = nullptr;
    long nn = std::strtol(s, &end, 10);
    if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: missing node count", path);
    s = end;
    long long mm = std::strtoll(s, &end, 10);
    if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: missing row count", path);
    s = end;
    if (nn < 0 || mm < 0) PN_FAIL(PN_ERR_FORMAT, "%s: negative header", path);
    *n = (int32_t)nn;
    *m = (int64_t)mm;
    if (cap == 0) return PN_OK;
    if (cap < mm) PN_FAIL(PN_ERR_CAPACITY, "edge buffers hold %lld rows, file has %lld", (long long)cap, mm);
    if (!u || !v || !p) PN_FAIL(PN_ERR_ARG, "pn_edges_read_text: null output with cap > 0");
    for (long long i = 0; i < mm; i++) {
        long a = std::strtol(s, &end, 10);
        if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: row %lld truncated", path, i);
        s = end;
        long b = std::strtol(s, &end, 10);
        if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: row %lld truncated", path, i);
        s = end;
        double pr = std::strtod(s, &end);  // same conversion scanf("%lf") performs
        if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: row %lld truncated", path, i);
        s = end;
        u[i] = (int32_t)a;
        v[i] = (int32_t)b;
        p[i] = pr;
    }
    return PN_OK;
} catch (...) {
    return host_exception("pn_edges_read_text");
}

// ------------------------------------------------------------------------------------------------
// the uniform random-walk sampler's input and graph (gen.cpp:80-94, gen_epoch.cpp)
// ------------------------------------------------------------------------------------------------
int pn_pairs_read_text(const char *path, int32_t *n, int64_t *m, int32_t *u, int32_t *v, int64_t cap) try {
    if (!path || !n || !m) PN_FAIL(PN_ERR_ARG, "pn_pairs_read_text: null argument");
    if (read_number_file_fast<2>(path, n, m, u, v, nullptr, cap)) return PN_OK;
    std::string txt;
    if (!slurp(path, txt)) PN_FAIL(PN_ERR_IO, "cannot read pair file %s: %s", path, std::strerror(errno));
    const char *s = txt.c_str();
    char *end = nullptr;
    long nn = std::strtol(s, &end, 10);
    if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: missing node count", path);
    s = end;
    long long mm = std::strtoll(s, &end, 10);
    if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: missing pair count", path);
    s = end;
    if (nn < 0 || mm < 0) PN_FAIL(PN_ERR_FORMAT, "%s: negative header", path);
    *n = (int32_t)nn;
    *m = (int64_t)mm;
    if (cap == 0) return PN_OK;
    if (cap < mm) PN_FAIL(PN_ERR_CAPACITY, "pair buffers hold %lld rows, file has %lld", (long long)cap, mm);
    if (!u || !v) PN_FAIL(PN_ERR_ARG, "pn_pairs_read_text: null output with cap > 0");
    for (long long i = 0; i < mm; i++) {
        long a = std::strtol(s, &end, 10);
        if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: pair %lld truncated", path, i);
        s = end;
        long b = std::strtol(s, &end, 10);
        if (end == s) PN_FAIL(PN_ERR_FORMAT, "%s: pair %lld truncated", path, i);
        s = end;
        u[i] = (int32_t)a;
        v[i] = (int32_t)b;
    }
    return PN_OK;
} catch (...) {
    return host_exception("pn_pairs_read_text");
}

int pn_uniform_build(int32_t n, int64_t m, const int32_t *u, const int32_t *v, int64_t *off, int32_t *packed,
                     int32_t *src, int32_t *nbr, int64_t cap, int64_t *total) try {
    if (n < 0 || m < 0 || (m > 0 && (!u || !v)) || !total) PN_FAIL(PN_ERR_ARG, "pn_uniform_build: bad argument");
    std::vector<int64_t> at((size_t)n + 1, 0);
    for (int32_t i = 0; i < n; i++) at[i] = 1;                       // link(i, i), gen.cpp:83-84
    for (int64_t i = 0; i < m; i++) {
        if (u[i] < 0 || u[i] >= n || v[i] < 0 || v[i] >= n)
            PN_FAIL(PN_ERR_FORMAT, "pair %lld (%d, %d) outside [0, %d)", (long long)i, u[i], v[i], n);
        if (u[i] == v[i]) continue;                                  // :90-91
        at[u[i]]++;
        at[v[i]]++;
    }
    int64_t sum = 0;
    for (int32_t i = 0; i < n; i++) sum += at[i];
    *total = sum;
    if (cap == 0) return PN_OK;
    if (cap < sum) PN_FAIL(PN_ERR_CAPACITY, "neighbour buffers hold %lld entries, graph has %lld", (long long)cap,
                           (long long)sum);
    if (!off || !packed) PN_FAIL(PN_ERR_ARG, "pn_uniform_build: null output with cap > 0");
    int64_t run = 0;
    for (int32_t i = 0; i < n; i++) {
        off[i] = run;
        run += at[i];
        at[i] = off[i];
    }
    off[n] = run;
    auto put = [&](int32_t a, int32_t b) {
        const int64_t k = at[a]++;
        packed[4 * k] = b;
        packed[4 * k + 1] = b;
        packed[4 * k + 2] = 0;
        packed[4 * k + 3] = 0;
        if (src) src[k] = a;
        if (nbr) nbr[k] = b;
    };
    for (int32_t i = 0; i < n; i++) put(i, i);
    for (int64_t i = 0; i < m; i++) {
        if (u[i] == v[i]) continue;
        put(u[i], v[i]);                                             // link(u, v); link(v, u), :92-93
        put(v[i], u[i]);
    }
    return PN_OK;
} catch (...) {
    return host_exception("pn_uniform_build");
}

// ------------------------------------------------------------------------------------------------
// alias tables
// ------------------------------------------------------------------------------------------------
// smallest draw r in [0, 2^31) with (1.0 * r / RAND_MAX) > s, or 2^31 if there is none.  The quotient
// is monotone in r, so a bisection over the exact fp64 expression of roll() is exact.
static uint32_t draw_threshold(double s) {
    const double rmax = 2147483647.0;
    if (!(1.0 * 2147483647 / rmax > s)) return 2147483648u;  // also catches NaN
    uint32_t lo = 0, hi = 2147483647u;                        // pred(hi) is true
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (1.0 * (double)mid / rmax > s)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

namespace {
struct Pending {
    int32_t id;
    double mass;
};
}  // namespace

int pn_alias_build(int32_t n, int64_t m, const int32_t *u, const int32_t *v, const double *p, int64_t *off,
                   int32_t *A, int32_t *B, double *S, uint32_t *thr, int64_t cap, int64_t *total) try {
    if (n < 0 || m < 0 || !off || !total || (m > 0 && (!u || !v || !p)))
        PN_FAIL(PN_ERR_ARG, "pn_alias_build: bad argument");
    // bucket the rows per source node, keeping file order (link(), gen_merw.cpp:95-99)
    std::vector<int64_t> start((size_t)n + 1, 0);
    for (int64_t e = 0; e < m; e++) {
        if (u[e] < 0 || u[e] >= n) PN_FAIL(PN_ERR_ARG, "edge row %lld: source %d out of range", (long long)e, u[e]);
        // the targets become the A / B entries the walker uses as node ids: same bound
        if (v[e] < 0 || v[e] >= n) PN_FAIL(PN_ERR_ARG, "edge row %lld: target %d out of range", (long long)e, v[e]);
        start[(size_t)u[e] + 1]++;
    }
    for (int32_t i = 0; i < n; i++) start[(size_t)i + 1] += start[i];
    std::vector<int64_t> cursor(start.begin(), start.end() - 1);
    std::vector<int32_t> nbr((size_t)m);
    std::vector<double> prob((size_t)m);
    for (int64_t e = 0; e < m; e++) {
        int64_t at = cursor[u[e]]++;
        nbr[(size_t)at] = v[e];
        prob[(size_t)at] = p[e];
    }
    // AliasTable::init of one node (gen_merw.cpp:23-79); emit(a, b, s) receives its triples in order
    auto node_table = [&](int32_t node, std::deque<Pending> &heavy, std::deque<Pending> &light, auto &&emit) {
        const int64_t k = start[(size_t)node + 1] - start[node];
        heavy.clear();
        light.clear();
        for (int64_t j = 0; j < k; j++) {
            Pending e{nbr[(size_t)(start[node] + j)], prob[(size_t)(start[node] + j)] * (double)k};
            (e.mass > 1.0 ? heavy : light).push_back(e);
        }
        while (!heavy.empty() && !light.empty()) {
            Pending hv = heavy.front();
            heavy.pop_front();
            Pending lt = light.front();
            light.pop_front();
            emit(hv.id, lt.id, lt.mass);
            const double rest = hv.mass - (1.0 - lt.mass);
            if (std::fabs(rest - 1.0) < 1e-5) {  // eps, gen_merw.cpp:5 and :48
                emit(hv.id, hv.id, rest);
                continue;
            }
            (rest > 1.0 ? heavy : light).push_back(Pending{hv.id, rest});
        }
        for (; !heavy.empty(); heavy.pop_front()) emit(heavy.front().id, heavy.front().id, 1.0);
        for (; !light.empty(); light.pop_front()) emit(light.front().id, light.front().id, 1.0);
    };
    // the nodes' tables are independent: blocks of nodes per host thread, first to count the triples of every node
    // (their number depends on the masses), then -- offsets known -- to write them
    const int T = host_threads(m + n, 1 << 16);
    auto node_lo = [&](int ti) { return (int32_t)((int64_t)n * ti / T); };
    run_threads(T, [&](int ti) {
        std::deque<Pending> heavy, light;
        for (int32_t node = node_lo(ti); node < node_lo(ti + 1); node++) {
            int64_t c = 0;
            node_table(node, heavy, light, [&](int32_t, int32_t, double) { c++; });
            off[node] = c;
        }
    });
    int64_t count = 0;
    for (int32_t node = 0; node < n; node++) {
        const int64_t c = off[node];
        off[node] = count;
        count += c;
    }
    off[n] = count;
    *total = count;
    if (cap == 0) return PN_OK;
    if (cap < count)
        PN_FAIL(PN_ERR_CAPACITY, "alias buffers hold %lld triples, need %lld", (long long)cap, (long long)count);
    run_threads(T, [&](int ti) {
        std::deque<Pending> heavy, light;
        for (int32_t node = node_lo(ti); node < node_lo(ti + 1); node++) {
            int64_t at = off[node];
            node_table(node, heavy, light, [&](int32_t a, int32_t b, double s) {
                if (A) A[at] = a;
                if (B) B[at] = b;
                if (S) S[at] = s;
                if (thr) thr[at] = draw_threshold(s);
                at++;
            });
        }
    });
    return PN_OK;
} catch (...) {
    return host_exception("pn_alias_build");
}

int pn_node_ref_pack(int32_t n, const int64_t *off, uint32_t *ref) {
    if (n < 0 || !off || (n > 0 && !ref)) PN_FAIL(PN_ERR_ARG, "pn_node_ref_pack: bad argument");
    if (off[n] >= (1LL << 32)) PN_FAIL(PN_ERR_ARG, "pn_node_ref_pack: %lld triples do not fit 32-bit positions", (long long)off[n]);
    for (int32_t i = 0; i < n; i++) {
        if (off[i + 1] < off[i]) PN_FAIL(PN_ERR_ARG, "pn_node_ref_pack: off[] is not a prefix sum at node %d", i);
        ref[2 * (size_t)i] = (uint32_t)off[i];
        ref[2 * (size_t)i + 1] = (uint32_t)(off[i + 1] - off[i]);
    }
    return PN_OK;
}

int pn_alias_pack(int64_t total, const int32_t *A, const int32_t *B, const uint32_t *thr, int32_t *dst) try {
    if (total < 0 || (total > 0 && (!A || !B || !thr || !dst))) PN_FAIL(PN_ERR_ARG, "pn_alias_pack: bad argument");
    for (int64_t i = 0; i < total; i++) {
        dst[4 * i + 0] = A[i];
        dst[4 * i + 1] = B[i];
        dst[4 * i + 2] = (int32_t)thr[i];
        dst[4 * i + 3] = 0;
    }
    return PN_OK;
} catch (...) {
    return host_exception("pn_alias_pack");
}

int pn_csr_build(int32_t n, int64_t m, const int32_t *u, const int32_t *v, int32_t reverse, int64_t *off,
                 int32_t *adj, int64_t cap, int64_t *count) try {
    if (n < 0 || m < 0 || !off || !count || (m > 0 && (!u || !v))) PN_FAIL(PN_ERR_ARG, "pn_csr_build: bad argument");
    const int32_t *src = reverse ? v : u, *dst = reverse ? u : v;
    std::vector<int64_t> start((size_t)n + 1, 0);
    for (int64_t e = 0; e < m; e++) {
        if (u[e] < 0 || u[e] >= n || v[e] < 0 || v[e] >= n)
            PN_FAIL(PN_ERR_ARG, "edge row %lld out of range", (long long)e);
        start[(size_t)src[e] + 1]++;
    }
    for (int32_t i = 0; i < n; i++) start[(size_t)i + 1] += start[i];
    std::vector<int64_t> cursor(start.begin(), start.end() - 1);
    std::vector<int32_t> raw((size_t)m);
    for (int64_t e = 0; e < m; e++) raw[(size_t)cursor[src[e]]++] = dst[e];
    // per-node sort + unique in place (blocks of nodes per host thread), then the compaction to the output
    const int T = host_threads(m + n, 1 << 16);
    auto node_lo = [&](int ti) { return (int32_t)((int64_t)n * ti / T); };
    run_threads(T, [&](int ti) {
        for (int32_t i = node_lo(ti); i < node_lo(ti + 1); i++) {
            int32_t *b = raw.data() + start[i], *e = raw.data() + start[(size_t)i + 1];
            std::sort(b, e);
            off[i] = std::unique(b, e) - b;
        }
    });
    int64_t total = 0;
    for (int32_t i = 0; i < n; i++) {
        const int64_t c = off[i];
        off[i] = total;
        total += c;
    }
    off[n] = total;
    *count = total;
    if (cap != 0 && cap < total)
        PN_FAIL(PN_ERR_CAPACITY, "adjacency buffer holds %lld entries, need %lld", (long long)cap, (long long)total);
    if (cap != 0 && adj)
        run_threads(T, [&](int ti) {
            for (int32_t i = node_lo(ti); i < node_lo(ti + 1); i++)
                std::copy(raw.data() + start[i], raw.data() + start[i] + (off[(size_t)i + 1] - off[i]), adj + off[i]);
        });
    return PN_OK;
} catch (...) {
    return host_exception("pn_csr_build");
}

int pn_glibc_draws(uint32_t seed, uint64_t first, int64_t count, int32_t *out) try {
    if (count < 0 || (count > 0 && !out)) PN_FAIL(PN_ERR_ARG, "pn_glibc_draws: bad argument");
    GlibcState st = glibc_apply(glibc_poly_xpow(first), glibc_seed_state(seed));
    uint32_t ring[31];
    std::memcpy(ring, st.s, sizeof ring);
    // ring[j] holds word (pos + j); the word after the window is w[0] + w[28]
    for (int64_t k = 0; k < count; k++) {
        const int at = (int)(k % 31);
        out[k] = (int32_t)(ring[at] >> 1);
        ring[at] = ring[at] + ring[(at + 28) % 31];
    }
    return PN_OK;
} catch (...) {
    return host_exception("pn_glibc_draws");
}

// ------------------------------------------------------------------------------------------------
// hop table
// ------------------------------------------------------------------------------------------------
int pn_hops_dense(int32_t n, int64_t m, const int32_t *u, const int32_t *v, int32_t seq_len, uint8_t *dis) try {
    if (n < 0 || m < 0 || seq_len < 1 || seq_len > 254 || !dis || (m > 0 && (!u || !v)))
        PN_FAIL(PN_ERR_ARG, "pn_hops_dense: bad argument");
    std::vector<int64_t> start((size_t)n + 1, 0);
    for (int64_t e = 0; e < m; e++) {
        if (u[e] < 0 || u[e] >= n || v[e] < 0 || v[e] >= n)
            PN_FAIL(PN_ERR_ARG, "edge row %lld out of range", (long long)e);
        start[(size_t)u[e] + 1]++;
    }
    for (int32_t i = 0; i < n; i++) start[(size_t)i + 1] += start[i];
    std::vector<int64_t> cursor(start.begin(), start.end() - 1);
    std::vector<int32_t> nbr((size_t)m);
    for (int64_t e = 0; e < m; e++) nbr[(size_t)cursor[u[e]]++] = v[e];
    // every source's BFS is independent and owns one row of dis: contiguous blocks of sources per host thread
    const int T = host_threads(n, 64);
    run_threads(T, [&](int ti) {
        const int32_t lo = (int32_t)((int64_t)n * ti / T), hi = (int32_t)((int64_t)n * (ti + 1) / T);
        std::vector<int32_t> frontier, next;
        for (int32_t src = lo; src < hi; src++) {
            uint8_t *row = dis + (size_t)src * (size_t)n;
            std::memset(row, 0, (size_t)n);
            row[src] = 1;
            frontier.assign(1, src);
            // a walk of seq_len nodes reaches at most seq_len-1 hops: label levels 1..seq_len
            for (int32_t level = 1; level < seq_len && !frontier.empty(); level++) {
                next.clear();
                for (int32_t x : frontier)
                    for (int64_t j = start[x]; j < start[(size_t)x + 1]; j++) {
                        int32_t y = nbr[(size_t)j];
                        if (row[y] == 0) {
                            row[y] = (uint8_t)(level + 1);
                            next.push_back(y);
                        }
                    }
                frontier.swap(next);
            }
        }
    });
    return PN_OK;
} catch (...) {
    return host_exception("pn_hops_dense");
}

// ------------------------------------------------------------------------------------------------
// path file
// ------------------------------------------------------------------------------------------------
}  // extern "C"

static inline char *put_uint(char *w, uint32_t x) {
    char tmp[12];
    int k = 0;
    do {
        tmp[k++] = (char)('0' + x % 10);
        x /= 10;
    } while (x);
    while (k) *w++ = tmp[--k];
    return w;
}

// "[v0, ..., v_{L-1}, d0, ..., d_{L-1}]\n" for paths [p0, p1) -> w; returns the end of the text
static char *format_paths(char *w, const int32_t *ids, const uint8_t *codes, int64_t p0, int64_t p1, int32_t L) {
    for (int64_t q = p0; q < p1; q++) {
        *w++ = '[';
        for (int32_t t = 0; t < L; t++) {
            int32_t id = ids[q * L + t];
            if (id < 0) {
                *w++ = '-';
                w = put_uint(w, (uint32_t)(-(int64_t)id));
            } else
                w = put_uint(w, (uint32_t)id);
            *w++ = ',';
            *w++ = ' ';
        }
        for (int32_t t = 0; t < L; t++) {
            w = put_uint(w, codes[q * L + t]);
            if (t + 1 < L) {
                *w++ = ',';
                *w++ = ' ';
            }
        }
        *w++ = ']';
        *w++ = '\n';
    }
    return w;
}

static inline int digits10(uint32_t x) {
    return x < 10 ? 1 : x < 100 ? 2 : x < 1000 ? 3 : x < 10000 ? 4 : x < 100000 ? 5 : x < 1000000 ? 6
         : x < 10000000 ? 7 : x < 100000000 ? 8 : x < 1000000000 ? 9 : 10;
}

// exact size of format_paths' output for paths [p0, p1)
static int64_t text_bytes(const int32_t *ids, const uint8_t *codes, int64_t p0, int64_t p1, int32_t L) {
    int64_t n = (p1 - p0) * (int64_t)(1 + 2 * L + 2 * (L - 1) + 2);     // "[", ", " per id, ", " between codes, "]\n"
    for (int64_t k = p0 * L; k < p1 * L; k++) {
        const int32_t id = ids[k];
        n += id < 0 ? 1 + digits10((uint32_t)(-(int64_t)id)) : digits10((uint32_t)id);
        n += digits10(codes[k]);
    }
    return n;
}

static bool pwrite_all(int fd, const char *buf, size_t n, int64_t at) {
    while (n) {
        ssize_t k = ::pwrite(fd, buf, n, (off_t)at);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        buf += k;
        at += k;
        n -= (size_t)k;
    }
    return true;
}

extern "C" {

// Every host thread owns one contiguous range of the paths: a first pass sizes the text of each range (digit
// counting only), the prefix sum gives each range its file offset, the second pass formats the range block by block
// and writes the blocks in place -- no serial section, two thread spawns per call.
int pn_paths_write_text(const char *path, const int32_t *ids, const uint8_t *codes, int64_t npaths, int32_t L,
                        int32_t append) try {
    if (!path || npaths < 0 || L < 1 || (npaths > 0 && (!ids || !codes)))
        PN_FAIL(PN_ERR_ARG, "pn_paths_write_text: bad argument");
    const int fd = ::open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0666);
    if (fd < 0) PN_FAIL(PN_ERR_IO, "cannot open %s for writing: %s", path, std::strerror(errno));
    struct FdGuard {        // closes on every early way out (errors, exceptions)
        int fd;
        ~FdGuard() {
            if (fd >= 0) ::close(fd);
        }
    } guard{fd};
    int64_t at = 0;
    if (append) {
        at = (int64_t)::lseek(fd, 0, SEEK_END);
        if (at < 0) PN_FAIL(PN_ERR_IO, "cannot seek in %s: %s", path, std::strerror(errno));
    }
    const int T = host_threads(npaths, 1 << 14);
    std::vector<int64_t> lo((size_t)T + 1), off((size_t)T + 1, 0);
    for (int i = 0; i <= T; i++) lo[i] = npaths / T * i + std::min<int64_t>(i, npaths % T);
    if (T > 1) run_threads(T, [&](int i) { off[i + 1] = text_bytes(ids, codes, lo[i], lo[i + 1], L); });
    off[0] = at;
    for (int i = 0; i < T; i++) off[i + 1] += off[i];
    std::atomic<int> failed_errno(0);
    run_threads(T, [&](int i) {
        // per path node at most "-2147483648, " (13) for the id and "255, " (5) for the code, plus "[" and "]\n"
        const size_t line_max = (size_t)L * (13 + 5) + 4;
        const int64_t block = 1 << 14;
        std::vector<char> buf(line_max * (size_t)std::min<int64_t>(block, std::max<int64_t>(lo[i + 1] - lo[i], 1)));
        int64_t cursor = off[i];
        for (int64_t p0 = lo[i]; p0 < lo[i + 1] && !failed_errno; p0 += block) {
            const size_t len = (size_t)(format_paths(buf.data(), ids, codes, p0, std::min(lo[i + 1], p0 + block), L) -
                                        buf.data());
            if (!pwrite_all(fd, buf.data(), len, cursor)) failed_errno = errno ? errno : EIO;
            cursor += (int64_t)len;
        }
    });
    guard.fd = -1;
    if (::close(fd) != 0 && !failed_errno) failed_errno = errno ? errno : EIO;
    if (failed_errno) PN_FAIL(PN_ERR_IO, "short write to %s: %s", path, std::strerror(failed_errno));
    return PN_OK;
} catch (...) {
    return host_exception("pn_paths_write_text");
}

}  // extern "C"

namespace {
enum ParseErr { PE_NONE = 0, PE_START, PE_FIELD, PE_FEWER, PE_END };
struct ParseResult {
    int64_t lines = 0;     // complete lines parsed before the range ended or failed
    ParseErr err = PE_NONE;
    int field = 0;
};

// Parses whole lines from [s, e); line `line0 + k` of the range goes to slot line0 + k while that is below cap.
ParseResult parse_lines(const char *s, const char *e, int32_t L, int32_t *ids, uint8_t *codes, int64_t cap,
                        int64_t line0) {
    ParseResult r;
    while (s < e) {
        // the reader slices line[1:-2] (PathNet_run.py:327): first char '[' and the last two "]\n"
        if (*s != '[') {
            r.err = PE_START;
            return r;
        }
        s++;
        const int64_t slot = line0 + r.lines;
        const bool keep = cap > 0 && slot < cap;
        for (int32_t k = 0; k < 2 * L; k++) {
            while (s < e && *s == ' ') s++;
            bool neg = false;
            if (s < e && *s == '-') {
                neg = true;
                s++;
            }
            if (s >= e || *s < '0' || *s > '9') {
                r.err = PE_FIELD;
                r.field = k;
                return r;
            }
            int64_t val = 0;
            while (s < e && *s >= '0' && *s <= '9') val = val * 10 + (*s++ - '0');
            if (neg) val = -val;
            if (keep) {
                if (k < L)
                    ids[slot * L + k] = (int32_t)val;
                else
                    codes[slot * L + (k - L)] = (uint8_t)val;
            }
            if (k + 1 < 2 * L) {
                if (s >= e || *s != ',') {
                    r.err = PE_FEWER;
                    return r;
                }
                s++;
            }
        }
        if (s >= e || *s != ']' || s + 1 >= e || s[1] != '\n') {
            r.err = PE_END;
            return r;
        }
        s += 2;
        r.lines++;
    }
    return r;
}

int64_t count_newlines(const char *s, const char *e) {
    int64_t n = 0;
    while (s < e) {
        const char *q = static_cast<const char *>(std::memchr(s, '\n', (size_t)(e - s)));
        if (!q) break;
        n++;
        s = q + 1;
    }
    return n;
}
}  // namespace

extern "C" {

// The file is cut into one byte range per host thread at line ends; a first pass counts the lines of each range (so
// that every range knows its first slot), the second parses the ranges concurrently.  A malformed line is reported
// exactly as a sequential scan would: the first one in file order, by its line number.
int pn_paths_read_text(const char *path, int32_t L, int32_t *ids, uint8_t *codes, int64_t cap, int64_t *npaths) try {
    if (!path || L < 1 || !npaths) PN_FAIL(PN_ERR_ARG, "pn_paths_read_text: bad argument");
    if (cap > 0 && (!ids || !codes)) PN_FAIL(PN_ERR_ARG, "pn_paths_read_text: cap > 0 needs ids and codes");
    MappedFile txt;
    if (!txt.open(path)) PN_FAIL(PN_ERR_IO, "cannot read path file %s: %s", path, std::strerror(errno));
    const char *s = txt.data, *e = s + txt.size;
    const int T = host_threads((int64_t)txt.size, 1 << 20);
    std::vector<const char *> cut((size_t)T + 1, e);
    cut[0] = s;
    for (int i = 1; i < T; i++) {
        const char *c = std::max(cut[i - 1], s + txt.size / T * i);
        const char *q = c < e ? static_cast<const char *>(std::memchr(c, '\n', (size_t)(e - c))) : nullptr;
        cut[i] = q ? q + 1 : e;
    }
    std::vector<int64_t> first((size_t)T + 1, 0);
    run_threads(T, [&](int i) { first[i + 1] = count_newlines(cut[i], cut[i + 1]); });
    for (int i = 0; i < T; i++) first[i + 1] += first[i];
    if (cap == 0 && (txt.size == 0 || e[-1] == '\n')) {
        // sizing call on a file that ends in a line end: its lines are validated by the filling call
        *npaths = first[T];
        return PN_OK;
    }
    std::vector<ParseResult> res((size_t)T);
    run_threads(T, [&](int i) { res[i] = parse_lines(cut[i], cut[i + 1], L, ids, codes, cap, first[i]); });
    int64_t count = 0;
    for (int i = 0; i < T; i++) {
        // (a range without error holds exactly the lines its newline count promised, so `line` is the file's line number)
        const long long line = (long long)(first[i] + res[i].lines);
        switch (res[i].err) {
            case PE_NONE: break;
            case PE_START: PN_FAIL(PN_ERR_FORMAT, "%s: line %lld does not start with '['", path, line);
            case PE_FIELD: PN_FAIL(PN_ERR_FORMAT, "%s: line %lld field %d is not an integer", path, line, res[i].field);
            case PE_FEWER: PN_FAIL(PN_ERR_FORMAT, "%s: line %lld has fewer than %d fields", path, line, 2 * L);
            case PE_END:
                PN_FAIL(PN_ERR_FORMAT, "%s: line %lld does not end with \"]\\n\" after %d fields", path, line, 2 * L);
        }
        count += res[i].lines;
    }
    *npaths = count;
    if (cap > 0 && count > cap)
        PN_FAIL(PN_ERR_CAPACITY, "path buffers hold %lld paths, file has %lld", (long long)cap, (long long)count);
    return PN_OK;
} catch (...) {
    return host_exception("pn_paths_read_text");
}

namespace {
struct BinHeader {
    char magic[8];
    int32_t L;
    int32_t reserved0;
    int64_t npaths;
    int64_t reserved1;
};
static_assert(sizeof(BinHeader) == 32, "header is 32 bytes");
const char kBinMagic[8] = {'P', 'N', 'P', 'A', 'T', 'H', 'S', '1'};
}  // namespace

int pn_paths_write_bin(const char *path, const int32_t *ids, const uint8_t *codes, int64_t npaths, int32_t L) try {
    if (!path || npaths < 0 || L < 1 || (npaths > 0 && (!ids || !codes)))
        PN_FAIL(PN_ERR_ARG, "pn_paths_write_bin: bad argument");
    FILE *f = std::fopen(path, "wb");
    if (!f) PN_FAIL(PN_ERR_IO, "cannot open %s for writing: %s", path, std::strerror(errno));
    BinHeader h{};
    std::memcpy(h.magic, kBinMagic, 8);
    h.L = L;
    h.npaths = npaths;
    const size_t n = (size_t)npaths * (size_t)L;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    ok = ok && (n == 0 || std::fwrite(ids, sizeof(int32_t), n, f) == n);
    ok = ok && (n == 0 || std::fwrite(codes, 1, n, f) == n);
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) PN_FAIL(PN_ERR_IO, "short write to %s", path);
    return PN_OK;
} catch (...) {
    return host_exception("pn_paths_write_bin");
}

int pn_paths_read_bin(const char *path, int32_t *L_out, int32_t *ids, uint8_t *codes, int64_t cap, int64_t *npaths) try {
    if (!path || !npaths || !L_out) PN_FAIL(PN_ERR_ARG, "pn_paths_read_bin: bad argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) PN_FAIL(PN_ERR_IO, "cannot read path file %s: %s", path, std::strerror(errno));
    BinHeader h{};
    if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, kBinMagic, 8) != 0 || h.L < 1 || h.npaths < 0) {
        std::fclose(f);
        PN_FAIL(PN_ERR_FORMAT, "%s is not a PNPATHS1 file", path);
    }
    *npaths = h.npaths;
    *L_out = h.L;
    if (cap == 0) {
        std::fclose(f);
        return PN_OK;
    }
    if (cap < h.npaths || !ids || !codes) {
        std::fclose(f);
        PN_FAIL(PN_ERR_CAPACITY, "path buffers hold %lld paths, file has %lld", (long long)cap, (long long)h.npaths);
    }
    const size_t n = (size_t)h.npaths * (size_t)h.L;
    const bool ok = (n == 0) || (std::fread(ids, sizeof(int32_t), n, f) == n && std::fread(codes, 1, n, f) == n);
    std::fclose(f);
    if (!ok) PN_FAIL(PN_ERR_FORMAT, "%s is truncated", path);
    return PN_OK;
} catch (...) {
    return host_exception("pn_paths_read_bin");
}

}  // extern "C"
