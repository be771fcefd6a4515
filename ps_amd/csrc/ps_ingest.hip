// ps_ingest.hip -- the step in front of the hot path: libsvm text ->
// the arrays Model.train consumes, resident in HBM.
//
//   data/LibsvmParser.java:13-25   line.split(" "): cols[0] = label (Float.parseFloat),
//                                  cols[i] = "<Long.parseLong idx>:<Float.parseFloat value>"
//   CTR.java:47-68 parseFeature    Y = cols[0]; E[f] = idx of col 1+f (f < F, the long goes
//                                  through a float); X[j] = value of col 1+F+j (j < X);
//                                  W = MatrixUtil.hash(E, wideSize) = fmodf(E, wideSize)
//   data/DataSource.java:25-46     worker sharding: this reader takes lines offset, offset+step, ...
//   data/DataSet.java              reader threads fill a queue of parsed batches
//
// Here: the text is mapped once, the line index of this worker's lines is built once, and a
// pool of host threads parses one batch at a time straight into PINNED buffers (int64 ids,
// f32 dense / labels, int64 wide ids) while the previous batch trains; the H2D copies run on
// their own stream into one of two device slots (double buffering), and ps_ingest_next hands
// out a ps_batch_t whose pointers are device pointers.  ps_libsvm_parse is the same parser as
// a plain host function (no GPU) -- what the CPU tests pin against the restated reference.
#include <fcntl.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ps_store.h"

namespace {

// ---- numbers ------------------------------------------------------------------
const float kPow10f[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};

// Float.parseFloat: the decimal string rounded once to the nearest float.  Fast path: a
// mantissa below 2^24 and a power of ten up to 10^10 are both exact floats, so ONE IEEE
// multiply/divide is the correctly rounded result.  Everything else (long mantissas, large
// exponents, nan/inf, hex) goes to strtof, which glibc rounds correctly as well.
bool parse_float_tok(const char *p, const char *e, float *out) {
    const char *q = p;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    uint64_t m = 0;
    int exp10 = 0;
    bool any = false, slow = false;
    while (q < e && *q >= '0' && *q <= '9') {
        if (m < (1ull << 24)) m = m * 10 + (uint64_t)(*q - '0');
        else slow = true;
        any = true; ++q;
    }
    if (q < e && *q == '.') {
        ++q;
        while (q < e && *q >= '0' && *q <= '9') {
            if (m < (1ull << 24)) { m = m * 10 + (uint64_t)(*q - '0'); --exp10; }
            else slow = true;
            any = true; ++q;
        }
    }
    if (!any) slow = true;
    if (!slow && q < e && (*q == 'e' || *q == 'E')) {
        const char *r = q + 1;
        bool eneg = false;
        if (r < e && (*r == '-' || *r == '+')) { eneg = *r == '-'; ++r; }
        int ev = 0;
        bool edig = false;
        while (r < e && *r >= '0' && *r <= '9' && ev < 1000) { ev = ev * 10 + (*r - '0'); ++r; edig = true; }
        if (!edig) slow = true;
        exp10 += eneg ? -ev : ev;
        q = r;
    }
    if (!slow && q == e && m < (1ull << 24) && exp10 >= -10 && exp10 <= 10) {
        float v = (float)m;
        if (exp10 < 0) v = v / kPow10f[-exp10];
        else if (exp10 > 0) v = v * kPow10f[exp10];
        *out = neg ? -v : v;
        return true;
    }
    char buf[64];
    const size_t n = (size_t)(e - p);
    if (n == 0 || n >= sizeof buf) return false;
    memcpy(buf, p, n);
    buf[n] = 0;
    if (buf[n - 1] == 'f' || buf[n - 1] == 'F' || buf[n - 1] == 'd' || buf[n - 1] == 'D') buf[n - 1] = 0;   // Java accepts a type suffix
    char *end = nullptr;
    const float v = strtof(buf, &end);
    if (end == buf || *end != 0) return false;
    *out = v;
    return true;
}

bool parse_long_tok(const char *p, const char *e, int64_t *out) {
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    if (p == e) return false;
    uint64_t v = 0;
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        if (v > (uint64_t)INT64_MAX / 10) return false;
        v = v * 10 + (uint64_t)(*p - '0');
        if (v > (uint64_t)INT64_MAX) return false;
    }
    *out = neg ? -(int64_t)v : (int64_t)v;
    return true;
}

struct LineOut { int64_t *ids; float *dense; float *label; int64_t *wide; };

// one line -> one sample.  0 ok, else the 1-based column that failed (1 = label), -1 = too few columns
int parse_line(const char *p, const char *e, const ps_ingest_config_t &c, const LineOut &o) {
    while (e > p && (e[-1] == '\r' || e[-1] == ' ' || e[-1] == '\t')) --e;
    int col = 0;
    const int need = 1 + c.F + c.X;
    while (p < e && col < need) {
        while (p < e && *p == ' ') ++p;
        const char *t = p;
        while (p < e && *p != ' ') ++p;
        if (t == p) break;
        if (col == 0) {
            if (!parse_float_tok(t, p, o.label)) return 1;
        } else {
            const char *colon = (const char *)memchr(t, ':', (size_t)(p - t));
            if (!colon) return col + 1;
            if (col <= c.F) {
                int64_t idx;
                if (!parse_long_tok(t, colon, &idx)) return col + 1;
                float junk;
                if (!parse_float_tok(colon + 1, p, &junk)) return col + 1;     // LibsvmParser parses (and boxes) it anyway
                int64_t id = idx, w = 0;
                if (c.ids_via_float) {
                    const float fe = (float)idx;                               // E[j-1][i] = cols.get(j).getIdx()  (long -> float)
                    id = (int64_t)fe;
                    if (c.wide_size > 0) w = (int64_t)fmodf(fe, (float)c.wide_size);   // MatrixUtil.hash (util/MatrixUtil.java:27-33)
                } else if (c.wide_size > 0) {
                    w = idx % c.wide_size;
                }
                o.ids[col - 1] = id;
                if (o.wide) o.wide[col - 1] = w;
            } else {
                int64_t idx;
                if (!parse_long_tok(t, colon, &idx)) return col + 1;
                if (!parse_float_tok(colon + 1, p, &o.dense[col - 1 - c.F])) return col + 1;
            }
        }
        ++col;
    }
    return col == need ? 0 : -1;
}

// ---- a small pool: run fn(i) for i in [0,n) on the pool's threads + the caller ----
class Pool {
  public:
    explicit Pool(int nthreads) {
        for (int i = 1; i < nthreads; ++i) th_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)> &fn) {
        if (n <= 0) return;
        {
            std::lock_guard<std::mutex> l(mu_);
            fn_ = &fn; n_ = n; grain_ = grain < 1 ? 1 : grain; next_.store(0); pending_ = (int)th_.size(); ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void work() {
        for (;;) {
            const int64_t b = next_.fetch_add(grain_);
            if (b >= n_) return;
            (*fn_)(b, b + grain_ < n_ ? b + grain_ : n_);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
            {
                std::lock_guard<std::mutex> l(mu_);
                --pending_;
            }
            done_.notify_one();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int64_t, int64_t)> *fn_ = nullptr;
    int64_t n_ = 0, grain_ = 1;
    std::atomic<int64_t> next_{0};
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// start offsets of this worker's lines.  DataSource.readLine (data/DataSource.java:25-46) counts RAW lines, blank ones
// included: raw line g is this reader's iff g >= offset and (g - offset) % step == 0; a selected line that is blank
// then yields an empty feature list (LibsvmParser.parse, StringUtils.isBlank) and is dropped.
void index_lines(const char *d, size_t len, int offset, int step, std::vector<size_t> *starts, std::vector<size_t> *ends) {
    size_t pos = 0;
    int64_t g = 0;
    while (pos < len) {
        const char *nl = (const char *)memchr(d + pos, '\n', len - pos);
        const size_t end = nl ? (size_t)(nl - d) : len;
        if (g >= offset && (g - offset) % step == 0) {
            bool blank = true;
            for (size_t i = pos; i < end; ++i)
                if (d[i] != ' ' && d[i] != '\t' && d[i] != '\r') { blank = false; break; }
            if (!blank) { starts->push_back(pos); ends->push_back(end); }
        }
        ++g;
        pos = end + 1;
    }
}

int check_cfg(const ps_ingest_config_t *c) {
    if (!c || c->F < 0 || c->X < 0 || c->F + c->X <= 0 || c->batch <= 0 || c->offset < 0 || c->step < 1 || c->wide_size < 0)
        return ps_set_err(PS_E_BAD_ARG, "bad ingest config");
    return PS_OK;
}

int parse_range(const char *d, const std::vector<size_t> &st, const std::vector<size_t> &en, int64_t first, int64_t n,
                const ps_ingest_config_t &c, int64_t *ids, float *dense, float *labels, int64_t *wide, Pool *pool) {
    std::atomic<int64_t> bad_line{-1};
    std::atomic<int> bad_col{0};
    auto body = [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            LineOut o{ids + i * c.F, dense + i * c.X, labels + i, (wide && c.wide_size > 0) ? wide + i * c.F : nullptr};
            const int rc = parse_line(d + st[first + i], d + en[first + i], c, o);
            if (rc != 0) {
                int64_t exp = -1;
                if (bad_line.compare_exchange_strong(exp, first + i)) bad_col.store(rc);
            }
        }
    };
    if (pool) pool->run(n, 64, body);
    else body(0, n);
    if (bad_line.load() >= 0) {
        if (bad_col.load() < 0)
            return ps_set_err(PS_E_BAD_ARG, "libsvm line %lld of this reader: fewer than %d columns", (long long)bad_line.load(), 1 + c.F + c.X);
        return ps_set_err(PS_E_BAD_ARG, "libsvm line %lld of this reader: column %d does not parse", (long long)bad_line.load(), bad_col.load());
    }
    return PS_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// host-only parser
// ---------------------------------------------------------------------------
extern "C" int ps_libsvm_count(const char *text, size_t len, int offset, int step, int64_t *n_lines) {
    if (!text || !n_lines || offset < 0 || step < 1) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    std::vector<size_t> st, en;
    index_lines(text, len, offset, step, &st, &en);
    *n_lines = (int64_t)st.size();
    return PS_OK;
}

extern "C" int ps_libsvm_parse(const char *text, size_t len, const ps_ingest_config_t *cfg, int64_t first_line, int64_t max_lines,
                               int64_t *ids, float *dense, float *labels, int64_t *wide_ids, int64_t *n_parsed) {
    PSCHK(check_cfg(cfg));
    if (!text || !n_parsed || first_line < 0 || max_lines < 0 || (cfg->F > 0 && !ids) || (cfg->X > 0 && !dense) || !labels)
        return ps_set_err(PS_E_BAD_ARG, "null argument");
    std::vector<size_t> st, en;
    index_lines(text, len, cfg->offset, cfg->step, &st, &en);
    int64_t n = (int64_t)st.size() - first_line;
    if (n < 0) n = 0;
    if (n > max_lines) n = max_lines;
    *n_parsed = n;
    if (n == 0) return PS_OK;
    const int nt = cfg->threads > 1 ? cfg->threads : 1;
    if (nt > 1) {
        Pool pool(nt);
        return parse_range(text, st, en, first_line, n, *cfg, ids, dense, labels, wide_ids, &pool);
    }
    return parse_range(text, st, en, first_line, n, *cfg, ids, dense, labels, wide_ids, nullptr);
}

// ---------------------------------------------------------------------------
// the pipeline: parse batch k+1 into pinned memory and copy it to HBM while batch k trains
// ---------------------------------------------------------------------------
struct ps_ingest {
    ps_store *s = nullptr;
    ps_ingest_config_t cfg{};
    const char *data = nullptr;
    size_t len = 0;
    void *map = nullptr; size_t map_len = 0;          // mmap of a file
    std::vector<char> copy;                           // or a private copy of the caller's memory
    std::vector<size_t> st, en;
    int64_t next_line = 0;
    Pool *pool = nullptr;
    hipStream_t copy_stream = nullptr;
    struct Slot {
        int64_t *ids_h = nullptr, *wide_h = nullptr; float *dense_h = nullptr, *labels_h = nullptr;     // pinned
        int64_t *ids_d = nullptr, *wide_d = nullptr; float *dense_d = nullptr, *labels_d = nullptr;     // HBM
        hipEvent_t copied = nullptr;                  // the slot's H2D copies are done (copy stream)
        hipEvent_t consumed = nullptr;                // the kernels that read the slot were enqueued before this (store stream)
        bool consumed_recorded = false;
        int B = 0;
        int rc = PS_OK;
        char err[256] = "";
        bool pending = false;                         // a fill was requested and has not finished
    } slot[2];
    // ONE persistent filler thread (parse + H2D of one slot at a time).  Not a thread per fill: HIP calls from
    // short-lived threads raced with the training thread's launches (measured: occasional wrong batches).
    std::thread filler;
    std::mutex mu;
    std::condition_variable cv_req, cv_done;
    int req = -1;
    bool stop = false;
    int cur = 0;                                      // slot the next ps_ingest_next hands out
    bool primed = false;
    double parse_s = 0; int64_t parsed_lines = 0, parsed_bytes = 0;
};

namespace {

void ingest_copy(ps_ingest *g, int k);

void ingest_fill(ps_ingest *g, int k) {               // runs on the slot's worker thread
    ps_ingest::Slot &S = g->slot[k];
    const ps_ingest_config_t &c = g->cfg;
    S.rc = PS_OK;
    int64_t n = (int64_t)g->st.size() - g->next_line;
    if (n > c.batch) n = c.batch;
    if (n < 0) n = 0;
    const int64_t first = g->next_line;
    g->next_line += n;
    S.B = (int)n;
    if (n == 0) return;
    (void)hipSetDevice(g->s->device);
    const auto t0 = std::chrono::steady_clock::now();
    S.rc = parse_range(g->data, g->st, g->en, first, n, c, S.ids_h, S.dense_h, S.labels_h, S.wide_h, g->pool);
    g->parse_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g->parsed_lines += n;
    g->parsed_bytes += (int64_t)(g->en[first + n - 1] - g->st[first]);
    if (S.rc != PS_OK) { snprintf(S.err, sizeof S.err, "%s", ps_last_error()); return; }
    ingest_copy(g, k);
}

void ingest_copy(ps_ingest *g, int k) {
    ps_ingest::Slot &S = g->slot[k];
    const ps_ingest_config_t &c = g->cfg;
    const int64_t n = S.B;
    if (n == 0 || S.rc != PS_OK) return;
    hipStream_t st = g->copy_stream;
    hipError_t e = hipSuccess;
    // Everything is host-synchronised HERE, in the background: wait until the kernels that read this slot's
    // previous batch are done, copy, and wait for the copies to land.  The training thread then needs no
    // cross-stream event at all (cross-thread event waits proved unreliable: occasional stale batches).
    if (S.consumed_recorded) e = hipEventSynchronize(S.consumed);
    if (e == hipSuccess && c.F > 0) e = hipMemcpyAsync(S.ids_d, S.ids_h, sizeof(int64_t) * n * c.F, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && c.F > 0 && c.wide_size > 0) e = hipMemcpyAsync(S.wide_d, S.wide_h, sizeof(int64_t) * n * c.F, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && c.X > 0) e = hipMemcpyAsync(S.dense_d, S.dense_h, sizeof(float) * n * c.X, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(S.labels_d, S.labels_h, sizeof(float) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { S.rc = PS_E_HIP; snprintf(S.err, sizeof S.err, "ingest H2D: %s", hipGetErrorString(e)); }
}

void ingest_wait(ps_ingest *g, int k) {
    std::unique_lock<std::mutex> l(g->mu);
    g->cv_done.wait(l, [&] { return !g->slot[k].pending; });
}

void filler_loop(ps_ingest *g) {
    for (;;) {
        int k;
        {
            std::unique_lock<std::mutex> l(g->mu);
            g->cv_req.wait(l, [&] { return g->stop || g->req >= 0; });
            if (g->stop) return;
            k = g->req; g->req = -1;
        }
        ingest_fill(g, k);
        {
            std::lock_guard<std::mutex> l(g->mu);
            g->slot[k].pending = false;
        }
        g->cv_done.notify_all();
    }
}

void ingest_start(ps_ingest *g, int k) {
    ingest_wait(g, k);
    ingest_wait(g, k ^ 1);                            // one fill at a time (they share next_line and the pool)
    if (!g->filler.joinable()) g->filler = std::thread(filler_loop, g);
    {
        std::lock_guard<std::mutex> l(g->mu);
        g->slot[k].pending = true;
        g->req = k;
    }
    g->cv_req.notify_one();
}

int ingest_alloc(ps_ingest *g) {
    const ps_ingest_config_t &c = g->cfg;
    const size_t nb = (size_t)c.batch;
    for (int k = 0; k < 2; ++k) {
        ps_ingest::Slot &S = g->slot[k];
        HIPCHK(hipHostMalloc((void **)&S.ids_h, sizeof(int64_t) * nb * (c.F > 0 ? c.F : 1), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&S.wide_h, sizeof(int64_t) * nb * (c.F > 0 ? c.F : 1), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&S.dense_h, sizeof(float) * nb * (c.X > 0 ? c.X : 1), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&S.labels_h, sizeof(float) * nb, hipHostMallocDefault));
        HIPCHK(hipMalloc((void **)&S.ids_d, sizeof(int64_t) * nb * (c.F > 0 ? c.F : 1)));
        HIPCHK(hipMalloc((void **)&S.wide_d, sizeof(int64_t) * nb * (c.F > 0 ? c.F : 1)));
        HIPCHK(hipMalloc((void **)&S.dense_d, sizeof(float) * nb * (c.X > 0 ? c.X : 1)));
        HIPCHK(hipMalloc((void **)&S.labels_d, sizeof(float) * nb));
        HIPCHK(hipEventCreateWithFlags(&S.copied, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&S.consumed, hipEventDisableTiming));
    }
    HIPCHK(hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking));
    return PS_OK;
}

int ingest_open(ps_ingest *g) {
    g->st.clear(); g->en.clear();
    index_lines(g->data, g->len, g->cfg.offset, g->cfg.step, &g->st, &g->en);
    g->next_line = 0; g->cur = 0; g->primed = false;
    return PS_OK;
}

void ingest_quiesce(ps_ingest *g) {
    ingest_wait(g, 0); ingest_wait(g, 1);
    if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
}

}  // namespace

extern "C" int ps_ingest_create(ps_store_t *s, const ps_ingest_config_t *cfg, ps_ingest_t **out) {
    if (!s || !out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    *out = nullptr;
    PSCHK(check_cfg(cfg));
    PSCHK(store_enter(s));
    ps_ingest *g = new ps_ingest();
    g->s = s; g->cfg = *cfg;
    const int rc = ingest_alloc(g);
    if (rc != PS_OK) { ps_ingest_destroy(g); return rc; }
    if (cfg->threads > 1) g->pool = new Pool(cfg->threads);
    *out = g;
    return PS_OK;
}

extern "C" int ps_ingest_destroy(ps_ingest_t *g) {
    if (!g) return PS_OK;
    (void)hipSetDevice(g->s->device);
    ingest_quiesce(g);
    if (g->filler.joinable()) {
        { std::lock_guard<std::mutex> l(g->mu); g->stop = true; }
        g->cv_req.notify_all();
        g->filler.join();
    }
    for (int k = 0; k < 2; ++k) {
        ps_ingest::Slot &S = g->slot[k];
        if (S.ids_h) (void)hipHostFree(S.ids_h);
        if (S.wide_h) (void)hipHostFree(S.wide_h);
        if (S.dense_h) (void)hipHostFree(S.dense_h);
        if (S.labels_h) (void)hipHostFree(S.labels_h);
        if (S.ids_d) (void)hipFree(S.ids_d);
        if (S.wide_d) (void)hipFree(S.wide_d);
        if (S.dense_d) (void)hipFree(S.dense_d);
        if (S.labels_d) (void)hipFree(S.labels_d);
        if (S.copied) (void)hipEventDestroy(S.copied);
        if (S.consumed) (void)hipEventDestroy(S.consumed);
    }
    if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
    if (g->map) munmap(g->map, g->map_len);
    delete g->pool;
    delete g;
    return PS_OK;
}

extern "C" int ps_ingest_open_file(ps_ingest_t *g, const char *path) {
    if (!g || !path) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ingest_quiesce(g);
    if (g->map) { munmap(g->map, g->map_len); g->map = nullptr; }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return ps_set_err(PS_MISSING, "cannot open %s", path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return ps_set_err(PS_MISSING, "cannot stat %s", path); }
    g->map_len = (size_t)sb.st_size;
    if (g->map_len > 0) {
        g->map = mmap(nullptr, g->map_len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (g->map == MAP_FAILED) { g->map = nullptr; close(fd); return ps_set_err(PS_E_HIP, "mmap of %s failed", path); }
        (void)madvise(g->map, g->map_len, MADV_SEQUENTIAL);
    }
    close(fd);
    g->data = (const char *)g->map; g->len = g->map_len;
    return ingest_open(g);
}

extern "C" int ps_ingest_open_memory(ps_ingest_t *g, const char *text, size_t len) {
    if (!g || (!text && len)) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ingest_quiesce(g);
    g->copy.assign(text, text + len);
    g->data = g->copy.data(); g->len = len;
    return ingest_open(g);
}

extern "C" int ps_ingest_lines(ps_ingest_t *g, int64_t *n_lines) {
    if (!g || !n_lines) return ps_set_err(PS_E_BAD_ARG, "null argument");
    *n_lines = (int64_t)g->st.size();
    return PS_OK;
}

// The next batch (B <= cfg.batch samples) as device pointers, valid until the call after the next one.
// PS_MISSING at the end of the data (FileSource returns null -> DataSet ends the epoch); ps_ingest_reset rewinds.
extern "C" int ps_ingest_next(ps_ingest_t *g, ps_batch_t *out) {
    if (!g || !out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    if (!g->data && g->len == 0 && g->st.empty()) return ps_set_err(PS_E_STATE, "ps_ingest_open_* first");
    HIPCHK(hipSetDevice(g->s->device));
    if (!g->primed) { ingest_start(g, g->cur); g->primed = true; }
    ps_ingest::Slot &S = g->slot[g->cur];
    ingest_wait(g, g->cur);
    if (S.rc != PS_OK) return ps_set_err(S.rc, "%s", S.err);
    if (S.B == 0) return ps_set_err(PS_MISSING, "end of data");
    // the filler thread already waited for the copies to land: the batch is in HBM
    memset(out, 0, sizeof *out);
    out->B = S.B;
    out->ids = S.ids_d; out->offsets = nullptr; out->dense = g->cfg.X > 0 ? S.dense_d : nullptr;
    out->labels = S.labels_d; out->wide_ids = g->cfg.wide_size > 0 ? S.wide_d : nullptr;
    out->on_device = 1;
    g->cur ^= 1;
    // the slot being refilled was handed out by the previous call: its consumer kernels were enqueued on the
    // store's stream before this call; the filler waits (host side) for this event before overwriting it
    HIPCHK(hipEventRecord(g->slot[g->cur].consumed, g->s->stream));
    g->slot[g->cur].consumed_recorded = true;
    ingest_start(g, g->cur);
    return PS_OK;
}

extern "C" int ps_ingest_reset(ps_ingest_t *g) {
    if (!g) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ingest_quiesce(g);
    g->next_line = 0; g->cur = 0; g->primed = false;
    return PS_OK;
}

extern "C" int ps_ingest_stats(ps_ingest_t *g, double *parse_seconds, int64_t *lines, int64_t *bytes) {
    if (!g) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ingest_quiesce(g);
    if (parse_seconds) *parse_seconds = g->parse_s;
    if (lines) *lines = g->parsed_lines;
    if (bytes) *bytes = g->parsed_bytes;
    return PS_OK;
}
