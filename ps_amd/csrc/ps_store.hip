// ps_store.hip -- the GPU-resident KVStore shard: parameter tables in HBM,
// string-key get/put for parity with store/KVStore.java, updater registry
// (update/*.java), router (net/Mod.java).  Host-side C++; kernels live in
// kernels_*.hip.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include <mutex>

#include "ps_store.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

#include <dlfcn.h>
#include <stdlib.h>
namespace {
struct Roctx {
    int state = 0;                                   // 0 unknown, 1 on, -1 off
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
} g_roctx;
bool roctx_on() {
    if (g_roctx.state == 0) {
        g_roctx.state = -1;
        const char *e = getenv("PS_AMD_ROCTX");
        if (e && e[0] == '1') {
            void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                *(void **)(&g_roctx.push) = dlsym(h, "roctxRangePushA");
                *(void **)(&g_roctx.pop) = dlsym(h, "roctxRangePop");
                if (g_roctx.push && g_roctx.pop) g_roctx.state = 1;
            }
        }
    }
    return g_roctx.state == 1;
}
}  // namespace
void ps_roctx_push(const char *name) { if (roctx_on()) (void)g_roctx.push(name); }
void ps_roctx_pop() { if (g_roctx.state == 1) (void)g_roctx.pop(); }

std::recursive_mutex &ps_rt_mutex() {
    static std::recursive_mutex mu;
    return mu;
}

int ps_set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *ps_last_error(void) { return g_err; }
#ifndef PS_GEMM_LAB
#define PS_GEMM_LAB 0
#endif
extern "C" const char *ps_version(void) { return PS_GEMM_LAB ? "ps_amd 0.1 gfx950 hip +gemm_lab" : "ps_amd 0.1 gfx950 hip"; }
extern "C" int ps_device_count(int *count) {
    if (!count) return ps_set_err(PS_E_BAD_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return ps_set_err(PS_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return PS_OK;
}

float ps_xavier_scale(int in_dims, int out_dims) {
    // layer/EmbeddingField.java:40, layer/FcLayer.java:39,46
    return (float)(4 * (sqrt(6.0) / sqrt((double)(in_dims + out_dims))));
}

// ---------------------------------------------------------------------------
// update.Updater
// ---------------------------------------------------------------------------
extern "C" void ps_updater_default_adam(ps_updater_t *u) {
    memset(u, 0, sizeof *u);
    u->kind = PS_UPD_ADAM;                    // model/DNN.java:95
    u->alfa = (float)0.005; u->beta1 = (float)0.9; u->beta2 = (float)0.999; u->epsilon = (float)pow(10, -8);
}
extern "C" void ps_updater_default_ftrl(ps_updater_t *u) {
    memset(u, 0, sizeof *u);
    u->kind = PS_UPD_FTRL;                    // model/WideDeepNN.java:109
    u->alfa = 0.005f; u->beta = 1.0f; u->l1 = 0.001f; u->l2 = 0.001f;
}

// java.lang.Float.toString for the hyper-parameter range (shortest round-trip digits)
static int java_float_str(float v, char *buf, int cap) {
    if (v == 0.0f) return snprintf(buf, cap, "0.0");
    char digs[32];
    for (int prec = 1; prec <= 9; ++prec) {
        snprintf(digs, sizeof digs, "%.*e", prec - 1, (double)v);
        if (strtof(digs, nullptr) == v) break;
    }
    char mant[16]; int nm = 0, neg = 0; const char *p = digs;
    if (*p == '-') { neg = 1; ++p; }
    for (; *p && *p != 'e'; ++p) if (*p != '.') mant[nm++] = *p;
    mant[nm] = 0;
    const int ex = atoi(p + 1);
    while (nm > 1 && mant[nm - 1] == '0') mant[--nm] = 0;
    std::string o = neg ? "-" : "";
    const float a = fabsf(v);
    if (a >= 1e-3f && a < 1e7f) {
        if (ex >= 0) {
            for (int i = 0; i <= ex; ++i) o += i < nm ? mant[i] : '0';
            o += '.';
            if (nm > ex + 1) o.append(mant + ex + 1); else o += '0';
        } else {
            o += "0.";
            o.append((size_t)(-ex - 1), '0');
            o.append(mant);
        }
    } else {
        o += mant[0]; o += '.';
        if (nm > 1) o.append(mant + 1); else o += '0';
        o += "E" + std::to_string(ex);
    }
    return snprintf(buf, cap, "%s", o.c_str());
}

extern "C" int ps_updater_name(const ps_updater_t *u, char *buf, int cap) {
    if (!u || !buf) return ps_set_err(PS_E_BAD_ARG, "null argument");
    char a[32], b[32], c[32], d[32];
    if (u->kind == PS_UPD_ADAM) {             // update/AdamUpdater.java:72-74
        java_float_str(u->alfa, a, 32); java_float_str(u->beta1, b, 32);
        java_float_str(u->beta2, c, 32); java_float_str(u->epsilon, d, 32);
        snprintf(buf, cap, "adam@alfa:%s@beta1:%s@beta2:%s@epsilon:%s@", a, b, c, d);
    } else if (u->kind == PS_UPD_FTRL) {      // update/FtrlUpdater.java:78-80 (sic: "adam@")
        java_float_str(u->alfa, a, 32); java_float_str(u->beta, b, 32);
        java_float_str(u->l1, c, 32); java_float_str(u->l2, d, 32);
        snprintf(buf, cap, "adam@alfa:%s@beta:%s@l1:%s@l2:%s@", a, b, c, d);
    } else if (u->kind == PS_UPD_SIMPLE) {    // update/SimpleUpdater.java:24-26
        java_float_str(u->eta, a, 32);
        snprintf(buf, cap, "simple@eta:%s@", a);
    } else {
        return ps_set_err(PS_NO_UPDATER, "unknown updater kind %d", u->kind);
    }
    return PS_OK;
}

static bool between(const char *s, const char *open, float *out) {
    // StringUtils.substringBetween(str, open, "@") + Float.parseFloat
    const char *p = strstr(s, open);
    if (!p) return false;
    p += strlen(open);
    const char *e = strchr(p, '@');
    if (!e) return false;
    char tmp[48];
    const size_t n = (size_t)(e - p) < sizeof tmp - 1 ? (size_t)(e - p) : sizeof tmp - 1;
    memcpy(tmp, p, n); tmp[n] = 0;
    char *endp = nullptr;
    *out = strtof(tmp, &endp);
    return endp != tmp;
}

extern "C" int ps_updater_from_name(const char *name, ps_updater_t *out) {
    if (!name || !out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    memset(out, 0, sizeof *out);
    if (strncmp(name, "simple@", 7) == 0) {
        out->kind = PS_UPD_SIMPLE;
        if (!between(name, "eta:", &out->eta)) return ps_set_err(PS_NO_UPDATER, "updater is null: %s", name);
        return PS_OK;
    }
    if (strncmp(name, "adam@", 5) != 0) return ps_set_err(PS_NO_UPDATER, "updater is null: %s", name);
    if (strstr(name, "@beta1:")) {
        out->kind = PS_UPD_ADAM;
        if (between(name, "alfa:", &out->alfa) && between(name, "beta1:", &out->beta1) &&
            between(name, "beta2:", &out->beta2) && between(name, "epsilon:", &out->epsilon))
            return PS_OK;
    } else if (strstr(name, "@l1:")) {
        out->kind = PS_UPD_FTRL;
        if (between(name, "alfa:", &out->alfa) && between(name, "beta:", &out->beta) &&
            between(name, "l1:", &out->l1) && between(name, "l2:", &out->l2))
            return PS_OK;
    }
    return ps_set_err(PS_NO_UPDATER, "updater is null: %s", name);
}

UpdParams make_upd_params(const ps_updater_t &u) {
    UpdParams p;
    memset(&p, 0, sizeof p);
    p.kind = u.kind;
    p.alfa = u.alfa; p.beta1 = u.beta1; p.beta2 = u.beta2; p.eps = u.epsilon;
    p.c1 = 1 - u.beta1;            // float arithmetic, as "1 - beta1" in Java
    p.c2 = 1 - u.beta2;
    p.neg_alfa = -1 * u.alfa;
    p.beta = u.beta; p.l1 = u.l1; p.l2 = u.l2; p.eta = u.eta;
    return p;
}

// ---------------------------------------------------------------------------
// net.Mod / Router
// ---------------------------------------------------------------------------
extern "C" int32_t ps_java_string_hash(const char *key) {
    uint32_t h = 0;
    for (; key && *key; ++key) h = 31u * h + (uint32_t)(unsigned char)*key;
    return (int32_t)h;
}
extern "C" int ps_router_shard_key(const char *key, int nshards) {
    if (nshards <= 0) return 0;
    const int32_t h = ps_java_string_hash(key);
    int r = (int)(h % nshards);     // net/Mod.java:14 (Java % truncates) ...
    if (r < 0) r += nshards;        // ... with the floorMod fix
    return r;
}
extern "C" int ps_router_shard_id(int route_mode, int field, int64_t id, int nshards) {
    if (nshards <= 1) return 0;
    if (route_mode == PS_ROUTE_JAVA_STRING) {
        char key[64];
        snprintf(key, sizeof key, "emF%d.%lld.0", field, (long long)id);   // Float.toString of an integer id < 1e7
        return ps_router_shard_key(key, nshards);
    }
    int64_t r = id % nshards;
    if (r < 0) r += nshards;
    return (int)r;
}

// ---------------------------------------------------------------------------
// store
// ---------------------------------------------------------------------------
namespace {
struct StreamPool { std::mutex mu; std::vector<hipStream_t> free_[PS_MAX_DEVICES][3]; } g_stream_pool;
}
int pool_stream_acquire(int device, int cls, hipStream_t *out) {
    if (device < 0 || device >= PS_MAX_DEVICES || cls < 0 || cls > 2) return ps_set_err(PS_E_BAD_ARG, "stream pool: bad device / class");
    {
        std::lock_guard<std::mutex> lk(g_stream_pool.mu);
        auto &v = g_stream_pool.free_[device][cls];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return PS_OK; }
    }
    if (cls == 2) { HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking)); return PS_OK; }
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));      // lo = least urgent (numerically largest)
    HIPCHK(hipStreamCreateWithPriority(out, hipStreamNonBlocking, cls == 1 ? hi : lo));
    return PS_OK;
}
void pool_stream_release(int device, int cls, hipStream_t st) {
    if (!st) return;
    (void)hipStreamSynchronize(st);
    if (device < 0 || device >= PS_MAX_DEVICES || cls < 0 || cls > 2) { (void)hipStreamDestroy(st); return; }
    std::lock_guard<std::mutex> lk(g_stream_pool.mu);
    g_stream_pool.free_[device][cls].push_back(st);
}

int store_dev_alloc(ps_store *s, void **p, size_t bytes, bool zero) {
    RtGuard rt_guard;
    if (bytes == 0) bytes = 16;
    HIPCHK(hipMalloc(p, bytes));
    if (zero) HIPCHK(hipMemsetAsync(*p, 0, bytes, s->stream));
    s->bytes += (int64_t)bytes;
    return PS_OK;
}

extern "C" int ps_store_create(int device, uint64_t seed, ps_store_t **out) {
    RtGuard rt_guard;
    if (!out) return ps_set_err(PS_E_BAD_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return ps_set_err(PS_E_HIP, "no HIP device (hipGetDeviceCount: %s, %d devices): the HIP path is required",
                          hipGetErrorString(e), n);
    if (device < 0 || device >= n) return ps_set_err(PS_E_BAD_ARG, "device %d out of range [0,%d)", device, n);
    HIPCHK(hipSetDevice(device));
    ps_store *s = new ps_store();
    s->device = device;
    s->seed = seed;
    PSCHK(pool_stream_acquire(device, 1, &s->own_stream));      // main chain: most urgent (from the process-wide pool: ps_store.h)
    s->stream = s->own_stream;
    PSCHK(store_dev_alloc(s, (void **)&s->err_dev, 8 * sizeof(int), true));      // [0] bad ids | [1] timed-out device waits, [2] last, [3] first | [4] XCD mismatches
    ps_updater_t a;
    ps_updater_default_adam(&a);
    s->updaters["default"] = a;
    *out = s;
    return PS_OK;
}

extern "C" int ps_store_destroy(ps_store_t *s) {
    RtGuard rt_guard;
    if (!s) return PS_OK;
    (void)hipSetDevice(s->device);
    (void)store_settle(s);
    (void)hipStreamSynchronize(s->stream);
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fr(s->emb.W); fr(s->emb.state); fr(s->emb.row_base_dev);
    fr(s->emb.owner_dev); fr(s->emb.local_dev); fr(s->emb.grow_base_dev);
    fr(s->wide.W); fr(s->wide.state); fr(s->wide.touched); fr(s->wide.bias); fr(s->wide.bias_state);
    for (auto &f : s->fc) { fr(f.W); fr(f.Wt); fr(f.Wp); fr(f.S1); fr(f.S2); fr(f.pending); }
    {
        ps_store::OpScratch &o = s->ops;
        sort_ws_free(o.ws);
        fr(o.part); fr(o.masked); fr(o.keys); fr(o.ents); fr(o.ent_bag); fr(o.seg_start); fr(o.seg_id); fr(o.nseg); fr(o.uniq_row);
        fr(o.partials); fr(o.partials2); fr(o.grads);
    }
    fr(s->err_dev); fr(s->idx_dev); fr(s->rowbuf_dev);
    sort_ws_free(s->push_ws);
    fr(s->push_keys); fr(s->push_ents); fr(s->push_seg_start); fr(s->push_seg_id); fr(s->push_nseg);
    fr(s->push_mask); fr(s->push_pos);
    pool_stream_release(s->device, 1, s->own_stream);    // (an adopted stream belongs to the host: only the store's own goes back)
    pool_stream_release(s->device, 1, s->prefetch_stream);
    delete s;
    return PS_OK;
}

// Device-side waits (launch_spin_until, the GEMMs' end wait, the waits folded into the dW GEMM / the dense update) need
// the streams of ONE model to run concurrently.  Several models driving one device from one process (tests run N ranks
// as N threads on one GPU) multiply the streams beyond the runtime's hardware queues (GPU_MAX_HW_QUEUES, 4 per
// priority): streams of different models then share a queue, and a waiter of one model can sit in front of another
// model's releaser -- the likely cause of round 1's hang in the 4-rank thread test.  With more than one live model on a
// device every join takes its event form.
std::atomic<int> g_models_on_device[PS_MAX_DEVICES];
bool dev_waits_ok(const ps_store *s) {
    const bool one_model = s->device >= 0 && s->device < PS_MAX_DEVICES && g_models_on_device[s->device].load() <= 1;
    if (g_dev_wait && !s->dev_wait_off && !one_model) {
        // a second live model (a train + an eval model, the two-model prefetch) silently costs every join ~3-10 us: say so once
        static std::atomic<bool> said{false};
        if (!said.exchange(true) && getenv("PS_AMD_QUIET") == nullptr)
            fprintf(stderr, "[ps_amd] more than one model drives device %d from this process: stream joins take their event form (slower steps); "
                            "ps_store_join_mode() reports the mode\n", s->device);
    }
    return g_dev_wait && !s->dev_wait_off && one_model;
}

int store_settle(ps_store *s) {
    if (s->pending_ev) {
        hipEvent_t e = s->pending_ev;
        s->pending_ev = nullptr; s->pending_flag = nullptr; s->pending_start = nullptr;
        HIPCHK(hipStreamWaitEvent(s->stream, e, 0));
    }
    return PS_OK;
}
int store_enter(ps_store *s) {
    HIPCHK(hipSetDevice(s->device));
    return store_settle(s);
}

int store_check_bad_ids(ps_store *s) {
    int err[5] = {0, 0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(err, s->err_dev, sizeof err, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (err[1]) {
        // a bounded device-side wait gave up: whatever ran behind it ran without one of its dependencies.  From here
        // on this store's models use the event form of every join (the caller decides what to do with the tables:
        // the steps since the last successful check are suspect).
        // (words 1..3 only: the bad-id count and the XCD-mismatch count stay for the next check -- ADVICE r3 -- but are named here too)
        HIPCHK(hipMemsetAsync(s->err_dev + 1, 0, 3 * sizeof(int), s->stream));
        s->dev_wait_off = true;
        s->wait_timeouts += err[1];
        if (err[4]) { HIPCHK(hipMemsetAsync(s->err_dev + 4, 0, sizeof(int), s->stream)); s->fwd_pair_off = true; }
        return ps_set_err(PS_E_STATE, "%d device-side wait(s) timed out after %.0f ms (first: wait %d, last: wait %d); the steps since the last check ran "
                          "without a dependency -- this store now joins its streams by events%s%s", err[1], (double)g_spin_timeout_ticks * 1e-5, err[3] - 1000, err[2],
                          err[0] ? " [ids outside their table were seen too: reported by the next check]" : "",
                          err[4] ? " [workgroups of the paired forward GEMM off their XCD too: the forward GEMMs are launched one by one from now on]" : "");
    }
    if (err[4]) {
        // k_fc_fwd_pair hands a row panel from one workgroup to another through the L2 of "their" XCD, which it takes to be
        // blockIdx % 8: that did not hold, so the second layer may have read stale rows.  Never again on this store.
        HIPCHK(hipMemsetAsync(s->err_dev + 4, 0, sizeof(int), s->stream));
        s->fwd_pair_off = true;
        return ps_set_err(PS_E_STATE, "%d workgroups of the paired forward GEMM did not run on XCD (blockIdx %% 8): the steps since the last check are "
                          "suspect; the forward GEMMs are launched one by one from now on", err[4]);
    }
    if (err[0]) {
        HIPCHK(hipMemsetAsync(s->err_dev, 0, sizeof(int), s->stream));
        return ps_set_err(PS_MISSING, "%d ids were outside their table (treated as id 0)", err[0]);
    }
    return PS_OK;
}

extern "C" int ps_store_device(const ps_store_t *s) { return s ? s->device : -1; }
extern "C" int64_t ps_store_global_step(const ps_store_t *s) { return s ? s->global_step : -1; }
extern "C" int64_t ps_store_bytes(const ps_store_t *s) { return s ? s->bytes : -1; }

extern "C" int ps_store_set_updater(ps_store_t *s, const char *key, const ps_updater_t *u) {
    if (!s || !key || !u) return ps_set_err(PS_E_BAD_ARG, "null argument");
    if (u->kind < PS_UPD_ADAM || u->kind > PS_UPD_SIMPLE) return ps_set_err(PS_NO_UPDATER, "unknown updater kind %d", u->kind);
    s->updaters[key] = *u;
    return PS_OK;
}

int store_resolve_updater(const ps_store *s, const char *key, ps_updater_t *out) {
    // store/KVStore.java:242-252
    auto it = s->updaters.find(key);
    if (it != s->updaters.end()) { *out = it->second; return PS_OK; }
    const ps_updater_t *hit = nullptr;
    for (auto &kv : s->updaters)
        if (kv.first != "default" && strncmp(key, kv.first.c_str(), kv.first.size()) == 0) hit = &kv.second;
    if (!hit) {
        it = s->updaters.find("default");
        if (it == s->updaters.end()) return ps_set_err(PS_NO_UPDATER, "no updater for %s", key);
        hit = &it->second;
    }
    *out = *hit;
    return PS_OK;
}

int store_fill_field_upd(const ps_store *s, UpdParams *upd, FieldUpd *fu, bool *stateful) {
    memset(fu, 0, sizeof *fu);
    fu->row_base = s->emb.row_base_dev; fu->F = s->emb.F; fu->ngroups = 1;
    const int F = s->emb.F;
    bool field_level = false;
    struct RowKey { int field; int64_t id; ps_updater_t u; };
    std::vector<RowKey> row_keys;
    for (auto &kv : s->updaters) {
        const std::string &k = kv.first;
        if (k.compare(0, 3, "emF") != 0 || k.size() == 3) continue;
        const size_t dot = k.find('.');
        if (dot != std::string::npos && dot + 1 < k.size()) {
            // "emF<f>.<id>.0": a row's own key, KVStore.update(Map)'s exact match (store/KVStore.java:242); anything that ends
            // inside the id is a prefix of a family of rows (emF3.1 -> 1, 10 .. 19, 100 ..): refused, not guessed at
            ParsedKey pk;
            if (!store_parse_key(k.c_str(), &pk) || pk.kind != 0 || k.compare(k.size() - 2, 2, ".0") != 0)
                return ps_set_err(PS_E_UNSUPPORTED, "updater key %s is neither a field prefix (\"emF<f>.\") nor a row's key (\"emF<f>.<id>.0\")", k.c_str());
            row_keys.push_back({pk.idx, pk.id, kv.second});
            continue;
        }
        field_level = true;
    }
    ps_updater_t groups[PS_EMB_UPD_GROUPS];
    int ng = 0;
    bool st = false;
    auto group_of = [&](const ps_updater_t &u) -> int {
        int g = 0;
        while (g < ng && memcmp(&groups[g], &u, sizeof u) != 0) ++g;
        if (g == ng) {
            if (ng == PS_EMB_UPD_GROUPS) return -1;
            groups[ng++] = u;
        }
        st = st || u.kind != PS_UPD_SIMPLE;
        return g;
    };
    if (!field_level || F <= 0) {          // one updater for every field: the "emF" prefix (or "default")
        ps_updater_t u0;
        PSCHK(store_resolve_updater(s, "emF", &u0));
        (void)group_of(u0);
        if (row_keys.empty()) {
            *upd = make_upd_params(u0);
            if (stateful) *stateful = st;
            return PS_OK;
        }
        if (F > 64) return ps_set_err(PS_E_UNSUPPORTED, "row-level updaters need F <= 64 (F = %d)", F);
    } else {
        if (F > 64) return ps_set_err(PS_E_UNSUPPORTED, "per-field updaters need F <= 64 (F = %d)", F);
        for (int f = 0; f < F; ++f) {
            char probe[32];
            snprintf(probe, sizeof probe, "emF%d.", f);
            ps_updater_t u;
            PSCHK(store_resolve_updater(s, probe, &u));
            const int g = group_of(u);
            if (g < 0) return ps_set_err(PS_E_UNSUPPORTED, "more than %d distinct embedding updaters", PS_EMB_UPD_GROUPS);
            fu->grp[f] = (unsigned char)g;
        }
    }
    for (const RowKey &rk : row_keys) {
        if (rk.field < 0 || rk.field >= F) continue;                    // (names no row of this store)
        const int64_t row = store_local_row(s, rk.field, rk.id);
        if (row < 0) continue;                                          // another shard's row (or beyond the vocabulary)
        const int g = group_of(rk.u);
        if (g < 0) return ps_set_err(PS_E_UNSUPPORTED, "more than %d distinct embedding updaters", PS_EMB_UPD_GROUPS);
        if (g == fu->grp[rk.field]) continue;                           // the row's field already resolves to this updater
        if (fu->nover == PS_EMB_ROW_OVERRIDES)
            return ps_set_err(PS_E_UNSUPPORTED, "more than %d embedding rows with an updater of their own (exact updater keys \"emF<f>.<id>.0\")", PS_EMB_ROW_OVERRIDES);
        fu->over_row[fu->nover] = (uint32_t)row; fu->over_grp[fu->nover] = (unsigned char)g;
        ++fu->nover;
    }
    *upd = make_upd_params(groups[0]);
    for (int g = 1; g < ng; ++g) fu->alt[g - 1] = make_upd_params(groups[g]);
    fu->ngroups = ng;
    if (stateful) *stateful = st;
    return PS_OK;
}

static int64_t local_count(int64_t rows, int shard, int nshards) {
    return rows > shard ? (rows - shard + nshards - 1) / nshards : 0;
}

extern "C" int ps_store_create_embedding(ps_store_t *s, int F, const int64_t *rows, int D, int state_slots,
                                         int shard, int nshards, int route_mode) {
    if (!s || !rows || F <= 0 || D <= 0) return ps_set_err(PS_E_BAD_ARG, "bad embedding shape");
    if (s->emb.W) return ps_set_err(PS_E_STATE, "embedding tables already exist");
    if (nshards < 1 || shard < 0 || shard >= nshards) return ps_set_err(PS_E_BAD_ARG, "bad shard %d/%d", shard, nshards);
    if (route_mode != PS_ROUTE_ID_MOD && route_mode != PS_ROUTE_JAVA_STRING) return ps_set_err(PS_E_BAD_ARG, "bad route_mode %d", route_mode);
    if (route_mode == PS_ROUTE_JAVA_STRING && nshards > 1) {
        // the key string is "emF<f>.<id>.0" only while Float.toString(id) is plain decimal (ids < 10^7)
        if (nshards > 255) return ps_set_err(PS_E_UNSUPPORTED, "java_string routing supports up to 255 shards");
        for (int f = 0; f < F; ++f)
            if (rows[f] > 10000000) return ps_set_err(PS_E_UNSUPPORTED, "java_string routing needs vocabularies <= 10^7 (Float.toString switches to exponent form)");
    }
    if ((D % 4 == 0 && D > 256) || (D % 4 != 0 && D > 64)) return ps_set_err(PS_E_UNSUPPORTED, "embedding dim %d too wide for one wave per row", D);
    if (state_slots != 0 && state_slots != 2) return ps_set_err(PS_E_BAD_ARG, "state_slots must be 0 or 2");
    PSCHK(store_enter(s));
    EmbTables &e = s->emb;
    e.F = F; e.D = D; e.state_slots = state_slots; e.shard = shard; e.nshards = nshards; e.route_mode = route_mode;
    e.rows.assign(rows, rows + F);
    e.row_base.assign(F + 1, 0);
    for (int f = 0; f < F; ++f)
        if (rows[f] <= 0) return ps_set_err(PS_E_BAD_ARG, "rows[%d] = %lld", f, (long long)rows[f]);
    if (e.java_route()) {
        // Mod.shard(key) = String.hashCode(key) mod n for every (field, id) -- once, on the host (31-polynomial over
        // "emF<f>." + decimal id + ".0"; ps_router_shard_key is the same arithmetic on a C string)
        e.grow_base.assign(F + 1, 0);
        for (int f = 0; f < F; ++f) e.grow_base[f + 1] = e.grow_base[f] + rows[f];
        const int64_t G = e.grow_base[F];
        e.owner_h.resize((size_t)G); e.local_h.resize((size_t)G);
        e.owner_cnt.assign((size_t)nshards * F, 0);
        for (int f = 0; f < F; ++f) {
            char pre[32];
            snprintf(pre, sizeof pre, "emF%d.", f);
            uint32_t hp = 0;
            for (const char *c = pre; *c; ++c) hp = 31u * hp + (uint32_t)(unsigned char)*c;
            for (int64_t id = 0; id < rows[f]; ++id) {
                char dig[24];
                int nd = 0;
                int64_t v = id;
                do { dig[nd++] = (char)('0' + v % 10); v /= 10; } while (v);
                uint32_t h = hp;
                while (nd) h = 31u * h + (uint32_t)dig[--nd];
                h = 31u * h + (uint32_t)'.';
                h = 31u * h + (uint32_t)'0';
                int o = (int)((int32_t)h % nshards);
                if (o < 0) o += nshards;                 // floorMod (net/Mod.java would go negative)
                const size_t g = (size_t)(e.grow_base[f] + id);
                e.owner_h[g] = (uint8_t)o;
                e.local_h[g] = (uint32_t)e.owner_cnt[(size_t)o * F + f]++;
            }
            e.row_base[f + 1] = e.row_base[f] + e.owner_cnt[(size_t)shard * F + f];
        }
        e.ids_local_h.resize((size_t)e.row_base[F]);
        for (int f = 0; f < F; ++f)
            for (int64_t id = 0; id < rows[f]; ++id) {
                const size_t g = (size_t)(e.grow_base[f] + id);
                if (e.owner_h[g] == shard) e.ids_local_h[(size_t)(e.row_base[f] + e.local_h[g])] = (uint32_t)id;
            }
    } else {
        for (int f = 0; f < F; ++f) e.row_base[f + 1] = e.row_base[f] + local_count(rows[f], shard, nshards);
    }
    e.total_rows = e.row_base[F];
    if (e.total_rows >= (1ll << 32)) return ps_set_err(PS_E_UNSUPPORTED, "more than 2^32 rows on one shard");
    PSCHK(store_dev_alloc(s, (void **)&e.W, sizeof(float) * (size_t)e.total_rows * D, false));
    if (state_slots) PSCHK(store_dev_alloc(s, (void **)&e.state, sizeof(float) * (size_t)e.total_rows * 2 * D, true));
    PSCHK(store_dev_alloc(s, (void **)&e.row_base_dev, sizeof(int64_t) * (F + 1), false));
    HIPCHK(hipMemcpyAsync(e.row_base_dev, e.row_base.data(), sizeof(int64_t) * (F + 1), hipMemcpyHostToDevice, s->stream));
    const float scale = ps_xavier_scale(1, D);          // EmbeddingField.java:40 (in = 1, out = D)
    uint32_t *ids_dev = nullptr;
    if (e.java_route() && e.total_rows > 0) {
        RtGuard rt_guard;
        HIPCHK(hipMalloc((void **)&ids_dev, sizeof(uint32_t) * (size_t)e.total_rows));
        HIPCHK(hipMemcpyAsync(ids_dev, e.ids_local_h.data(), sizeof(uint32_t) * (size_t)e.total_rows, hipMemcpyHostToDevice, s->stream));
    }
    for (int f = 0; f < F; ++f)
        PSCHK(launch_init_emb(e.W + (size_t)e.row_base[f] * D, e.row_base[f + 1] - e.row_base[f], D, s->seed,
                              (uint64_t)f, scale, shard, nshards, ids_dev ? ids_dev + e.row_base[f] : nullptr, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (ids_dev) { RtGuard rt_guard; (void)hipFree(ids_dev); }
    return PS_OK;
}

extern "C" int ps_store_create_wide(ps_store_t *s, int64_t wide_size) {
    if (!s || wide_size <= 0) return ps_set_err(PS_E_BAD_ARG, "bad wide size");
    if (s->wide.W) return ps_set_err(PS_E_STATE, "wide table already exists");
    PSCHK(store_enter(s));
    WideTable &w = s->wide;
    w.rows = wide_size;
    // layer/LRLayer.java:37-52: weights and bias start at zero
    PSCHK(store_dev_alloc(s, (void **)&w.W, sizeof(float) * (size_t)wide_size, true));
    PSCHK(store_dev_alloc(s, (void **)&w.state, sizeof(float) * 2 * (size_t)wide_size, true));
    PSCHK(store_dev_alloc(s, (void **)&w.touched, (size_t)wide_size + 16, true));
    PSCHK(store_dev_alloc(s, (void **)&w.bias, sizeof(float) * 4, true));
    PSCHK(store_dev_alloc(s, (void **)&w.bias_state, sizeof(float) * 4, true));
    if (!s->updaters.count("wide.weights")) {            // model/WideDeepNN.java:109-112
        ps_updater_t f;
        ps_updater_default_ftrl(&f);
        s->updaters["wide.weights"] = f;
        s->updaters["wide.bias"] = f;
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}

extern "C" int ps_store_create_fc(ps_store_t *s, int layer, int in_dims, int out_dims) {
    if (!s || layer < 0 || layer >= 8 || in_dims <= 0 || out_dims <= 0) return ps_set_err(PS_E_BAD_ARG, "bad fc layer");
    PSCHK(store_enter(s));
    if ((int)s->fc.size() <= layer) s->fc.resize(layer + 1);
    FcParams &f = s->fc[layer];
    if (f.present) {
        if (f.K != in_dims || f.N != out_dims) return ps_set_err(PS_E_STATE, "fc%d exists with another shape", layer);
        return PS_OK;
    }
    f.K = in_dims; f.N = out_dims;
    f.Kpad = (int)round_up(in_dims + 1, 16);
    f.ldw = (int)round_up(out_dims, 16);
    const size_t wn = (size_t)f.Kpad * f.ldw, wtn = (size_t)f.N * f.Kpad;
    PSCHK(store_dev_alloc(s, (void **)&f.W, sizeof(float) * wn, true));
    PSCHK(store_dev_alloc(s, (void **)&f.Wt, sizeof(float) * wtn, true));
#if PS_GEMM_LAB
    if (f.N % 16 == 0) PSCHK(store_dev_alloc(s, (void **)&f.Wp, sizeof(float) * wtn, true));      // (fragment order: kernels_panel.hip, lab build)
#endif
    PSCHK(store_dev_alloc(s, (void **)&f.S1, sizeof(float) * wn, true));
    PSCHK(store_dev_alloc(s, (void **)&f.S2, sizeof(float) * wn, true));
    // layer/FcLayer.java:34-50: W ~ xavier(in+out), b ~ xavier(in+1)
    PSCHK(launch_init_dense(f.W, f.Wt, f.K, f.N, f.ldw, f.Kpad, s->seed, PS_TABLE_FC(layer), ps_xavier_scale(in_dims, out_dims),
                            PS_TABLE_FC(layer) + 1, ps_xavier_scale(in_dims, 1), s->stream));
    PSCHK(launch_pack_w(f.Wt, f.Wp, f.N, f.Kpad, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    f.present = true;
    return PS_OK;
}

int64_t store_local_row(const ps_store *s, int field, int64_t id) {
    const EmbTables &e = s->emb;
    if (field < 0 || field >= e.F || id < 0 || id >= e.rows[field]) return -1;
    if (e.java_route()) {
        const size_t g = (size_t)(e.grow_base[field] + id);
        return e.owner_h[g] == e.shard ? e.row_base[field] + (int64_t)e.local_h[g] : -1;
    }
    if (ps_router_shard_id(e.route_mode, field, id, e.nshards) != e.shard) return -1;
    return e.row_base[field] + id / e.nshards;
}

int store_ensure_scratch(ps_store *s, int64_t rows, int D) {
    RtGuard rt_guard;
    if (rows <= s->scratch_rows && D <= s->scratch_D) return PS_OK;
    if (s->idx_dev) (void)hipFree(s->idx_dev);
    if (s->rowbuf_dev) (void)hipFree(s->rowbuf_dev);
    s->idx_dev = nullptr; s->rowbuf_dev = nullptr;
    const int64_t r = rows > s->scratch_rows ? rows : s->scratch_rows;
    const int d = D > s->scratch_D ? D : s->scratch_D;
    HIPCHK(hipMalloc((void **)&s->idx_dev, sizeof(int64_t) * (size_t)r));
    HIPCHK(hipMalloc((void **)&s->rowbuf_dev, sizeof(float) * (size_t)r * d));
    s->scratch_rows = r; s->scratch_D = d;
    return PS_OK;
}

static int rows_io(ps_store *s, float *table, int64_t row_stride, int64_t col_off, int D,
                   const std::vector<int64_t> &lrows, float *host, int to_table) {
    const int64_t n = (int64_t)lrows.size();
    if (n == 0) return PS_OK;
    PSCHK(store_enter(s));
    PSCHK(store_ensure_scratch(s, n, D));
    HIPCHK(hipMemcpyAsync(s->idx_dev, lrows.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, s->stream));
    if (to_table) HIPCHK(hipMemcpyAsync(s->rowbuf_dev, host, sizeof(float) * n * D, hipMemcpyHostToDevice, s->stream));
    PSCHK(launch_rows_copy(table, row_stride, col_off, s->idx_dev, n, D, s->rowbuf_dev, to_table, s->stream));
    if (!to_table) HIPCHK(hipMemcpyAsync(host, s->rowbuf_dev, sizeof(float) * n * D, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}

static int emb_rows_io(ps_store *s, int field, const int64_t *ids, int64_t n, int which, float *host, int to_table) {
    if (!s || !ids || !host || n < 0) return ps_set_err(PS_E_BAD_ARG, "null argument");
    EmbTables &e = s->emb;
    if (!e.W) return ps_set_err(PS_MISSING, "no embedding tables");
    if (which < 0 || which > 2 || (which > 0 && !e.state)) return ps_set_err(PS_E_BAD_ARG, "bad state slot %d", which);
    std::vector<int64_t> lr((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        lr[i] = store_local_row(s, field, ids[i]);
        if (lr[i] < 0) return ps_set_err(PS_MISSING, "emF%d.%lld is not held by this shard", field, (long long)ids[i]);
    }
    if (which == 0) return rows_io(s, e.W, e.D, 0, e.D, lr, host, to_table);
    return rows_io(s, e.state, 2 * (int64_t)e.D, (which - 1) * (int64_t)e.D, e.D, lr, host, to_table);
}

extern "C" int ps_store_get_rows(ps_store_t *s, int field, const int64_t *ids, int64_t n, int which, float *out) {
    return emb_rows_io(s, field, ids, n, which, out, 0);
}
extern "C" int ps_store_put_rows(ps_store_t *s, int field, const int64_t *ids, int64_t n, int which, const float *val) {
    return emb_rows_io(s, field, ids, n, which, const_cast<float *>(val), 1);
}

static int wide_rows_io(ps_store *s, const int64_t *ids, int64_t n, int which, float *host, int to_table) {
    if (!s || !ids || !host || n < 0) return ps_set_err(PS_E_BAD_ARG, "null argument");
    WideTable &w = s->wide;
    if (!w.W) return ps_set_err(PS_MISSING, "no wide table");
    if (which < 0 || which > 2) return ps_set_err(PS_E_BAD_ARG, "bad state slot %d", which);
    std::vector<int64_t> lr(ids, ids + n);
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= w.rows) return ps_set_err(PS_MISSING, "wide.weights.%lld out of range", (long long)ids[i]);
    if (which == 0) return rows_io(s, w.W, 1, 0, 1, lr, host, to_table);
    return rows_io(s, w.state, 2, which - 1, 1, lr, host, to_table);
}
extern "C" int ps_store_get_wide(ps_store_t *s, const int64_t *ids, int64_t n, int which, float *out) {
    return wide_rows_io(s, ids, n, which, out, 0);
}
extern "C" int ps_store_put_wide(ps_store_t *s, const int64_t *ids, int64_t n, int which, const float *val) {
    return wide_rows_io(s, ids, n, which, const_cast<float *>(val), 1);
}

// ---- string keys ------------------------------------------------------------
// "emF<f>.<id>.0" | "wide.weights.<id>.0" | "wide.bias" | "fc<i>.weights" | "fc<i>.bias"
static bool parse_float_id(const char *p, int64_t *id) {
    char *e = nullptr;
    const double v = strtod(p, &e);                  // "28305.0" (also accepts "2.8305E4")
    if (e == p || *e != 0) return false;
    if (v != floor(v)) return false;
    *id = (int64_t)v;
    return true;
}
static bool parse_key(const char *key, ParsedKey *k) {
    if (strncmp(key, "emF", 3) == 0) {
        char *e = nullptr;
        const long f = strtol(key + 3, &e, 10);
        if (e == key + 3 || *e != '.') return false;
        k->kind = 0; k->idx = (int)f;
        return parse_float_id(e + 1, &k->id);
    }
    if (strncmp(key, "wide.weights.", 13) == 0) { k->kind = 1; k->idx = 0; return parse_float_id(key + 13, &k->id); }
    if (strcmp(key, "wide.bias") == 0) { k->kind = 2; k->idx = 0; k->id = 0; return true; }
    if (strncmp(key, "fc", 2) == 0) {
        char *e = nullptr;
        const long i = strtol(key + 2, &e, 10);
        if (e == key + 2) return false;
        k->idx = (int)i; k->id = 0;
        if (strcmp(e, ".weights") == 0) { k->kind = 3; return true; }
        if (strcmp(e, ".bias") == 0) { k->kind = 4; return true; }
    }
    return false;
}
bool store_parse_key(const char *key, ParsedKey *k) { return parse_key(key, k); }

static int fc_io(ps_store *s, int layer, int bias, float *host, int cap, int *len, int to_dev) {
    if (layer < 0 || layer >= (int)s->fc.size() || !s->fc[layer].present) return ps_set_err(PS_MISSING, "fc%d absent", layer);
    FcParams &f = s->fc[layer];
    const int n = bias ? f.N : f.K * f.N;
    if (len) *len = n;
    if (cap < n) return ps_set_err(PS_E_BAD_ARG, "buffer too small: %d < %d", cap, n);
    PSCHK(store_enter(s));
    const int krows = bias ? 1 : f.K;
    float *src = f.W + (bias ? (size_t)f.K * f.ldw : 0);
    if (!to_dev) {
        HIPCHK(hipMemcpy2DAsync(host, sizeof(float) * f.N, src, sizeof(float) * f.ldw, sizeof(float) * f.N, krows,
                                hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        return PS_OK;
    }
    HIPCHK(hipMemcpy2DAsync(src, sizeof(float) * f.ldw, host, sizeof(float) * f.N, sizeof(float) * f.N, krows,
                            hipMemcpyHostToDevice, s->stream));
    // keep the transposed copy in step: Wt[n][k]
    std::vector<float> t((size_t)f.N * krows);
    for (int k = 0; k < krows; ++k)
        for (int n2 = 0; n2 < f.N; ++n2) t[(size_t)n2 * krows + k] = host[(size_t)k * f.N + n2];
    HIPCHK(hipMemcpy2DAsync(f.Wt + (bias ? f.K : 0), sizeof(float) * f.Kpad, t.data(), sizeof(float) * krows,
                            sizeof(float) * krows, f.N, hipMemcpyHostToDevice, s->stream));
    PSCHK(launch_pack_w(f.Wt, f.Wp, f.N, f.Kpad, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}

extern "C" int ps_store_get(ps_store_t *s, const char *key, float *out, int cap, int *len) {
    if (!s || !key || !out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ParsedKey k;
    if (!parse_key(key, &k)) return ps_set_err(PS_MISSING, "unknown key %s", key);
    switch (k.kind) {
    case 0:
        if (!s->emb.W) return ps_set_err(PS_MISSING, "no embedding tables");
        if (len) *len = s->emb.D;
        if (cap < s->emb.D) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
        return ps_store_get_rows(s, k.idx, &k.id, 1, 0, out);
    case 1:
        if (len) *len = 1;
        if (cap < 1) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
        return ps_store_get_wide(s, &k.id, 1, 0, out);
    case 2:
        if (!s->wide.W) return ps_set_err(PS_MISSING, "no wide table");
        if (len) *len = 1;
        if (cap < 1) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
        PSCHK(store_enter(s));
        HIPCHK(hipMemcpyAsync(out, s->wide.bias, sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        return PS_OK;
    case 3: return fc_io(s, k.idx, 0, out, cap, len, 0);
    default: return fc_io(s, k.idx, 1, out, cap, len, 0);
    }
}

extern "C" int ps_store_put(ps_store_t *s, const char *key, const float *val, int n) {
    if (!s || !key || !val) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ParsedKey k;
    if (!parse_key(key, &k)) return ps_set_err(PS_MISSING, "unknown key %s", key);
    int len = 0;
    switch (k.kind) {
    case 0:
        if (!s->emb.W) return ps_set_err(PS_MISSING, "no embedding tables");
        if (n != s->emb.D) return ps_set_err(PS_E_BAD_ARG, "%s wants %d floats", key, s->emb.D);
        return ps_store_put_rows(s, k.idx, &k.id, 1, 0, val);
    case 1:
        if (n != 1) return ps_set_err(PS_E_BAD_ARG, "%s wants 1 float", key);
        return ps_store_put_wide(s, &k.id, 1, 0, val);
    case 2:
        if (!s->wide.W) return ps_set_err(PS_MISSING, "no wide table");
        PSCHK(store_enter(s));
        HIPCHK(hipMemcpyAsync(s->wide.bias, val, sizeof(float), hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        return PS_OK;
    case 3: return fc_io(s, k.idx, 0, const_cast<float *>(val), n, &len, 1);
    default: return fc_io(s, k.idx, 1, const_cast<float *>(val), n, &len, 1);
    }
}
