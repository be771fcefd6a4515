// ps_keyed.hip -- the PS server's side of the reference's wire, by STRING KEY:
//   PServer.push(key, gradient, isAsync, updaterKey)   net/PServer.java:164-195   (KVStore.sum)
//   PServer.psUpdate()                                 net/PServer.java:197-214   (KVStore.update(updater, key) per pushed key)
// for any key the store holds: "emF<f>.<id>.0", "wide.weights.<id>.0", "wide.bias", "fc<i>.weights", "fc<i>.bias".
// ps_store_push_update takes the messages of one BSP round (every worker's pushes, in arrival order) -- or, async,
// the messages to apply one by one -- and runs sum / divi(count) / Updater.update on the device with the arithmetic of
// the hot path's kernels: embedding rows through the stable-sort path of ps_shard_apply_push (mean over a key's pushes
// in arrival order), dense tensors through k_dense_update with the pushed gradients as its slabs (sum in arrival order,
// / count), wide keys through k_wide_list.  The host only parses keys and groups message indices (integers).
// ps_amd/ps_server.py (the gRPC facade over ps.proto) is the caller.
#include <string.h>

#include <map>
#include <vector>

#include "ps_store.h"

namespace {

struct DevBuf {            // scratch for one call (the facade is not a hot path)
    void *p = nullptr;
    ~DevBuf() { if (p) { RtGuard g; (void)hipFree(p); } }
    int alloc(size_t bytes) { RtGuard g; HIPCHK(hipMalloc(&p, bytes ? bytes : 4)); return PS_OK; }
};

}  // namespace

// The number of floats a push / upsert of `key` must carry on THIS shard (PS_MISSING: a key the store does not hold).
// The gRPC facade validates a BSP push with it when the push ARRIVES (the reference answers 500 there), not when the
// round's last barrier applies it.
extern "C" int ps_store_key_length(const ps_store_t *s, const char *key, int *len_out) {
    if (!s || !key || !len_out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ParsedKey k;
    if (!store_parse_key(key, &k)) return ps_set_err(PS_MISSING, "unknown key %s", key);
    if (k.kind == 0) {
        if (!s->emb.W || k.idx < 0 || k.idx >= s->emb.F) return ps_set_err(PS_MISSING, "%s: no such field", key);
        if (store_local_row(s, k.idx, k.id) < 0) return ps_set_err(PS_MISSING, "%s is not held by this shard", key);
        *len_out = s->emb.D;
    } else if (k.kind == 1 || k.kind == 2) {
        if (!s->wide.W) return ps_set_err(PS_MISSING, "no wide table");
        if (k.kind == 1 && (k.id < 0 || k.id >= s->wide.rows)) return ps_set_err(PS_MISSING, "%s out of range", key);
        *len_out = 1;
    } else {
        if (k.idx < 0 || k.idx >= (int)s->fc.size() || !s->fc[k.idx].present) return ps_set_err(PS_MISSING, "%s absent", key);
        *len_out = k.kind == 4 ? s->fc[k.idx].N : s->fc[k.idx].K * s->fc[k.idx].N;
    }
    return PS_OK;
}

extern "C" int ps_store_push_update(ps_store_t *s, int n, const char *const *keys, const float *const *grads, const int *lens,
                                    int is_async) {
    if (!s || n < 0 || (n > 0 && (!keys || !grads || !lens))) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (n == 0) return PS_OK;
    PSCHK(store_enter(s));
    hipStream_t st = s->stream;
    std::vector<ParsedKey> pk((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!keys[i] || !grads[i]) return ps_set_err(PS_E_BAD_ARG, "message %d: null key or gradient", i);
        if (!store_parse_key(keys[i], &pk[i])) return ps_set_err(PS_MISSING, "unknown key %s", keys[i]);
    }
    // ---- pass 1: validate EVERY message (kind, length, range, local row) and group the indices; no device work yet, so a
    // bad message leaves the store untouched (validation and application used to interleave) ----
    std::vector<uint32_t> rows;                                    // embedding rows: one list in arrival order, duplicates =
    std::vector<float> eg;                                         // several workers' pushes of one key
    std::map<int64_t, std::vector<int>> wide;                      // wide keys: ascending key, message order inside
    std::map<std::pair<int, int>, std::vector<int>> dense;        // (layer, bias) -> messages
    const int D = s->emb.D;
    for (int i = 0; i < n; ++i) {
        if (lens[i] < 0) return ps_set_err(PS_E_BAD_ARG, "message %d: negative length", i);
        if (pk[i].kind == 0) {
            if (!s->emb.W) return ps_set_err(PS_MISSING, "no embedding tables");
            if (lens[i] != D) return ps_set_err(PS_E_BAD_ARG, "%s wants %d floats, got %d", keys[i], D, lens[i]);
            if (pk[i].idx < 0 || pk[i].idx >= s->emb.F) return ps_set_err(PS_MISSING, "%s: no such field", keys[i]);
            const int64_t r = store_local_row(s, pk[i].idx, pk[i].id);
            if (r < 0) return ps_set_err(PS_MISSING, "%s is not held by this shard", keys[i]);
            rows.push_back((uint32_t)r);
        } else if (pk[i].kind == 1 || pk[i].kind == 2) {
            if (!s->wide.W) return ps_set_err(PS_MISSING, "no wide table");
            if (lens[i] != 1) return ps_set_err(PS_E_BAD_ARG, "%s wants 1 float, got %d", keys[i], lens[i]);
            if (pk[i].kind == 1 && (pk[i].id < 0 || pk[i].id >= s->wide.rows)) return ps_set_err(PS_MISSING, "%s out of range", keys[i]);
            wide[pk[i].kind == 2 ? s->wide.rows : pk[i].id].push_back(i);
        } else {
            const int l = pk[i].idx;
            if (l < 0 || l >= (int)s->fc.size() || !s->fc[l].present) return ps_set_err(PS_MISSING, "%s absent", keys[i]);
            const int want = pk[i].kind == 4 ? s->fc[l].N : s->fc[l].K * s->fc[l].N;
            if (lens[i] != want) return ps_set_err(PS_E_BAD_ARG, "%s wants %d floats, got %d", keys[i], want, lens[i]);
            dense[{l, pk[i].kind == 4 ? 1 : 0}].push_back(i);
        }
    }
    {   // (updaters too: PServer.push answers 500 for an unknown updater before anything is summed)
        ps_updater_t u;
        UpdParams eu;
        FieldUpd efu;
        bool stateful = false;
        if (!rows.empty()) PSCHK(store_fill_field_upd(s, &eu, &efu, &stateful));
        if (!rows.empty() && !s->emb.state && stateful) return ps_set_err(PS_MISSING, "no embedding tables with updater state");
        if (!wide.empty()) PSCHK(store_resolve_updater(s, "wide.weights", &u));
        for (auto &kv : dense) {
            char name[64];
            snprintf(name, sizeof name, "fc%d.%s", kv.first.first, kv.first.second ? "bias" : "weights");
            PSCHK(store_resolve_updater(s, name, &u));
        }
    }
    // ---- pass 2: apply ----
    for (int i = 0; i < n; ++i)
        if (pk[i].kind == 0) eg.insert(eg.end(), grads[i], grads[i] + D);
    DevBuf d_rows, d_eg;
    if (!rows.empty()) {
        PSCHK(d_rows.alloc(sizeof(uint32_t) * rows.size()));
        PSCHK(d_eg.alloc(sizeof(float) * eg.size()));
        HIPCHK(hipMemcpyAsync(d_rows.p, rows.data(), sizeof(uint32_t) * rows.size(), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_eg.p, eg.data(), sizeof(float) * eg.size(), hipMemcpyHostToDevice, st));
        PSCHK(shard_apply_push(s, (const uint32_t *)d_rows.p, (const float *)d_eg.p, (int64_t)rows.size(), nullptr, 0, is_async, false));
    }
    // ---- wide keys: CSR of pushes per key ----
    DevBuf d_wid, d_woff, d_wg;
    if (!wide.empty()) {
        std::vector<int64_t> ids;
        std::vector<uint32_t> off(1, 0u);
        std::vector<float> g;
        for (auto &kv : wide) {
            ids.push_back(kv.first);
            for (int i : kv.second) g.push_back(grads[i][0]);
            off.push_back((uint32_t)g.size());
        }
        PSCHK(d_wid.alloc(sizeof(int64_t) * ids.size()));
        PSCHK(d_woff.alloc(sizeof(uint32_t) * off.size()));
        PSCHK(d_wg.alloc(sizeof(float) * g.size()));
        HIPCHK(hipMemcpyAsync(d_wid.p, ids.data(), sizeof(int64_t) * ids.size(), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_woff.p, off.data(), sizeof(uint32_t) * off.size(), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_wg.p, g.data(), sizeof(float) * g.size(), hipMemcpyHostToDevice, st));
        ps_updater_t u;
        PSCHK(store_resolve_updater(s, "wide.weights", &u));
        WideListArgs a;
        memset(&a, 0, sizeof a);
        a.rows = s->wide.rows; a.nkeys = (int)ids.size(); a.is_async = is_async ? 1 : 0;
        a.key_ids = (const int64_t *)d_wid.p; a.key_off = (const uint32_t *)d_woff.p; a.grads = (const float *)d_wg.p;
        a.W = s->wide.W; a.state = s->wide.state; a.bias = s->wide.bias; a.bias_state = s->wide.bias_state;
        a.upd = make_upd_params(u);
        PSCHK(launch_wide_list(a, st));
    }
    // ---- dense tensors: the pushes of one tensor are the slabs of its update ----
    std::vector<DevBuf> slabs(dense.size());
    size_t di = 0;
    for (auto &kv : dense) {
        const int l = kv.first.first, bias = kv.first.second;
        FcParams &p = s->fc[l];
        const size_t len = bias ? (size_t)p.N : (size_t)p.K * p.N;
        const int m = (int)kv.second.size();
        DevBuf &buf = slabs[di++];
        PSCHK(buf.alloc(sizeof(float) * len * m));
        for (int j = 0; j < m; ++j)
            HIPCHK(hipMemcpyAsync((float *)buf.p + (size_t)j * len, grads[kv.second[j]], sizeof(float) * len, hipMemcpyHostToDevice, st));
        ps_updater_t u;
        char name[64];
        snprintf(name, sizeof name, "fc%d.%s", l, bias ? "bias" : "weights");
        PSCHK(store_resolve_updater(s, name, &u));
        const int launches = is_async ? m : 1;
        for (int j = 0; j < launches; ++j) {
            DenseUpdArgs d;
            memset(&d, 0, sizeof d);
            d.nlayers = 1; d.apply = 1; d.upd = make_upd_params(u);
            d.B = is_async ? 1 : m;                                    // KVStore.update: divi(sumCnt)
            DenseLayer &L = d.L[0];
            L.W = p.W; L.Wt = p.Wt; L.Wp = p.Wp; L.S1 = p.S1; L.S2 = p.S2;
            L.K = p.K; L.N = p.N; L.ldw = p.ldw; L.ldwt = p.Kpad; L.ldp = p.N;
            L.part_stride = (int64_t)len; L.nsplit = is_async ? 1 : m;
            // a slab holds rows [row_lo, row_lo + row_cnt) only: shift the base so that row k lands on it
            L.row_lo = bias ? p.K : 0; L.row_cnt = bias ? 1 : p.K;
            L.part = (const float *)buf.p + (size_t)(is_async ? j : 0) * len - (size_t)L.row_lo * L.ldp;
            L.elem_begin = 0; L.elem_end = (int64_t)(p.K + 1) * p.N;
            PSCHK(launch_dense_update(d, st));
        }
    }
    HIPCHK(hipStreamSynchronize(st));          // the scratch buffers die with this call
    return PS_OK;
}

// PServer.globalStep (net/PServer.java:40): psUpdate / an async barrier() add one
extern "C" int ps_store_advance_global_step(ps_store_t *s, int64_t by) {
    if (!s) return ps_set_err(PS_E_BAD_ARG, "null store");
    s->global_step += by;
    return PS_OK;
}
