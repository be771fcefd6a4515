// ps_layer_ops.hip -- the layer-granular BACKWARD operators of the C ABI: what a Java `GpuFcLayer extends Layer` /
// `GpuEmbeddingLayer extends Layer` binds for Layer.backward() (layer/Layer.java:39-45), plus the KVStore.update of
// the gradients they leave behind.  They run the same kernels as the whole-model step (ps_model.hip); only the
// buffers are the caller's and the gradients wait in the store between backward and update, as KVStore.sum keeps
// them between Layer.backward and Trainer's kvStore.update (train/Trainer.java:90-100).
//
//   FcLayer.backward          layer/FcLayer.java:93-110        -> ps_fc_backward
//   KVStore.sum / update      store/KVStore.java:192-200,240-277 -> pending gradient + ps_dense_update
//   EmbeddingLayer.backward   layer/EmbeddingLayer.java:59-69,
//   EmbeddingField.backward   layer/EmbeddingField.java:86-104 (twice) -> ps_emb_backward_update
#include <string.h>

#include "ps_store.h"

namespace {

int bits_for(int64_t n) {
    int b = 1;
    while (b < 32 && (1ll << b) < n) ++b;
    return b;
}

template <typename T>
int grow(ps_store *s, T **p, int64_t *cap, int64_t need) {
    if (need <= *cap) return PS_OK;
    RtGuard rt_guard;
    HIPCHK(hipStreamSynchronize(s->stream));
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    HIPCHK(hipMalloc((void **)p, sizeof(T) * (size_t)need));
    *cap = need;
    return PS_OK;
}

// delta *= act'(y) in place (activations/Relu.java:14-19, activations/Sigmoid.java:16-21)
__global__ __launch_bounds__(256) void k_act_backward(float *__restrict__ d, int ldd, const float *__restrict__ y, int ldy, int B, int N, int act) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)B * N) return;
    const int b = (int)(t / N), n = (int)(t % N);
    const float v = y[(size_t)b * ldy + n];
    float &x = d[(size_t)b * ldd + n];
    if (act == PS_ACT_RELU) x = x * (v > 0.f ? 1.f : 0.f);
    else if (act == PS_ACT_SIGMOID) x = x * (v * (1.f - v));
}

// pending[t] (+)= (sum of the split slabs in slab order) / B        KVStore.sum: put on first touch, addi after
__global__ __launch_bounds__(256) void k_pending_accum(float *__restrict__ pending, const float *__restrict__ part, int64_t part_stride, int ldp,
                                                       int nsplit, int K1, int N, int B, int first) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)K1 * N) return;
    const int k = (int)(t / N), n = (int)(t % N);
    float sacc = 0.f;
    for (int z = 0; z < nsplit; ++z) sacc += part[(size_t)z * part_stride + (size_t)k * ldp + n];
    const float g = sacc / (float)B;                    // divi(delta.columns) FcLayer.java:105 / rowMeans :103
    pending[t] = first ? g : g + pending[t];
}

// masked[b][c] = delta[b][c] * relu'(a[b][c]) for the F*D embedding columns (EmbeddingField.java:91)
__global__ __launch_bounds__(256) void k_emb_mask(float *__restrict__ out, int ldo, const float *__restrict__ delta, int ldd,
                                                  const float *__restrict__ a, int lda, int B, int C, int act) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)B * C) return;
    const int b = (int)(t / C), c = (int)(t % C);
    float v = delta[(size_t)b * ldd + c];
    if (act == PS_ACT_RELU) v = v * (a[(size_t)b * lda + c] > 0.f ? 1.f : 0.f);
    out[(size_t)b * ldo + c] = v;
}

// sort key (global row) of every entry, and its bag
__global__ __launch_bounds__(256) void k_emb_keys(const int64_t *__restrict__ ids, const int64_t *__restrict__ offsets, int64_t nbags, int F,
                                                  const int64_t *__restrict__ row_base, uint32_t *__restrict__ keys, uint32_t *__restrict__ ent_bag, int *err) {
    const int64_t bag = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (bag >= nbags) return;
    const int f = (int)(bag % F);
    const int64_t rb = row_base[f], rn = row_base[f + 1] - rb;
    const int64_t p0 = offsets ? offsets[bag] : bag, p1 = offsets ? offsets[bag + 1] : bag + 1;
    for (int64_t p = p0; p < p1; ++p) {
        int64_t id = ids[p];
        if (id < 0 || id >= rn) { atomicAdd(err, 1); id = 0; }
        keys[p] = (uint32_t)(rb + id);
        ent_bag[p] = (uint32_t)bag;
    }
}

}  // namespace

extern "C" int ps_fc_backward(ps_store_t *s, int layer, int act, const float *x_dev, int ldx, const float *y_dev, int ldy,
                              float *delta_dev, int ldd, int B, float *dx_dev, int lddx) {
    RoctxRange roctx_range("ps_fc_backward");
    if (!s || !x_dev || !delta_dev || B <= 0) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    if (layer < 0 || layer >= (int)s->fc.size() || !s->fc[layer].present) return ps_set_err(PS_MISSING, "fc%d absent", layer);
    FcParams &p = s->fc[layer];
    if (ldx != p.Kpad) return ps_set_err(PS_E_BAD_ARG, "fc%d wants ldx = %d (in+1 rounded up to 16, ones column at %d)", layer, p.Kpad, p.K);
    if (ldd != p.ldw) return ps_set_err(PS_E_BAD_ARG, "fc%d wants ldd = %d (out rounded up to 16, padding columns zero)", layer, p.ldw);
    if (act != PS_ACT_NONE && (!y_dev || ldy < p.N)) return ps_set_err(PS_E_BAD_ARG, "act' needs the layer's output y_dev");
    if (dx_dev && (lddx < p.K || (lddx & 3))) return ps_set_err(PS_E_BAD_ARG, "bad lddx %d", lddx);
    PSCHK(store_enter(s));
    hipStream_t st = s->stream;
    // activation.backward in place (:100-102)
    if (act != PS_ACT_NONE) {
        hipLaunchKernelGGL(k_act_backward, dim3(cdiv((int64_t)B * p.N, 256)), dim3(256), 0, st, delta_dev, ldd, y_dev, ldy, B, p.N, act);
        HIPCHK(hipGetLastError());
    }
    // weightsGradient / biasGradient (:103-106) -> KVStore.sum: the bias rides in row K through the ones column
    const int nsplit = gemm_tn_choose_split(p.K + 1, p.N, B);
    const int64_t part_stride = (int64_t)p.Kpad * p.ldw;
    PSCHK(grow(s, &s->ops.part, &s->ops.part_cap, part_stride * nsplit));
    PSCHK(gemm_tn_splitk(x_dev, ldx, ldx, delta_dev, ldd, ldd, s->ops.part, p.ldw, part_stride, p.K + 1, p.N, B, nsplit, nullptr, st));
    if (!p.pending) PSCHK(store_dev_alloc(s, (void **)&p.pending, sizeof(float) * (size_t)(p.K + 1) * p.N, true));
    hipLaunchKernelGGL(k_pending_accum, dim3(cdiv((int64_t)(p.K + 1) * p.N, 256)), dim3(256), 0, st, p.pending, s->ops.part, part_stride, p.ldw,
                       nsplit, p.K + 1, p.N, B, p.pending_cnt == 0 ? 1 : 0);
    HIPCHK(hipGetLastError());
    p.pending_cnt++;
    // this.delta = weights^T * delta (:108): what the layer below reads as next.delta
    if (dx_dev)
        PSCHK(gemm_nt(delta_dev, ldd, B, p.W, p.ldw, p.K, dx_dev, lddx, B, p.K, ldd, EPI_NONE, nullptr, 0, 0, nullptr, st));
    return PS_OK;
}

extern "C" int ps_fc_pending_grad(ps_store_t *s, int layer, int bias, float *out, int cap, int *count) {
    if (!s || !out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    if (layer < 0 || layer >= (int)s->fc.size() || !s->fc[layer].present) return ps_set_err(PS_MISSING, "fc%d absent", layer);
    FcParams &p = s->fc[layer];
    if (count) *count = p.pending_cnt;
    if (!p.pending || p.pending_cnt == 0) return ps_set_err(PS_MISSING, "fc%d has no pending gradient", layer);
    const int n = bias ? p.N : p.K * p.N;
    if (cap < n) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
    PSCHK(store_enter(s));
    HIPCHK(hipMemcpyAsync(out, p.pending + (bias ? (size_t)p.K * p.N : 0), sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}

extern "C" int ps_dense_update(ps_store_t *s, int layer) {
    RoctxRange roctx_range("ps_dense_update");
    if (!s) return ps_set_err(PS_E_BAD_ARG, "store is NULL");
    PSCHK(store_enter(s));
    int done = 0;
    for (int l = 0; l < (int)s->fc.size(); ++l) {
        if (layer >= 0 && l != layer) continue;
        FcParams &p = s->fc[l];
        if (!p.present || !p.pending || p.pending_cnt == 0) continue;
        // KVStore.update(Map): g = sum / sumCnt, updater by key (exact, prefix, "default"), then clear (:240-277)
        char key[32];
        snprintf(key, sizeof key, "fc%d.weights", l);
        ps_updater_t u;
        PSCHK(store_resolve_updater(s, key, &u));
        DenseUpdArgs d;
        memset(&d, 0, sizeof d);
        d.nlayers = 1; d.B = 1; d.apply = 1; d.flat_grad = p.pending; d.flat_div = (float)p.pending_cnt;
        d.upd = make_upd_params(u);
        DenseLayer &L = d.L[0];
        L.W = p.W; L.Wt = p.Wt; L.Wp = p.Wp; L.S1 = p.S1; L.S2 = p.S2; L.K = p.K; L.N = p.N; L.ldw = p.ldw; L.ldwt = p.Kpad;
        L.elem_begin = 0; L.elem_end = (int64_t)(p.K + 1) * p.N;
        PSCHK(launch_dense_update(d, s->stream));
        p.pending_cnt = 0;
        ++done;
    }
    if (layer >= 0 && !done) return ps_set_err(PS_MISSING, "fc%d has no pending gradient", layer);
    if (done) s->global_step++;
    return PS_OK;
}

extern "C" int ps_emb_backward_update(ps_store_t *s, const int64_t *ids_dev, const int64_t *offsets_dev, int64_t nnz, int B, int act,
                                      const float *a_dev, int lda, const float *delta_dev, int ldd, int grad_mode, int sum_order, int apply) {
    RoctxRange roctx_range("ps_emb_backward_update");
    if (!s || !ids_dev || !delta_dev || B <= 0 || nnz < 0) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    EmbTables &e = s->emb;
    if (!e.W) return ps_set_err(PS_MISSING, "no embedding tables");
    const int F = e.F, D = e.D, C = F * D;
    if (ldd < C || (act == PS_ACT_RELU && (!a_dev || lda < C))) return ps_set_err(PS_E_BAD_ARG, "bad leading dimension / missing layer output");
    if (!offsets_dev && nnz != (int64_t)B * F) return ps_set_err(PS_E_BAD_ARG, "single-hot: nnz must be B*F");
    if (grad_mode != PS_GRAD_COMPAT && grad_mode != PS_GRAD_INTENDED) return ps_set_err(PS_E_BAD_ARG, "bad grad_mode");
    if (sum_order < PS_SUM_AUTO || sum_order > PS_SUM_CHUNKED) return ps_set_err(PS_E_BAD_ARG, "bad sum_order");
    if (e.nshards > 1) return ps_set_err(PS_E_UNSUPPORTED, "a sharded store takes its gradients through ps_shard_apply_push");
    UpdParams emb_upd;
    FieldUpd emb_fu;
    bool stateful = false;
    PSCHK(store_fill_field_upd(s, &emb_upd, &emb_fu, &stateful));
    if (apply && !e.state && stateful)
        return ps_set_err(PS_E_STATE, "the embedding table was created weights-only (state_slots = 0): Adam / Ftrl cannot train it");
    if (nnz == 0) return PS_OK;
    PSCHK(store_enter(s));
    hipStream_t st = s->stream;
    ps_store::OpScratch &o = s->ops;
    const int ldm = (int)round_up(C, 16);
    if (nnz > o.nnz_cap) {
        RtGuard rt_guard;
        HIPCHK(hipStreamSynchronize(st));
        auto fr = [](void *p) { if (p) (void)hipFree(p); };
        sort_ws_free(o.ws);
        fr(o.keys); fr(o.ents); fr(o.ent_bag); fr(o.seg_start); fr(o.seg_id); fr(o.nseg); fr(o.uniq_row); fr(o.partials); fr(o.partials2); fr(o.grads);
        // (if an allocation below fails the scratch is empty, not dangling: the next call starts over, destroy frees nothing twice)
        o.keys = o.ents = o.ent_bag = o.seg_start = o.seg_id = o.nseg = o.uniq_row = nullptr;
        o.partials = o.partials2 = o.grads = nullptr;
        o.nnz_cap = 0;
        const int64_t cap = nnz + nnz / 4 + 1024;
        PSCHK(sort_ws_alloc(o.ws, cap));
        HIPCHK(hipMalloc((void **)&o.keys, 4 * (size_t)(cap + 1))); HIPCHK(hipMalloc((void **)&o.ents, 4 * (size_t)(cap + 1)));
        HIPCHK(hipMalloc((void **)&o.ent_bag, 4 * (size_t)(cap + 1))); HIPCHK(hipMalloc((void **)&o.seg_start, 4 * (size_t)(cap + 2)));
        HIPCHK(hipMalloc((void **)&o.seg_id, 4 * (size_t)(cap + 1))); HIPCHK(hipMalloc((void **)&o.nseg, 16));
        HIPCHK(hipMalloc((void **)&o.uniq_row, 4 * (size_t)(cap + 1)));
        const size_t np = 2 * (size_t)((cap + PS_EMB_CHUNK - 1) / PS_EMB_CHUNK + 1) * D;
        HIPCHK(hipMalloc((void **)&o.partials, 4 * np)); HIPCHK(hipMalloc((void **)&o.partials2, 4 * np));
        HIPCHK(hipMalloc((void **)&o.grads, 4 * (size_t)(cap + 1) * D));
        o.nnz_cap = cap;
    }
    PSCHK(grow(s, &o.masked, &o.masked_cap, (int64_t)B * ldm));
    // g_k = relu'(A) .* delta slice (EmbeddingField.java:91), keys, stable sort by row, runs of equal rows
    hipLaunchKernelGGL(k_emb_mask, dim3(cdiv((int64_t)B * C, 256)), dim3(256), 0, st, o.masked, ldm, delta_dev, ldd, a_dev, lda, B, C, act);
    const int64_t nbags = (int64_t)B * F;
    hipLaunchKernelGGL(k_emb_keys, dim3(cdiv(nbags, 256)), dim3(256), 0, st, ids_dev, offsets_dev, nbags, F, e.row_base_dev, o.keys, o.ent_bag, s->err_dev);
    HIPCHK(hipGetLastError());
    uint32_t *sk = nullptr, *se = nullptr;
    PSCHK(radix_sort_pairs(o.ws, o.keys, o.ents, nnz, bits_for(e.total_rows), true, &sk, &se, st));
    PSCHK(build_segments(o.ws, sk, nnz, o.seg_start, o.seg_id, o.nseg, st));
    EmbBwdArgs g;
    memset(&g, 0, sizeof g);
    g.nnz = nnz; g.F = F; g.D = D; g.grad_mode = grad_mode; g.apply = apply ? 1 : 0;
    g.sorted_key = sk; g.sorted_ent = se; g.seg_start = o.seg_start; g.seg_id = o.seg_id; g.nseg = o.nseg;
    g.ent_bag = offsets_dev ? o.ent_bag : nullptr;        // single-hot: entry == bag
    g.delta = o.masked; g.ldd = ldm; g.partials = o.partials; g.partials2 = o.partials2; g.W = e.W; g.state = e.state;
    g.long_runs = 1;
    g.seq_order = (sum_order == PS_SUM_SEQUENTIAL || (sum_order == PS_SUM_AUTO && !offsets_dev)) ? 1 : 0;
    g.upd = emb_upd; g.fu = emb_fu;
    g.grads_out = o.grads; g.uniq_row = o.uniq_row; g.uniq_cnt = nullptr; g.skip = nullptr;
    PSCHK(launch_emb_bwd(g, st));
    o.last_nnz = nnz;
    if (apply) s->global_step++;
    return PS_OK;
}

// the per-key gradients of the last ps_emb_backward_update (what KVStore.sum holds after /cnt): unique local rows + [n][D]
extern "C" int ps_emb_last_grads(ps_store_t *s, int64_t *rows_out, float *grads_out, int64_t cap_rows, int64_t *n_out) {
    if (!s || !n_out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ps_store::OpScratch &o = s->ops;
    if (!o.nseg || o.last_nnz <= 0) return ps_set_err(PS_MISSING, "no ps_emb_backward_update yet");
    PSCHK(store_enter(s));
    uint32_t nseg = 0;
    HIPCHK(hipMemcpyAsync(&nseg, o.nseg, 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    *n_out = nseg;
    if (!rows_out || !grads_out) return PS_OK;
    if (cap_rows < (int64_t)nseg) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
    std::vector<uint32_t> rows(nseg);
    if (nseg) {
        HIPCHK(hipMemcpyAsync(rows.data(), o.uniq_row, 4 * (size_t)nseg, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(grads_out, o.grads, 4 * (size_t)nseg * s->emb.D, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    for (uint32_t i = 0; i < nseg; ++i) rows_out[i] = rows[i];
    return PS_OK;
}
