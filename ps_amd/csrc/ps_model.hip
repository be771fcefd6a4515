// ps_model.hip -- the layer graph of model/DNN.java / model/WideDeepNN.java
// laid out for one MI355X: activations [B][features] with every FcLayer's
// bias folded into its GEMM through a ones column, the whole step issued on
// one HIP stream (optionally replayed as a hipGraph).
//
// Step = TrainerThread.call (train/TrainerThread.java:29-39) + the tail of
// Trainer.train (train/Trainer.java:90-100) for thread = 1:
//   forward : EmbeddingLayer -> ConcatLayer -> FcLayer x nfc [-> LRLayer -> AddLayer] -> CrossEntropy
//   backward: CrossEntropy' -> [AddLayer', LRLayer'] -> FcLayer' x nfc -> EmbeddingLayer' (twice, App. A.6)
//   update  : KVStore.update(updaters) (Adam / Ftrl fused into the reducers) ; clear
#include <string.h>

#include "ps_store.h"

namespace {

int model_alloc(ps_model *m, void **p, size_t bytes, bool zero) { return store_dev_alloc(m->s, p, bytes, zero); }

// event bracket around one kernel group when profiling is on
struct Prof {
    ps_model *m;
    long idx = -1;
    Prof(ps_model *mm, const char *name) : m(mm) {
        if (!m->profile) return;
        if (!m->prof_filter.empty()) {          // a comma-separated list of group names
            const std::string f = "," + m->prof_filter + ",", n = std::string(",") + name + ",";
            if (f.find(n) == std::string::npos) return;
        }
        ps_model::ProfEvent e;
        e.name = name;
        (void)hipEventCreate(&e.a);
        (void)hipEventCreate(&e.b);
        (void)hipEventRecord(e.a, m->s->stream);
        m->prof_events.push_back(e);
        idx = (long)m->prof_events.size() - 1;
    }
    ~Prof() {
        if (idx >= 0) (void)hipEventRecord(m->prof_events[idx].b, m->s->stream);
    }
};

// side chains: while profiling (events on the main stream) or with multi_stream off, everything
// stays on the main stream and fork/join are no-ops
hipStream_t side_stream(ps_model *m, int i) { return (m->profile || !m->multi_stream) ? m->s->stream : m->side[i]; }
int fork(ps_model *m, hipStream_t from, hipStream_t to) {
    if (from == to) return PS_OK;
    hipEvent_t e = m->events[m->next_event++ % m->events.size()];
    HIPCHK(hipEventRecord(e, from));
    HIPCHK(hipStreamWaitEvent(to, e, 0));
    return PS_OK;
}
int join(ps_model *m, hipStream_t from, hipStream_t to) { return fork(m, from, to); }
// "this launch on the main stream carries an event": the event goes into the launch's LaunchOpts; settle_event()
// afterwards: if the launcher took it, the event is the kernel's own completion signal; if not (nothing was launched, or
// a path that does not carry events), it is recorded the ordinary way.
hipEvent_t pick_event(ps_model *m) {
    if (m->profile || !m->multi_stream || !g_ext_events || m->cfg.use_graph) return nullptr;      // (stream capture: plain records)
    return m->events[m->next_event++ % m->events.size()];
}
int settle_event(ps_model *m, const LaunchOpts &lo) {
    if (lo.stop_event && !lo.launched) HIPCHK(hipEventRecord(lo.stop_event, m->s->stream));
    return PS_OK;
}
int wait_event(ps_model *m, hipStream_t to, hipEvent_t e) {
    if (e && to != m->s->stream) HIPCHK(hipStreamWaitEvent(to, e, 0));
    return PS_OK;
}
// one record on `from`, two waiters: an event record costs the recording stream ~5 us before its next kernel starts
// (tools/gpu_timeline.py: 3 records between the head and the first delta GEMM were a 20 us hole in the main chain)
int fork2(ps_model *m, hipStream_t from, hipStream_t to_a, hipStream_t to_b) {
    if (from == to_a && from == to_b) return PS_OK;
    hipEvent_t e = m->events[m->next_event++ % m->events.size()];
    HIPCHK(hipEventRecord(e, from));
    if (to_a != from) HIPCHK(hipStreamWaitEvent(to_a, e, 0));
    if (to_b != from && to_b != to_a) HIPCHK(hipStreamWaitEvent(to_b, e, 0));
    return PS_OK;
}

int bits_for(int64_t n) {
    int b = 1;
    while (b < 32 && (1ll << b) < n) ++b;
    return b;
}

}  // namespace

extern "C" int ps_model_create(ps_store_t *s, const ps_model_config_t *cfg, ps_model_t **out) {
    RtGuard rt_guard;
    if (!s || !cfg || !out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    *out = nullptr;
    if (cfg->F <= 0 || cfg->D <= 0 || cfg->X < 0 || cfg->nfc <= 0 || cfg->nfc > 8 || cfg->max_batch <= 0)
        return ps_set_err(PS_E_BAD_ARG, "bad model config");
    if (cfg->emb_sum_order < PS_SUM_AUTO || cfg->emb_sum_order > PS_SUM_CHUNKED) return ps_set_err(PS_E_BAD_ARG, "bad emb_sum_order %d", cfg->emb_sum_order);
    if (cfg->fc_dims[cfg->nfc - 1] != 1) return ps_set_err(PS_E_UNSUPPORTED, "the last FcLayer must have 1 output (CrossEntropy is binary)");
    if (!s->emb.W) return ps_set_err(PS_E_STATE, "create the embedding tables first (ps_store_create_embedding)");
    if (s->emb.F != cfg->F || s->emb.D != cfg->D) return ps_set_err(PS_E_BAD_ARG, "store embedding is %dx%d, model wants %dx%d", s->emb.F, s->emb.D, cfg->F, cfg->D);
    PSCHK(store_enter(s));
    if (cfg->kind == PS_MODEL_WIDEDEEP) {
        if (cfg->wide_size <= 0) return ps_set_err(PS_E_BAD_ARG, "WideDeep needs wide_size");
        if (!s->wide.W) PSCHK(ps_store_create_wide(s, cfg->wide_size));
        if (s->wide.rows != cfg->wide_size) return ps_set_err(PS_E_BAD_ARG, "wide table size mismatch");
    }
    ps_model *m = new ps_model();
    m->s = s;
    m->cfg = *cfg;
    m->Bcap = cfg->max_batch;
    m->nnz_cap = cfg->max_nnz > 0 ? cfg->max_nnz : (int64_t)cfg->max_batch * cfg->F;
    const int F = cfg->F, D = cfg->D, X = cfg->X, nfc = cfg->nfc, B = m->Bcap;
    // FcLayer.build (layer/FcLayer.java:53-70)
    int in = F * D + X;
    for (int l = 0; l < nfc; ++l) {
        PSCHK(ps_store_create_fc(s, l, in, cfg->fc_dims[l]));
        in = cfg->fc_dims[l];
    }
    m->fc.resize(nfc);
    m->dense_elems = 0;
    for (int l = 0; l < nfc; ++l) {
        FcParams &p = s->fc[l];
        FcBuf &b = m->fc[l];
        b.ldA = p.Kpad;
        PSCHK(model_alloc(m, (void **)&b.A, sizeof(float) * (size_t)B * b.ldA, true));
        PSCHK(launch_fill_col(b.A, B, b.ldA, p.K, 1.0f, s->stream));       // the ones column that carries the bias
        b.ldD = p.ldw;
        PSCHK(model_alloc(m, (void **)&b.dOut, sizeof(float) * (size_t)B * b.ldD, true));
        b.nsplit = gemm_tn_choose_split(p.K + 1, p.N, B);
        if (l == nfc - 1 && p.N == 1) {              // k_last_bwd: one partial slab per workgroup (16 batch rows; 32 and up beyond 256 workgroups)
            b.nsplit = cdiv(B, g_last_rows > 0 ? g_last_rows : 16);   // 16 rows per workgroup (256 workgroups at B = 4096; measured 0.1591 vs 0.1600 ms/step at 32)
            if (b.nsplit > 256) b.nsplit = 256;
        }
        b.ldp = p.ldw;
        b.part_stride = (int64_t)p.Kpad * b.ldp;
        PSCHK(model_alloc(m, (void **)&b.part, sizeof(float) * (size_t)b.nsplit * b.part_stride, true));
        m->dense_elems += (int64_t)(p.K + 1) * p.N;
    }
    m->ld_last = (int)round_up(cfg->fc_dims[nfc - 1] + 1, 16);
    PSCHK(model_alloc(m, (void **)&m->out_last, sizeof(float) * (size_t)B * m->ld_last, true));
    m->ldx = (int)round_up(F * D, 16);
    PSCHK(model_alloc(m, (void **)&m->dx, sizeof(float) * (size_t)B * m->ldx, true));
    PSCHK(model_alloc(m, (void **)&m->P, sizeof(float) * B, true));
    PSCHK(model_alloc(m, (void **)&m->wide_z, sizeof(float) * B, true));
    PSCHK(model_alloc(m, (void **)&m->terms, sizeof(float) * B, true));
    PSCHK(model_alloc(m, (void **)&m->loss_dev, sizeof(float) * 4, true));
    PSCHK(model_alloc(m, (void **)&m->gbar_dev, sizeof(float) * 4, true));
    PSCHK(model_alloc(m, (void **)&m->skip_dev, sizeof(int) * 4, true));
    PSCHK(model_alloc(m, (void **)&m->ids_dev, sizeof(int64_t) * (size_t)m->nnz_cap, false));
    PSCHK(model_alloc(m, (void **)&m->offsets_dev, sizeof(int64_t) * ((size_t)B * F + 1), false));
    PSCHK(model_alloc(m, (void **)&m->wide_ids_dev, sizeof(int64_t) * (size_t)B * F, false));
    PSCHK(model_alloc(m, (void **)&m->dense_dev, sizeof(float) * (size_t)B * (X > 0 ? X : 1), false));
    PSCHK(model_alloc(m, (void **)&m->labels_dev, sizeof(float) * B, false));
    const int64_t nc = m->nnz_cap;
    PSCHK(sort_ws_alloc(m->ws, nc));
    PSCHK(model_alloc(m, (void **)&m->keys, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(model_alloc(m, (void **)&m->ents, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(model_alloc(m, (void **)&m->ent_bag, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(model_alloc(m, (void **)&m->seg_start, sizeof(uint32_t) * (size_t)(nc + 2), false));
    PSCHK(model_alloc(m, (void **)&m->seg_id, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(model_alloc(m, (void **)&m->nseg_dev, sizeof(uint32_t) * 8, true));
    m->nseg_cur = m->nseg_dev;
    PSCHK(model_alloc(m, (void **)&m->seg_nseg_scratch, sizeof(uint32_t) * 4, true));
    PSCHK(model_alloc(m, (void **)&m->fs_keys, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(model_alloc(m, (void **)&m->fs_ents, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(model_alloc(m, (void **)&m->long_list, sizeof(uint32_t) * 3 * (size_t)(nc / (PS_EMB_SEQ_TILE + 1) + 2), false));
    // (behind the F look-back words: the field table [F][rbase, runs, lbase, long runs] the sort leaves for the embedding update -- PS_FS_TAB_OFF)
    PSCHK(model_alloc(m, (void **)&m->fs_pub, sizeof(unsigned long long) * (size_t)PS_FS_TAB_OFF(F) + 16 * (size_t)F, true));
    PSCHK(model_alloc(m, (void **)&m->start_flag, sizeof(unsigned int) * 16, true));
    PSCHK(model_alloc(m, (void **)&m->pair_ctr, sizeof(unsigned int) * (size_t)(cdiv(B, 64) + 8), true));       // k_fc_fwd_pair: tiles done per row panel      // [0..7] flags, [8] the dense update's workgroup count
    PSCHK(model_alloc(m, (void **)&m->uniq_row, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(model_alloc(m, (void **)&m->uniq_cnt, sizeof(uint32_t) * (size_t)(nc + 1), false));
    PSCHK(model_alloc(m, (void **)&m->partials, sizeof(float) * 2 * (size_t)((nc + PS_EMB_CHUNK - 1) / PS_EMB_CHUNK + 1) * D, false));
    PSCHK(model_alloc(m, (void **)&m->partials2, sizeof(float) * 2 * (size_t)((nc + PS_EMB_CHUNK - 1) / PS_EMB_CHUNK + 1) * D, false));
    PSCHK(model_alloc(m, (void **)&m->grads_out, sizeof(float) * (size_t)(nc + 1) * D, false));
    PSCHK(model_alloc(m, (void **)&m->dense_grad_flat, sizeof(float) * (size_t)m->dense_elems, true));
    // the side chains (sort, dW + dense update, the sharded step's list chain) yield to the main FC chain, which is the critical
    // path: least urgent streams, from the process-wide pool (ps_store.h pool_stream_acquire)
    for (int i = 0; i < 3; ++i) PSCHK(pool_stream_acquire(s->device, 0, &m->side[i]));
    m->events.resize(64);
    for (auto &e : m->events) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->loss_ev, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->s0_ev, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->dw_ev, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&m->tail_ev, hipEventDisableTiming));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->device >= 0 && s->device < PS_MAX_DEVICES) { ++g_models_on_device[s->device]; m->counted = true; }      // (dev_waits_ok)
    *out = m;
    return PS_OK;
}

extern "C" int ps_model_destroy(ps_model_t *m) {
    RtGuard rt_guard;
    if (!m) return PS_OK;
    if (m->counted) --g_models_on_device[m->s->device];
    (void)hipSetDevice(m->s->device);
    (void)store_settle(m->s);            // (its flag lives in this model's memory)
    (void)hipStreamSynchronize(m->s->stream);
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
    for (int i = 0; i < 3; ++i) pool_stream_release(m->s->device, 0, m->side[i]);
    for (auto &e : m->events) (void)hipEventDestroy(e);
    if (m->loss_ev) (void)hipEventDestroy(m->loss_ev);
    if (m->s0_ev) (void)hipEventDestroy(m->s0_ev);
    if (m->dw_ev) (void)hipEventDestroy(m->dw_ev);
    if (m->tail_ev) (void)hipEventDestroy(m->tail_ev);
    if (m->hstage.copy_stream) {
        (void)hipStreamSynchronize(m->hstage.copy_stream);
        for (int k = 0; k < 2; ++k) {
            if (m->hstage.pin[k]) (void)hipHostFree(m->hstage.pin[k]);
            if (m->hstage.dev[k]) (void)hipFree(m->hstage.dev[k]);
            if (m->hstage.copied[k]) (void)hipEventDestroy(m->hstage.copied[k]);
            if (m->hstage.done[k]) (void)hipEventDestroy(m->hstage.done[k]);
        }
        pool_stream_release(m->s->device, 2, m->hstage.copy_stream);
    }
    if (m->sh.plan_ev) (void)hipEventDestroy(m->sh.plan_ev);
    if (m->sh.owner_start_host) (void)hipHostFree(m->sh.owner_start_host);
    if (m->sh.counts_host) (void)hipHostFree(m->sh.counts_host);
    if (m->sh.flat_ev) (void)hipEventDestroy(m->sh.flat_ev);
    for (auto &e : m->sh.coll_ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int k = 0; k < 2; ++k) {
        if (m->sh.has_full) { fr(m->sh.x_send_full[k]); fr(m->sh.x_recv_full[k]); }
        fr(m->sh.x_send_blk[k]); fr(m->sh.x_recv_blk[k]);
    }
    if (m->sh.x_ev) (void)hipEventDestroy(m->sh.x_ev);
    if (m->sh.done_ev) (void)hipEventDestroy(m->sh.done_ev);
    shard_mapped_release(m);        // (before the buffers the peers map are freed)
    fr(m->sh.x_recv_rows); fr(m->sh.x_rows_out); fr(m->sh.x_recv_grads); fr(m->sh.x_cache);
    for (auto &b : m->fc) { fr(b.A); fr(b.dOut); fr(b.part); }
    fr(m->out_last); fr(m->dx); fr(m->P); fr(m->wide_z); fr(m->terms); fr(m->loss_dev); fr(m->gbar_dev); fr(m->skip_dev);
    fr(m->ids_dev); fr(m->offsets_dev); fr(m->wide_ids_dev); fr(m->dense_dev); fr(m->labels_dev);
    sort_ws_free(m->ws); sort_ws_free(m->wws); seg_sort_free(m->seg);
    fr(m->wkeys); fr(m->wents); fr(m->wseg_start); fr(m->wseg_id); fr(m->wnseg);
    fr(m->fs_keys); fr(m->fs_ents); fr(m->long_list); fr(m->fs_pub); fr(m->start_flag); fr(m->pair_ctr);
    fr(m->sh.keys2); fr(m->seg_nseg_scratch); fr(m->keys); fr(m->ents); fr(m->ent_bag); fr(m->seg_start); fr(m->seg_id); fr(m->nseg_dev); fr(m->uniq_row); fr(m->uniq_cnt);
    fr(m->partials); fr(m->partials2); fr(m->grads_out); fr(m->dense_grad_flat);
    delete m;
    return PS_OK;
}

// ---------------------------------------------------------------------------
// batch staging
// ---------------------------------------------------------------------------
int stage_batch(ps_model *m, const ps_batch_t *b, bool need_labels) {
    const ps_model_config_t &c = m->cfg;
    if (!b || b->B <= 0 || b->B > m->Bcap) return ps_set_err(PS_E_BAD_ARG, "batch size %d out of (0,%d]", b ? b->B : 0, m->Bcap);
    if (!b->ids) return ps_set_err(PS_E_BAD_ARG, "batch.ids is NULL");
    if (c.X > 0 && !b->dense) return ps_set_err(PS_E_BAD_ARG, "batch.dense is NULL");
    if (need_labels && !b->labels) return ps_set_err(PS_E_BAD_ARG, "batch.labels is NULL");
    if (c.kind == PS_MODEL_WIDEDEEP && !b->wide_ids) return ps_set_err(PS_E_BAD_ARG, "batch.wide_ids is NULL");
    hipStream_t st = m->s->stream;
    m->dev_ok = dev_waits_ok(m->s);      // one decision per step: device-side flags or events (ps_store.h)
    if (m->side0_pending) {      // a forward without its backward: do not let the next gather race the old sort
        PSCHK(join(m, m->side[0], st));
        m->side0_pending = false;
    }
    const int64_t nbags = (int64_t)b->B * c.F;
    int64_t nnz = nbags;
    if (b->offsets) {
        if (b->on_device && b->nnz > 0) {
            nnz = b->nnz;                       // the caller knows it: no read-back, no host wait in front of the step
        } else if (b->on_device) {
            int64_t last = 0;
            HIPCHK(hipMemcpyAsync(&last, b->offsets + nbags, sizeof(int64_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            nnz = last;
        } else {
            nnz = b->offsets[nbags];
        }
    }
    if (nnz < 0 || nnz > m->nnz_cap) return ps_set_err(PS_E_BAD_ARG, "nnz %lld exceeds max_nnz %lld", (long long)nnz, (long long)m->nnz_cap);
    m->cur_B = b->B; m->cur_nnz = nnz;
    m->cur_on_device = b->on_device != 0;
    if (b->on_device) {
        m->cur_ids = b->ids; m->cur_offsets = b->offsets; m->cur_dense = b->dense;
        m->cur_labels = b->labels; m->cur_wide = b->wide_ids;
        return PS_OK;
    }
    // Host buffers (what a JNI caller hands over: Java heap arrays are pageable).  A pageable hipMemcpyAsync is a
    // blocking staged copy in line with the step (measured 0.32 ms/step at configs[1] instead of 0.20), so the
    // batch goes through model-owned PINNED memory and one of two device slots on a copy stream: the memcpy into
    // pinned memory and the DMA of step t overlap the kernels of step t-1; the step's stream waits for the DMA.
    ps_model::HostStage &hs = m->hstage;
    const size_t nb_ids = sizeof(int64_t) * (size_t)nnz, nb_off = b->offsets ? sizeof(int64_t) * (size_t)(nbags + 1) : 0;
    const size_t nb_dense = c.X > 0 ? sizeof(float) * (size_t)b->B * c.X : 0, nb_lab = b->labels ? sizeof(float) * (size_t)b->B : 0;
    const size_t nb_wide = c.kind == PS_MODEL_WIDEDEEP ? sizeof(int64_t) * (size_t)nbags : 0;
    if (!hs.copy_stream) {
        PSCHK(pool_stream_acquire(m->s->device, 2, &hs.copy_stream));
        const size_t cap_ids = sizeof(int64_t) * (size_t)m->nnz_cap, cap_off = sizeof(int64_t) * ((size_t)m->Bcap * c.F + 1);
        const size_t cap_wide = sizeof(int64_t) * (size_t)m->Bcap * c.F, cap_dense = sizeof(float) * (size_t)m->Bcap * (c.X > 0 ? c.X : 1);
        const size_t cap_lab = sizeof(float) * (size_t)m->Bcap;
        hs.off[0] = 0; hs.off[1] = cap_ids; hs.off[2] = hs.off[1] + cap_off; hs.off[3] = hs.off[2] + cap_wide;
        hs.off[4] = hs.off[3] + round_up((int64_t)cap_dense, 16); hs.bytes = hs.off[4] + round_up((int64_t)cap_lab, 16);
        for (int k = 0; k < 2; ++k) {
            HIPCHK(hipHostMalloc((void **)&hs.pin[k], hs.bytes, hipHostMallocDefault));
            HIPCHK(hipMalloc((void **)&hs.dev[k], hs.bytes));
            HIPCHK(hipEventCreateWithFlags(&hs.copied[k], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&hs.done[k], hipEventDisableTiming));
        }
    }
    const int k = hs.turn & 1;
    // everything enqueued so far (the step that used the OTHER slot included) precedes this marker
    if (hs.turn > 0) { HIPCHK(hipEventRecord(hs.done[k ^ 1], st)); hs.done_rec[k ^ 1] = true; }
    // slot k was used two steps ago: its kernels must be finished before its device buffers are overwritten, and its
    // last DMA before the pinned buffer is (both long done unless the host runs two steps ahead)
    if (hs.done_rec[k]) HIPCHK(hipEventSynchronize(hs.done[k]));
    ++hs.turn;
    char *pin = hs.pin[k], *dev = hs.dev[k];
    memcpy(pin + hs.off[0], b->ids, nb_ids);
    if (nb_off) memcpy(pin + hs.off[1], b->offsets, nb_off);
    if (nb_wide) memcpy(pin + hs.off[2], b->wide_ids, nb_wide);
    if (nb_dense) memcpy(pin + hs.off[3], b->dense, nb_dense);
    if (nb_lab) memcpy(pin + hs.off[4], b->labels, nb_lab);
    hipStream_t cs = hs.copy_stream;
    HIPCHK(hipMemcpyAsync(dev + hs.off[0], pin + hs.off[0], nb_ids, hipMemcpyHostToDevice, cs));
    if (nb_off) HIPCHK(hipMemcpyAsync(dev + hs.off[1], pin + hs.off[1], nb_off, hipMemcpyHostToDevice, cs));
    if (nb_wide) HIPCHK(hipMemcpyAsync(dev + hs.off[2], pin + hs.off[2], nb_wide, hipMemcpyHostToDevice, cs));
    if (nb_dense) HIPCHK(hipMemcpyAsync(dev + hs.off[3], pin + hs.off[3], nb_dense, hipMemcpyHostToDevice, cs));
    if (nb_lab) HIPCHK(hipMemcpyAsync(dev + hs.off[4], pin + hs.off[4], nb_lab, hipMemcpyHostToDevice, cs));
    HIPCHK(hipEventRecord(hs.copied[k], cs));
    HIPCHK(hipStreamWaitEvent(st, hs.copied[k], 0));
    m->cur_ids = reinterpret_cast<const int64_t *>(dev + hs.off[0]);
    m->cur_offsets = nb_off ? reinterpret_cast<const int64_t *>(dev + hs.off[1]) : nullptr;
    m->cur_wide = nb_wide ? reinterpret_cast<const int64_t *>(dev + hs.off[2]) : nullptr;
    m->cur_dense = nb_dense ? reinterpret_cast<const float *>(dev + hs.off[3]) : nullptr;
    m->cur_labels = nb_lab ? reinterpret_cast<const float *>(dev + hs.off[4]) : nullptr;
    return PS_OK;
}

// ---------------------------------------------------------------------------
// the three phases, enqueued on the store's stream
// ---------------------------------------------------------------------------
// Raised wave priority for the step's GEMMs and head (s_setprio, kernels_gemm.hip): where the critical path is the FC chain
// (single-hot: fused and sharded step).  A multi-hot step is bound by its sort chain and the sum of its kernels, and the
// priority takes from exactly those: 0.387 against 0.382 ms.
int g_super_list = 1;   // ps_tune_set("super_list", 0): the chunked order's super partials by the walk over every tile (k_emb_super) instead of the sort's list of very long runs
int g_wide_in_gather = 1;   // ps_tune_set("wide_in_gather", 0): the head walks the wide ids itself again
int g_mh_prio = 0;      // ps_tune_set("mh_prio", 1): raised wave priority for the GEMMs and the head of a MULTI-HOT step too (its sort chain left the critical path with mh_presort)
static bool gemm_prio(const ps_model *m) { return g_main_prio && (m->cur_offsets == nullptr || g_mh_prio); }

// FcLayer.backward arguments of the out = 1 layer (layer/FcLayer.java:93-110): delta_prev = W^T delta (outer
// product) and dW/db (column sums over the batch) in one pass over the layer's input instead of two sliver GEMMs
static void fill_last_bwd(ps_model *m, LastBwdArgs &q) {
    ps_store *s = m->s;
    const ps_model_config_t &c = m->cfg;
    const int l = c.nfc - 1, B = m->cur_B;
    FcParams &p = s->fc[l];
    FcBuf &b = m->fc[l];
    memset(&q, 0, sizeof q);
    q.B = B; q.K = p.K; q.Kp = p.Kpad; q.chunk = cdiv(B, b.nsplit);
    q.A = b.A; q.lda = b.ldA; q.dlast = b.dOut; q.ldd = b.ldD; q.W = p.W; q.ldw = p.ldw;
    if (l > 0) { q.dprev = m->fc[l - 1].dOut; q.ldp = m->fc[l - 1].ldD; q.dprev_cols = p.K; q.mask_cols = p.K; }
    else { q.dprev = m->dx; q.ldp = m->ldx; q.dprev_cols = c.F * c.D; q.mask_cols = c.F * c.D; }
    q.part = b.part; q.part_stride = b.part_stride; q.ldpart = b.ldp; q.skip = nullptr;
    q.prio = gemm_prio(m) ? 1 : 0;
}

// Can the backward of this step release its side chains by device flags (launch_spin_until) instead of events?
// Needs two side streams, no stream capture and the fused head (so that the first kernel after it is a delta GEMM):
// see enqueue_backward.
static bool dev_release(const ps_model *m) {
    const ps_model_config_t &c = m->cfg;
    return m->dev_ok && !c.use_graph && !m->profile && m->multi_stream && c.nfc >= 2 && m->s->fc[c.nfc - 1].N == 1;
}

// the one-launch field sort of a single-hot batch (keys left in m->keys by the gather) on stream ss
static int enqueue_field_sort(ps_model *m, hipStream_t ss) {
    ps_store *s = m->s;
    const ps_model_config_t &c = m->cfg;
    Prof pf(m, "emb_sort");
    if (++m->fs_epoch == 0) ++m->fs_epoch;
    int64_t span = 1;
    for (int f = 0; f < c.F; ++f) span = std::max(span, s->emb.row_base[f + 1] - s->emb.row_base[f]);
    PSCHK(field_sort_segments(m->keys, s->emb.row_base_dev, bits_for(span), m->cur_B, c.F, PS_EMB_SEQ_TILE, m->fs_keys, m->fs_ents,
                              m->seg_start, m->seg_id, m->nseg_dev, m->long_list, m->fs_pub, m->fs_epoch, ss));
    m->sorted_keys = m->fs_keys; m->sorted_ents = m->fs_ents;
    m->long_list_valid = true; m->nlong_ptr = m->nseg_dev + 1; m->field_sorted = true;
    m->side0_pending = true;
    return PS_OK;
}

int enqueue_forward(ps_model *m, bool train, bool defer_loss) {
    ps_store *s = m->s;
    const ps_model_config_t &c = m->cfg;
    hipStream_t st = s->stream;
    const int B = m->cur_B, nfc = c.nfc;
    // EmbeddingLayer.forward + ConcatLayer.forward
    EmbFwdArgs e;
    memset(&e, 0, sizeof e);
    e.W = s->emb.W; e.row_base = s->emb.row_base_dev; e.ids = m->cur_ids; e.offsets = m->cur_offsets;
    e.B = B; e.F = c.F; e.D = c.D; e.X = c.X; e.act = PS_ACT_RELU;   // EmbeddingLayer.build: Relu (EmbeddingLayer.java:53)
    e.out = m->fc[0].A; e.ld = m->fc[0].ldA; e.dense = m->cur_dense;
    e.table_bytes = sizeof(float) * (size_t)s->emb.total_rows * c.D;
    e.key_out = train ? m->keys : nullptr;
    e.ent_bag = (train && m->cur_offsets) ? m->ent_bag : nullptr;
    e.err = s->err_dev;
    if (m->sh.active) {
        // sharded worker: rows come from the cache pulled from their owners (store/KVStore.java:96),
        // keys and the sort were made by ps_shard_plan
        e.W = m->sh.cache; e.slot = m->sh.slot; e.key_out = nullptr; e.ent_bag = nullptr; e.table_bytes = 0;
        e.W_alt = m->sh.alt_W; e.alt_lo = m->sh.alt_lo; e.alt_hi = m->sh.alt_hi;      // this rank's own rows: never copied (ps_comm.hip)
    }
    // multi-hot: the sort's keys come from the ids alone (k_emb_keys), so the whole sort chain starts BESIDE the gather.
    // (Beside the gather the key kernel takes the gather's ~39 us -- the two share the memory system.  Running it alone
    // in front of the gather, 16 us, and releasing the sort by a flag starts the sort chain 40 us earlier and ends the
    // step no sooner: 0.388 vs 0.382 ms -- the later kernels then overlap the dW GEMMs and all of them slow down.  This
    // shape is bound by the sum of its kernels, not by a chain.)
    const bool keys_early = g_keys_early && train && !m->sh.active && m->cur_offsets && side_stream(m, 0) != st;
    // ... and the partition by FIELD costs no radix pass (round 4, kernels_sort.hip k_bag_scan): a column scan of the bag lengths, the
    // key kernel writes (id, bag) at the entry's place among its field's entries, two 9-bit passes sort every field on its own
    bool seg_sort = false, presort = false;
    const bool mh_late_ok = g_mh_presort == 3 && m->dev_ok && !c.use_graph && !(nfc == 1 && s->fc[0].N == 1);
    if (keys_early) {
        if (m->seg_fits < 0) {          // (the tables' shapes do not change: decided once)
            std::vector<int64_t> rows_f((size_t)c.F);
            for (int f = 0; f < c.F; ++f) rows_f[f] = s->emb.row_base[f + 1] - s->emb.row_base[f];
            m->seg_fits = seg_sort_fits(rows_f.data(), c.F) ? 1 : 0;
        }
        seg_sort = m->seg_fits == 1 && g_mh_seg_sort;
        if (seg_sort && !m->seg.pre) {
            RtGuard rt_guard;
            // (no memory for the workspace: this model sorts its multi-hot batches with the three-pass radix sort from now on)
            if (seg_sort_alloc(m->seg, m->nnz_cap, (int64_t)m->Bcap * c.F, c.F) != PS_OK) { m->seg_fits = 0; seg_sort = false; }
            else s->bytes += seg_sort_bytes(m->seg);
        }
        // The column scan on the MAIN chain in front of the gather, 7 us.  (Beside the gather it takes the gather's 45-57 us like
        // everything else that shares the memory system with it, and the key kernel -- with it the whole sort chain -- starts
        // behind it: 0.400 against 0.391 ms.  It reads the batch's offsets only and could run beside the PREVIOUS step's
        // embedding update -- but a device batch is valid in the order of the store's stream, which side chain 0 joins only here.)
        // Round 5 (mh_presort): for a device-resident batch the scan, the key kernel and the sort's FIRST pass touch nothing but the
        // batch and the sort's own workspace -- they go to side chain 0 WITHOUT a join with the training stream, i.e. they run while
        // the PREVIOUS step's backward (chunk partials, per-key reduce + Ftrl: latency-bound, a quarter of the HBM rate) still runs,
        // as far ahead as the host is.  The second pass writes the arrays that backward reads (sorted pairs, segments): the join with
        // the training stream sits in front of it.  (A device batch is complete when it is handed over: ps_native.h ps_batch_t.)
        presort = seg_sort && g_mh_presort && m->cur_on_device && !m->profile && m->multi_stream && !c.use_graph && side_stream(m, 0) != st;
        // (host batches are staged in the training stream's order: they keep the join in front of the key kernel)
        // (mh_presort = 2: held back until the previous step's embedding backward has STARTED -- beside the FC chain of that step,
        //  where an idle side chain 0 would otherwise start them at once, they slow its GEMMs: 0.402 against 0.388 ms)
        if (presort && g_mh_presort >= 2 && m->emb_started_valid && m->dev_ok)
            PSCHK(launch_spin_until(m->start_flag + 2, m->emb_started_epoch, side_stream(m, 0), s->werr(), 16));
        if (seg_sort) { Prof pf(m, "emb_bag_scan"); PSCHK(seg_sort_scan(m->seg, m->cur_offsets, B, c.F, presort ? side_stream(m, 0) : st)); }
        if (!presort) PSCHK(fork(m, st, side_stream(m, 0)));          // behind the staging of this batch and the previous step
        if (seg_sort) {
            Prof pf(m, "emb_keys");
            PSCHK(launch_emb_keys_seg(e, m->seg.pre, m->seg.ftotal, seg_sort_tile(), m->seg.kp, m->seg.vp, side_stream(m, 0)));
        } else { Prof pf(m, "emb_keys"); PSCHK(launch_emb_keys(e, side_stream(m, 0))); }
        if (presort) {
            Prof pf(m, "emb_sort");
            PSCHK(seg_sort_pairs(m->seg, m->cur_nnz, s->emb.row_base_dev, m->ws.keys_alt, m->ws.vals_alt, side_stream(m, 0), 1));
            // The previous step's backward reads what the second pass overwrites.  mh_presort = 3 holds the second pass behind a
            // spinner that this step's first forward GEMM releases -- on the training stream, behind that backward: the order is
            // already there, and an event recorded on the training stream between the embedding update and the next gather made
            // the gather start 19-23 us after the update's end instead of 4 (tools/gpu_timeline.py, round 5).
            if (!mh_late_ok) PSCHK(fork(m, st, side_stream(m, 0)));
        }
        e.key_out = nullptr; e.ent_bag = nullptr;
    }
    m->seg_sorted = seg_sort;
    // single-hot field sort released from the device: the FIRST forward GEMM's start (it starts only after the gather
    // has finished) flips a flag, the sort sits behind a spinner on side chain 0 -- the gather's launch carries no
    // event (a launch with a stop event starts ~2 us later than a plain one)
    const bool sort_dev = train && !m->sh.active && !m->cur_offsets && g_field_sort && field_sort_fits(B, c.F) && !g_sort_ablate &&
                          m->dev_ok && !c.use_graph && side_stream(m, 0) != st && !(nfc == 1 && s->fc[0].N == 1);
    LaunchOpts fwd_lo;
    fwd_lo.stop_event = (train && !m->sh.active && !keys_early && !sort_dev) ? pick_event(m) : nullptr;
    const hipEvent_t fwd_ev = fwd_lo.stop_event;
    if (!m->sh.active && s->pending_ev) {
        // the previous fused step's dense update (side chain 1 of the model that ran it) may still be running: this
        // launch's first workgroup ends only once the update's end flag is up -- the gather runs beside the update's
        // tail, the first GEMM (which reads W) starts behind both.  Without device-side joins: the event.
        if (m->dev_ok && s->pending_flag && s->pending_start && !m->profile) {
            fwd_lo.wait = s->pending_flag; fwd_lo.wait_val = s->pending_val;
            // ... and no workgroup of it STARTS before that update has started: the dW GEMMs in front of the update (side
            // chain 1) read the activations of the previous step, which this launch overwrites
            e.start_wait = s->pending_start; e.start_val = s->pending_val;
        } else PSCHK(store_settle(s));
    } else if (m->sh.active) PSCHK(store_settle(s));
    if (m->sh.active && m->sh.flat_pending && m->sh.flat_by_flag) {
        // sharded step: the previous step's replicated update (side chain 1) must be done before the first GEMM reads W --
        // the gather's first workgroup ends only once the update's end flag is up (normally it has been for a while)
        fwd_lo.wait = m->start_flag + 9; fwd_lo.wait_val = m->sh.flat_epoch;
    }
    if (m->sh.active && m->sh.flat_start_valid) {
        // ... and no workgroup of the gather STARTS before that step's flat-gradient launch has started: the dW GEMMs in front of it
        // read the activations this launch overwrites (see enqueue_backward)
        if (m->dev_ok && !m->profile) { e.start_wait = m->start_flag + 12; e.start_val = m->sh.flat_start_epoch; }
        else if (m->flat_stream && m->flat_stream != st) {        // (no device-side joins any more: an event behind that launch)
            hipEvent_t ev = m->events[m->next_event++ % m->events.size()];
            HIPCHK(hipEventRecord(ev, m->flat_stream));
            HIPCHK(hipStreamWaitEvent(st, ev, 0));
        }
        m->sh.flat_start_valid = false;
    }
    // wide_in_gather: LRLayer.forward (sample b's F wide weights summed in field order + bias) as a role of this launch -- the head, three
    // launches down the main chain, then reads wide_z[b] instead of walking id -> weight itself (one memory round trip less on the chain).
    // The wide table is what the previous step's wide update left: that update sits on the chain the main chain joined before its embedding
    // update.  (Sharded worker: the replicated update may still run beside this launch -- the head keeps the walk.)
    const bool wide_in_gather = g_wide_in_gather && c.kind == PS_MODEL_WIDEDEEP && !m->sh.active && m->cur_wide && B > 0;
    if (wide_in_gather) {
        e.wide_ids = m->cur_wide; e.wide_rows = s->wide.rows; e.wide_w = s->wide.W; e.wide_bias = s->wide.bias;
        e.wide_touched = s->wide.touched; e.wide_z = m->wide_z; e.wide_train = train ? 1 : 0;
    }
    { Prof pf(m, "emb_fwd"); PSCHK(launch_emb_fwd(e, st, &fwd_lo, s->werr())); }
    const bool wide_done = wide_in_gather && fwd_lo.launched;
    if (fwd_lo.wait && !m->sh.active) {          // (the fused step's deferred join)
        if (!fwd_lo.launched) PSCHK(store_settle(s));
        s->pending_ev = nullptr; s->pending_flag = nullptr;
    } else if (fwd_lo.wait) {
        if (!fwd_lo.launched) PSCHK(launch_spin_until(m->start_flag + 9, m->sh.flat_epoch, st, s->werr(), 9));   // (an empty batch)
        m->sh.flat_pending = false;
    }
    if (fwd_lo.stop_event && !fwd_lo.launched) HIPCHK(hipEventRecord(fwd_lo.stop_event, st));
    // mh_presort = 3: the sort's second half (second pass, run boundaries) is released by the first forward GEMM's START, like the
    // single-hot field sort: launched right behind the join it ran beside the gather (its histogram pass 37 us instead of 10) and
    // its 806-workgroup scatter was being placed when the first GEMM arrived -- that GEMM started 20 us after the gather's end
    const bool mh_late = presort && mh_late_ok;
    auto enqueue_sort = [&]() -> int {
        // the sort only needs the row keys the gather just emitted: run it beside the FC chain
        hipStream_t ss = side_stream(m, 0);
        if (mh_late) PSCHK(launch_spin_until(m->start_flag + 4, m->fwd_epoch, ss, s->werr(), 17));
        else if (keys_early) {}
        else if (sort_dev) PSCHK(launch_spin_until(m->start_flag + 4, m->fwd_epoch, ss, s->werr(), 4));
        else if (fwd_ev) PSCHK(wait_event(m, ss, fwd_ev)); else PSCHK(fork(m, st, ss));
        const int64_t nnz = m->cur_nnz;
        static int64_t sort_runs = 0;
        m->long_list_valid = false; m->field_sorted = false;
        if (g_sort_ablate && ++sort_runs > 4) {
            // measurement only (tools/sort_ablate.py, ONE batch repeated): the step without its sort chain
        } else if (!m->cur_offsets && g_field_sort && field_sort_fits(B, c.F)) {
            // single-hot: every field's keys live in their own interval, F sorts of B pairs in one launch
            // (sorted pairs, segments and the list of long runs; kernels_sort.hip)
            Prof pf(m, "emb_sort");
            if (++m->fs_epoch == 0) ++m->fs_epoch;
            int64_t span = 1;
            for (int f = 0; f < c.F; ++f) span = std::max(span, s->emb.row_base[f + 1] - s->emb.row_base[f]);
            PSCHK(field_sort_segments(m->keys, s->emb.row_base_dev, bits_for(span), B, c.F, PS_EMB_SEQ_TILE, m->fs_keys, m->fs_ents,
                                      m->seg_start, m->seg_id, m->nseg_dev, m->long_list, m->fs_pub, m->fs_epoch, ss));
            m->sorted_keys = m->fs_keys; m->sorted_ents = m->fs_ents;
            m->long_list_valid = true; m->nlong_ptr = m->nseg_dev + 1; m->field_sorted = true;
        } else {
            {
                Prof pf(m, "emb_sort");
                // payload of the sort = the BAG of every entry (single-hot: entry == bag, an iota).  The backward only
                // needs each entry's delta row (its bag's) in entry order, which the stable sort keeps: carrying the bag
                // through the sort saves the entry -> bag indirection (a random 4-byte load per entry) in both backward
                // kernels.  ent_bag is consumed as ping-pong storage here.
                if (m->cur_offsets && m->seg_sorted) {
                    PSCHK(seg_sort_pairs(m->seg, nnz, s->emb.row_base_dev, m->ws.keys_alt, m->ws.vals_alt, ss, presort ? 2 : 3));
                    m->sorted_keys = m->ws.keys_alt; m->sorted_ents = m->ws.vals_alt;
                } else if (m->cur_offsets)
                    PSCHK(radix_sort_pairs(m->ws, m->keys, m->ent_bag, nnz, bits_for(s->emb.total_rows), false, &m->sorted_keys,
                                           &m->sorted_ents, ss));
                else
                    PSCHK(radix_sort_pairs(m->ws, m->keys, m->ents, nnz, bits_for(s->emb.total_rows), true, &m->sorted_keys,
                                           &m->sorted_ents, ss));
            }
            {
                Prof pf(m, "emb_segments");
                // (the list of long runs only where the sequential order will walk it)
                const bool want_seq = c.emb_sum_order == PS_SUM_SEQUENTIAL || (c.emb_sum_order == PS_SUM_AUTO && !m->cur_offsets);
                // (chunked order: the runs above PS_EMB_CHUNK * PS_EMB_SUPER_MIN entries, for k_emb_super_list)
                const bool want_list = want_seq || g_super_list;
                // (... or, round 6, above emb_list_min chunks: the update launch's list role takes every run a lane group would walk for long)
                m->list_min = (g_super_in_update && g_emb_list_min > 0 && g_emb_list_min < PS_EMB_SUPER_MIN) ? g_emb_list_min : PS_EMB_SUPER_MIN;
                PSCHK(build_segments(m->ws, m->sorted_keys, nnz, m->seg_start, m->seg_id, m->nseg_dev, ss, want_list ? m->long_list : nullptr,
                                     want_seq ? PS_EMB_SEQ_TILE : PS_EMB_CHUNK * m->list_min));
                m->long_list_valid = want_list; m->nlong_ptr = m->nseg_dev + 1;
            }
        }
        m->side0_pending = true;
        return PS_OK;
    };
    if (sort_dev) m->field_sorted = true;            // (dev_release() below looks at it before the sort is enqueued)
    else if (train && !m->sh.active && !mh_late) PSCHK(enqueue_sort());
    // late: the sort is not released by the first forward GEMM but enqueued by the backward behind the first delta GEMM's
    // release -- beside the forward GEMMs its 26 workgroups share CUs with fc_fwd1's one-workgroup-per-CU grid
    const bool sort_late = sort_dev && g_sort_late;
    m->sort_deferred = false;
    bool sort_due = (sort_dev && !sort_late) || mh_late;
    // sharded worker: the first forward GEMM announces its start too -- everything of this step in front of it (the id and
    // row exchanges, the gather) is then done, which is what the NEXT step's plan waits for on side chain 0
    bool fwd_flag_due = train && m->sh.active && m->dev_ok && !c.use_graph && !m->profile && m->multi_stream;
    m->fwd_flag_valid = false;
    // LRLayer.forward + AddLayer.forward + loss
    HeadArgs h;
    memset(&h, 0, sizeof h);
    h.B = B; h.F = c.F; h.wide = c.kind == PS_MODEL_WIDEDEEP; h.train = train;
    h.zlast = m->out_last; h.ldz = m->ld_last;
    h.wide_ids = m->cur_wide; h.wide_rows = s->wide.rows; h.wide_w = s->wide.W; h.wide_bias = s->wide.bias;
    h.touched = s->wide.touched;
    h.labels = m->cur_labels;
    h.P = m->P; h.wide_z = m->wide_z; h.terms = m->terms;
    h.wide_z_in = wide_done ? m->wide_z : nullptr;
    h.dlast = m->fc[nfc - 1].dOut; h.ldd = m->fc[nfc - 1].ldD;
    h.err = s->err_dev;
    if (s->fc[nfc - 1].N == 1) {
        const FcParams &pl = s->fc[nfc - 1];
        h.a_last = m->fc[nfc - 1].A; h.lda_last = m->fc[nfc - 1].ldA; h.w_last = pl.Wt; h.k_last = pl.Kpad;
        h.last_sigmoid = c.kind == PS_MODEL_WIDEDEEP ? 0 : 1;    // FcLayer.java:58-62, WideDeepNN.java:128
        h.zout = m->out_last;
    }
    m->head_args = h;
    m->head_bwd_done = false;
    const FcParams &pl = s->fc[nfc - 1];
    FcBuf &bl = m->fc[nfc - 1];
    // Two hidden (relu) layers of the built shape in front of the out = 1 layer: the FC forward chain of a 16-row panel is ONE
    // launch (kernels_panel.hip), with the head and the out = 1 layer's backward of the same rows when this is a training step
    // whose head workgroups own 16 rows too
    const bool panel = g_fwd_panel && nfc == 3 && pl.N == 1 && s->fc[0].Wp && s->fc[1].Wp && !g_gemm_ablate &&
                       fwd_panel_shape_ok(s->fc[0].Kpad, s->fc[0].N, s->fc[1].N) && m->fc[1].ldA == s->fc[1].Kpad;
    const bool head_fusable = train && pl.N == 1 && h.labels && head_last_bwd_fusable(cdiv(B, bl.nsplit));
    const bool panel_head = panel && head_fusable && g_fwd_panel >= 2 && cdiv(B, bl.nsplit) == PS_PANEL_ROWS;
    // FcLayer.forward x nfc
    for (int l = 0; l < nfc; ++l) {
        FcParams &p = s->fc[l];
        float *out = l + 1 < nfc ? m->fc[l + 1].A : m->out_last;
        const int ldo = l + 1 < nfc ? m->fc[l + 1].ldA : m->ld_last;
        int epi = EPI_RELU;
        if (l == nfc - 1) epi = c.kind == PS_MODEL_WIDEDEEP ? EPI_NONE : EPI_SIGMOID;   // FcLayer.java:58-62, WideDeepNN.java:128
        static const char *names[8] = {"fc_fwd0", "fc_fwd1", "fc_fwd2", "fc_fwd3", "fc_fwd4", "fc_fwd5", "fc_fwd6", "fc_fwd7"};
        if (l == nfc - 1 && p.N == 1) break;      // the out = 1 layer is a per-sample dot product inside k_head
        // two consecutive hidden (relu) layers: ONE launch, the second layer's tiles start as their row panel of the first
        // layer's output completes (kernels_gemm.hip k_fc_fwd_pair)
        const bool panel_l = panel && l == 0;
        const bool pair = !panel_l && l + 2 < nfc && !s->fwd_pair_off && m->pair_ctr &&
                          gemm_nt_fwd_pair_ok(B, p.N, s->fc[l + 1].N, p.Kpad, s->fc[l + 1].Kpad);
        Prof pf(m, panel_l ? (panel_head ? "fwd_panel_head" : "fwd_panel") : pair ? (l == 0 ? "fc_fwd01" : "fc_fwd_pair") : names[l]);
        LaunchOpts lo;
        // (sort_layer: which forward GEMM's start releases the field sort -- 0, the first; measurement knob)
        const bool sort_here = sort_due && (panel_l || l >= g_sort_layer || l + 1 >= nfc - 1);
        if (sort_here || fwd_flag_due) { if (++m->fwd_epoch == 0) ++m->fwd_epoch; lo.flag = m->start_flag + 4; lo.flag_val = m->fwd_epoch; }
        lo.prio = (train && gemm_prio(m)) ? 1 : 0;
        if (panel_l) {
            FwdPanelArgs pa;
            memset(&pa, 0, sizeof pa);
            pa.X = m->fc[0].A; pa.ldx = m->fc[0].ldA; pa.B = B; pa.Kpad0 = p.Kpad; pa.N0 = p.N; pa.N1 = s->fc[1].N;
            pa.W0p = p.Wp; pa.W1p = s->fc[1].Wp;
            pa.H1 = m->fc[1].A; pa.ld1 = m->fc[1].ldA; pa.H2 = m->fc[2].A; pa.ld2 = m->fc[2].ldA;
            if (panel_head) {
                LastBwdArgs q;
                fill_last_bwd(m, q);
                // (both side chains of the backward are released from the device when they can be -- dev_release() -- and the
                //  launch then carries nothing: a launch with a stop event starts ~2 us later than a plain one)
                lo.stop_event = m->head_ev = dev_release(m) ? nullptr : pick_event(m);
                PSCHK(launch_fwd_panel(pa, &q, &h, bl.nsplit, st, &lo, s->werr()));
                PSCHK(settle_event(m, lo));
                m->head_bwd_done = true;
            } else PSCHK(launch_fwd_panel(pa, nullptr, nullptr, 0, st, &lo, s->werr()));
        } else if (pair) {
            FcParams &p2 = s->fc[l + 1];
            float *out2 = l + 2 < nfc ? m->fc[l + 2].A : m->out_last;
            const int ldo2 = l + 2 < nfc ? m->fc[l + 2].ldA : m->ld_last;
            PSCHK(gemm_nt_fwd_pair(m->fc[l].A, m->fc[l].ldA, B, p.Wt, p.Kpad, p.N, out, ldo, p.Kpad, p2.Wt, p2.Kpad, p2.N, out2, ldo2, p2.Kpad, B,
                                   m->pair_ctr, &m->pair_epoch, reinterpret_cast<unsigned int *>(s->err_dev) + 4, st, &lo, s->werr()));
        } else
        PSCHK(gemm_nt(m->fc[l].A, m->fc[l].ldA, B, p.Wt, p.Kpad, p.N, out, ldo, B, p.N, p.Kpad, epi,
                      nullptr, 0, 0, nullptr, st, &lo, s->werr()));
        if (lo.flag && !lo.launched) PSCHK(launch_flag_set(m->start_flag + 4, m->fwd_epoch, st));      // (an empty GEMM)
        if (fwd_flag_due) {
            fwd_flag_due = false; m->fwd_flag_valid = true;
            if (m->sh.active) PSCHK(shard_launch_deferred_sort(m, true));      // (the waiter after the launch that releases it)
        }
        if (sort_here) {        // the waiter is enqueued after the launch that releases it
            PSCHK(enqueue_sort());
            sort_due = false;
        }
        if (pair || panel_l) ++l;          // (layer l + 1 went with this launch)
    }
    if (m->sh.active) PSCHK(shard_launch_deferred_sort(m, false));      // (no GEMM carried the start flag: the sort at once)
    if (panel_head) {
        // (went with the panel launch)
    } else if (head_fusable) {
        // training: the head and the out = 1 layer's backward of the same rows in ONE launch (three tiny kernels
        // of the critical chain become one; the loss reduction leaves the chain altogether, see enqueue_backward)
        LastBwdArgs q;
        fill_last_bwd(m, q);
        Prof pf(m, "head_last_bwd");
        // (both side chains of the backward are released from the device when they can be -- dev_release() -- and the
        //  head's launch then carries nothing: a launch with a stop event starts ~2 us later than a plain one)
        LaunchOpts head_lo;
        head_lo.stop_event = m->head_ev = dev_release(m) ? nullptr : pick_event(m);
        PSCHK(launch_head_last_bwd(h, q, bl.nsplit, st, &head_lo));
        PSCHK(settle_event(m, head_lo));
        m->head_bwd_done = true;
    } else {
        Prof pf(m, "head");
        PSCHK(launch_head(h, nullptr, nullptr, nullptr, 0, st));
    }
    if (sort_late) {
        if (dev_release(m) && m->head_bwd_done && m->side[1] != st) m->sort_deferred = true;     // (enqueue_backward: first release)
        else { PSCHK(fork(m, st, side_stream(m, 0))); PSCHK(enqueue_field_sort(m, side_stream(m, 0))); }
    }
    m->loss_pending = h.labels != nullptr;
    if (m->loss_pending && !defer_loss) {
        Prof pf(m, "loss_reduce");
        PSCHK(launch_loss_reduce(h, m->loss_dev, m->gbar_dev, m->skip_dev, m->sh.active ? 1 : 0, st));
        m->loss_pending = false;
    }
    return PS_OK;
}

// wide_grad_mode = intended (not a reference path): the gradient of a wide key is the sum of delta
// over the (sample, field) occurrences of the key IN THIS BATCH, / B -- a stable sort of the batch's wide ids, their
// segments, one sequential sum per key, Ftrl on those keys only; "wide.bias" as in compat mode.
// to_flat (sharded worker, gradients only): the keys' gradients go to the flat buffer's [G | C] part instead of the table --
// G[key] = the batch's sum / B, C[key] = 1 for the keys of THIS worker's batch, zero elsewhere; the all-reduce and
// ps_shard_apply_flat then give every key the mean over the workers that pushed it (net/PServer.java:164-214).
static int enqueue_wide_intended(ps_model *m, WideUpdArgs w, hipStream_t st, bool to_flat = false) {
    ps_store *s = m->s;
    const int64_t n = (int64_t)m->cur_B * m->cfg.F;
    if (!m->wkeys) {
        const int64_t cap = (int64_t)m->Bcap * m->cfg.F;
        PSCHK(sort_ws_alloc(m->wws, cap));
        PSCHK(model_alloc(m, (void **)&m->wkeys, sizeof(uint32_t) * (size_t)(cap + 1), false));
        PSCHK(model_alloc(m, (void **)&m->wents, sizeof(uint32_t) * (size_t)(cap + 1), false));
        PSCHK(model_alloc(m, (void **)&m->wseg_start, sizeof(uint32_t) * (size_t)(cap + 2), false));
        PSCHK(model_alloc(m, (void **)&m->wseg_id, sizeof(uint32_t) * (size_t)(cap + 1), false));
        PSCHK(model_alloc(m, (void **)&m->wnseg, sizeof(uint32_t) * 4, true));
    }
    PSCHK(launch_wide_keys(m->cur_wide, n, s->wide.rows, m->wkeys, s->err_dev, st));
    uint32_t *sk = nullptr, *se = nullptr;
    PSCHK(radix_sort_pairs(m->wws, m->wkeys, m->wents, n, bits_for(s->wide.rows), true, &sk, &se, st));
    PSCHK(build_segments(m->wws, sk, n, m->wseg_start, m->wseg_id, m->wnseg, st));
    WideIntendedArgs a;
    memset(&a, 0, sizeof a);
    a.sorted_key = sk; a.sorted_ent = se; a.seg_start = m->wseg_start; a.nseg = m->wnseg;
    a.delta = m->fc[m->cfg.nfc - 1].dOut; a.ldd = m->fc[m->cfg.nfc - 1].ldD;
    a.B = m->cur_B; a.F = m->cfg.F; a.W = w.W; a.state = w.state; a.upd = w.upd; a.skip = w.skip;
    if (to_flat) {
        a.G = m->sh.flat + m->dense_elems; a.C = a.G + s->wide.rows;
        HIPCHK(hipMemsetAsync(a.G, 0, sizeof(float) * 2 * (size_t)s->wide.rows, st));
        return launch_wide_intended(a, n, st);              // ("wide.bias": by the flat gradient's launch, mode 5)
    }
    PSCHK(launch_wide_intended(a, n, st));
    w.mode = 3;                                             // "wide.bias": rowMeans(delta), as in compat mode
    return launch_wide_update(w, st);
}

int enqueue_backward(ps_model *m, bool apply) {
    ps_store *s = m->s;
    const ps_model_config_t &c = m->cfg;
    hipStream_t st = s->stream;
    const int B = m->cur_B, nfc = c.nfc;
    const int *skip = m->skip_dev;
    ps_updater_t u;
    UpdParams emb_upd;
    FieldUpd emb_fu;
    bool emb_stateful = false;
    PSCHK(store_fill_field_upd(s, &emb_upd, &emb_fu, &emb_stateful));
    if (apply && !s->emb.state && emb_stateful)                  // before anything is enqueued
        return ps_set_err(PS_E_STATE, "the embedding table was created weights-only (state_slots = 0): Adam / Ftrl cannot train it");
    // Main chain: delta GEMMs, embedding update, dense update (last: nothing crosses a stream at the step boundary).
    // Side chain 1: every dW GEMM as soon as its delta exists.  Side chain 0 (the sort ran there during the forward):
    // loss reduction, wide update, the out = 1 layer's slab fold.  DESIGN.md 4.1 has the measured timeline.
    hipStream_t sw = side_stream(m, 1), s0 = side_stream(m, 0);
    // The small kernels that hang off the head (loss / stop flag, wide update, the out = 1 layer's slab fold) go behind
    // the sort on side chain 0 when the sort is the one-launch field sort (done long before the head).  Behind the
    // 11-launch radix chain of a multi-hot batch (136 us at configs[4]'s shape, ending after the last delta GEMM) they
    // would sit on the critical path: there they go to the front of side chain 1 instead.
    hipStream_t sl = (m->field_sorted || m->sh.active || s0 == st) ? s0 : sw;      // (sharded: that sort ran during the exchange)
    // side chain 1 (the dW GEMMs) does not wait for the head by event: a spinner in front of its first GEMM is released
    // by the first delta GEMM's start (launch_spin_until in kernels_gemm.hip)
    // (the head's small kernels go behind the same release: on side chain 0 behind their own spinner, or -- multi-hot --
    // at the front of side chain 1)
    // (and not under stream capture: a captured graph needs its side streams joined by events)
    const bool dev_flags = m->dev_ok && !m->cfg.use_graph;
    const bool dev_wait = dev_release(m) && sw != st && m->head_bwd_done;
    if (dev_wait) {}                                              // both chains: spinners behind the first delta GEMM's launch
    else if (m->head_ev && m->head_bwd_done) { PSCHK(wait_event(m, sw, m->head_ev)); PSCHK(wait_event(m, s0, m->head_ev)); }
    else PSCHK(fork2(m, st, sw, s0));
    bool first_release = dev_wait;
    // (dw_split needs the fused tail: its dense update is what waits for both chains' GEMMs)
    const bool dw_split = g_dw_split && dev_wait && dev_flags && g_tail_dev && g_tail_fused && sw != st && s0 != st && s0 != sw && sl == s0 &&
                          !m->profile && m->cur_nnz > 0 && !m->sh.active;
    bool dw_split_done = false; unsigned int dw_split_epoch = 0;
    int deferred_l = -1;                      // dw_late: the dW GEMM held back for the next release
    bool sw_gated = false;                    // side chain 1 already sits behind a spinner: later dW GEMMs wait at their own start
    bool s0_joined = false;                   // the last delta GEMM's launch carries the join with side chain 0
    m->head_ev = nullptr;
    bool main_dirty = false;           // a kernel went onto the main chain since the last fork towards sw
    hipEvent_t data_ev = nullptr;      // carried by the last delta GEMM on the main chain, not yet waited on
    unsigned int *const werr = s->werr();
    // dense tensors: reduce the splits, / B, updater  (KVStore.update for "fc*.weights"/"fc*.bias")
    DenseUpdArgs d;
    memset(&d, 0, sizeof d);
    d.nlayers = nfc; d.B = B; d.apply = apply ? 1 : 0; d.skip = skip;
    PSCHK(store_resolve_updater(s, "fc0.weights", &u));
    d.upd = make_upd_params(u);
    d.grad_out = m->sh.active ? m->sh.flat : m->dense_grad_flat;   // sharded: straight into the all-reduce buffer
    int64_t off = 0;
    for (int l = 0; l < nfc; ++l) {
        FcParams &p = s->fc[l];
        DenseLayer &L = d.L[l];
        L.W = p.W; L.Wt = p.Wt; L.Wp = p.Wp; L.S1 = p.S1; L.S2 = p.S2;
        L.part = m->fc[l].part; L.part_stride = m->fc[l].part_stride; L.ldp = m->fc[l].ldp; L.nsplit = m->fc[l].nsplit;
        L.K = p.K; L.N = p.N; L.ldw = p.ldw; L.ldwt = p.Kpad;
        L.elem_begin = off; off += (int64_t)(p.K + 1) * p.N; L.elem_end = off;
    }
    // The small kernels that hang off the head.  Enqueued here when their chain waits for the head by event; behind the
    // first delta GEMM's launch (and a spinner) when it is released from the device.
    const bool sort_dev_wait = dev_flags && s0 != st && sl != s0 && !m->sh.active;
    auto small_kernels = [&]() -> int {
        if (m->loss_pending) {
            // loss = mean(terms), gbar = rowMeans(delta), the stop flag (model/DNN.java:58-63).  Nothing on the main chain
            // needs them before the embedding update: the GEMMs only write scratch, so they run regardless of the flag and
            // only the kernels that touch parameters (wide / dense / embedding updates) honour it.
            Prof pf(m, "loss_reduce");
            PSCHK(launch_loss_reduce(m->head_args, m->loss_dev, m->gbar_dev, m->skip_dev, m->sh.active ? 1 : 0, sl));
            m->loss_pending = false;
        }
        if (c.kind == PS_MODEL_WIDEDEEP && !apply && m->sh.active && c.wide_grad_mode == PS_GRAD_INTENDED) {
            // sharded worker, intended mode: the per-key sums into the flat buffer (this chain ends before the flat
            // gradient's launch starts: that one waits for the embedding backward, which joins this chain first)
            WideUpdArgs w;
            memset(&w, 0, sizeof w);
            w.skip = nullptr;
            Prof pf(m, "wide_update");
            PSCHK(enqueue_wide_intended(m, w, sl, true));
        }
        // wide part: LRLayer.backward (layer/LRLayer.java:100-120) + Ftrl
        if (c.kind == PS_MODEL_WIDEDEEP && apply) {
            WideUpdArgs w;
            memset(&w, 0, sizeof w);
            w.rows = s->wide.rows; w.W = s->wide.W; w.state = s->wide.state; w.touched = s->wide.touched;
            w.bias = s->wide.bias; w.bias_state = s->wide.bias_state; w.gbar = m->gbar_dev; w.skip = skip;
            PSCHK(store_resolve_updater(s, "wide.weights", &u));
            w.upd = make_upd_params(u);
            Prof pf(m, "wide_update");
            if (c.wide_grad_mode == PS_GRAD_INTENDED) PSCHK(enqueue_wide_intended(m, w, sl));
            else PSCHK(launch_wide_update(w, sl));
        }
        // the out = 1 layer's 128 row-block slabs (written by the head's launch) folded here, on side chain 0 behind the
        // wide update, not in front of the dense update at the end of the step
        if (m->head_bwd_done && s->fc[nfc - 1].N == 1) { Prof pf(m, "dense_prereduce"); PSCHK(dense_prereduce(d, nfc - 1, sl)); }
        // everything the main chain needs from side chain 0 ends here (sort, stop flag, wide update, slab fold)
        // A long sort chain (multi-hot) ends AFTER the last delta GEMM: the main chain would reach its wait first and
        // resume 10-20 us after the event.  There the chain's end sets a flag from the device and the main chain parks a
        // spinner in front of the embedding update instead (same mechanism as the dW chain's release).
        if (sort_dev_wait) {
            if (++m->start_epoch == 0) ++m->start_epoch;
            PSCHK(launch_flag_set(m->start_flag + 1, m->start_epoch, s0));
            m->sort_epoch = m->start_epoch;
        } else if (s0 != st && dev_flags) {
            // side chain 0's end as a flag too: a spinner that finds its flag set costs the main chain a tiny kernel
            // (~1.5 us), a hipStreamWaitEvent on an event that fired long ago ~3.5 (tools/gpu_timeline.py)
            if (++m->start_epoch == 0) ++m->start_epoch;
            if (m->sh.active && m->sh.defer_flag5 && s0 == m->side[0]) {      // (raised by the next plan's spinner on this stream)
                m->sh.deferred = true; m->sh.def_flag = m->start_flag + 5; m->sh.def_val = m->start_epoch;
            } else PSCHK(launch_flag_set(m->start_flag + 5, m->start_epoch, s0));
            m->s0_epoch = m->start_epoch;
        } else if (s0 != st) HIPCHK(hipEventRecord(m->s0_ev, s0));
        if (sl != s0) HIPCHK(hipEventRecord(m->loss_ev, sl));
        return PS_OK;
    };
    if (!dev_wait) PSCHK(small_kernels());
    // FcLayer.backward, last to first (layer/FcLayer.java:93-110)
    for (int l = nfc - 1; l >= 0; --l) {
        FcParams &p = s->fc[l];
        FcBuf &b = m->fc[l];
        if (l == nfc - 1 && p.N == 1) {
            if (!m->head_bwd_done) {       // otherwise the head's launch already did this layer's backward
                LastBwdArgs q;
                fill_last_bwd(m, q);
                Prof pf(m, "fc_bwd_last");
                PSCHK(launch_last_bwd(q, b.nsplit, st));
                main_dirty = true;
            }
            continue;       // (its slabs: folded above, or -- produced just now -- inside the dense update)
        }
        // every dW GEMM on side chain 1, in order.  (Alternating them over the two side chains, or holding the last
        // one back until the last delta GEMM is done so that it runs under the embedding update instead: both
        // measured slower -- 0.180 and 0.195 against 0.170 ms/step; the embedding update took 49 us instead of 30
        // with a GEMM beside it.)
        hipStream_t dws = sw;
        // dw_split: the FIRST dW GEMM (its delta comes from the head, long before the others') goes to side chain 0, behind
        // that chain's small kernels, and the later ones to side chain 1 -- the two chains' GEMMs run beside each other
        // instead of one after the other (side chain 1, dW_1 -> dW_0 -> dense update, had become the step's critical
        // path).  The dense update then also waits for "side chain 0's dW GEMM is done" (start_flag[11]).
        const bool split_here = dw_split && first_release && l > 0;
        if (split_here) dws = s0;
        // dw_late: the FIRST dW GEMM is held back until the NEXT delta GEMM starts (one spinner for both dW GEMMs, on that
        // launch's epoch): the first delta GEMM of the critical chain runs alone instead of sharing the matrix pipes with a
        // dW GEMM that has 20 us of slack on its side chain
        const bool defer_this = g_dw_late && dev_wait && !dw_split && sl != sw && l > 0 && first_release;
        LaunchOpts lo, tn_lo;
        if (dev_wait) {
            // released from the device: this delta GEMM's first workgroup announces "everything before me on the main
            // chain is done" (delta_l included), dW_l sits behind a spinner on that -- no event anywhere on the chain
            if (++m->start_epoch == 0) ++m->start_epoch;
            lo.flag = m->start_flag; lo.flag_val = m->start_epoch;
        } else {
            if (main_dirty) {                               // delta_l was just produced on the main chain
                if (data_ev) PSCHK(wait_event(m, dws, data_ev)); else PSCHK(fork(m, st, dws));
                main_dirty = false;
            }
            data_ev = l > 0 ? pick_event(m) : nullptr;  // delta_{l-1}: the next dW GEMM waits for it (nobody after the last)
            lo.stop_event = data_ev;
        }
        lo.prio = gemm_prio(m) ? 1 : 0;
        // delta_prev = W^T delta, with the previous layer's relu' fused in (its backward's first act)
        static const char *nd[8] = {"fc_bwd_data0", "fc_bwd_data1", "fc_bwd_data2", "fc_bwd_data3", "fc_bwd_data4", "fc_bwd_data5", "fc_bwd_data6", "fc_bwd_data7"};
        static const char *nw[8] = {"fc_bwd_dw0", "fc_bwd_dw1", "fc_bwd_dw2", "fc_bwd_dw3", "fc_bwd_dw4", "fc_bwd_dw5", "fc_bwd_dw6", "fc_bwd_dw7"};
        if (l > 0) {
            Prof pf(m, nd[l]);
            PSCHK(gemm_nt(b.dOut, b.ldD, B, p.W, p.ldw, p.K, m->fc[l - 1].dOut, m->fc[l - 1].ldD, B, p.K, b.ldD,
                          EPI_MASK_POS, b.A, b.ldA, p.K, nullptr, st, &lo, werr));
        } else {
            Prof pf(m, nd[l]);
            // the last launch in front of the embedding update also joins side chain 0 (sort, stop flag, wide update,
            // slab fold -- enqueued behind the FIRST delta GEMM, so before this launch): its first workgroup ends only
            // once that chain's end flag is up, and the main chain needs no spinner launch of its own for it
            if (g_end_wait && dev_flags && s0 != st && !first_release) {
                lo.wait = m->start_flag + (sort_dev_wait ? 1 : 5);          // (multi-hot: the end of the long sort chain)
                lo.wait_val = sort_dev_wait ? m->sort_epoch : m->s0_epoch;
            }
            PSCHK(gemm_nt(b.dOut, b.ldD, B, p.W, p.ldw, c.F * c.D, m->dx, m->ldx, B, c.F * c.D, b.ldD,
                          EPI_MASK_POS, b.A, b.ldA, c.F * c.D, nullptr, st, &lo, werr));
            s0_joined = lo.wait != nullptr && lo.launched;
        }
        PSCHK(settle_event(m, lo));
        main_dirty = true;
        if (dev_wait) {             // every waiter is enqueued AFTER the launch that releases it: none can be left spinning
            if (!lo.launched) PSCHK(launch_flag_set(m->start_flag, m->start_epoch, st));   // (an empty GEMM)
            // The FIRST dW GEMM of the chain sits behind a one-wave spinner launch (its stream is idle: without it the
            // GEMM's workgroups would be dispatched at once and hold their CU slots while the forward still runs).  The
            // later ones are in order behind a GEMM that outlasts their release -- "the next delta GEMM has started" --
            // so their workgroups check the flag themselves when they start (one load; start_wait in ps_common.h): no
            // spinner launch between two dW GEMMs (stamps: 7.4 us from the end of dW_1 to the start of dW_0 with it).
            if (split_here) {}       // (side chain 0 is parked behind its own spinner below)
            else if (defer_this) {}  // (released together with the next one)
            else if (!sw_gated || !g_tn_start_wait) { PSCHK(launch_spin_until(m->start_flag, m->start_epoch, sw, werr, 0)); sw_gated = true; }
            else { tn_lo.wait = m->start_flag; tn_lo.wait_val = m->start_epoch; }
            if (first_release) {     // the head's small kernels: on their own chain behind a spinner, or in front of dW_l
                if (sl != sw) PSCHK(launch_spin_until(m->start_flag, m->start_epoch, sl, werr, 10));
                if (m->sort_deferred) { PSCHK(enqueue_field_sort(m, sl)); m->sort_deferred = false; }
                PSCHK(small_kernels());
                first_release = false;
            }
            main_dirty = false;
        }
        if (defer_this) { deferred_l = l; continue; }
        if (deferred_l >= 0) {
            FcParams &dp = s->fc[deferred_l];
            FcBuf &db = m->fc[deferred_l];
            LaunchOpts dlo;
            dlo.prio = gemm_prio(m) ? (g_tn_prio & 7) : 0;
            Prof pfd(m, nw[deferred_l]);
            PSCHK(gemm_tn_splitk(db.A, db.ldA, db.ldA, db.dOut, db.ldD, db.ldD, db.part, db.ldp, db.part_stride, dp.K + 1, dp.N, B,
                                 db.nsplit, nullptr, dws, &dlo, werr));
            deferred_l = -1;
        }
        Prof pf2(m, nw[l]);
        // dW (+ db through the ones column), split over the batch
        tn_lo.prio = gemm_prio(m) ? (((g_tn_prio & 8) && l > 0) ? 1 : (g_tn_prio & 7)) : 0;
        PSCHK(gemm_tn_splitk(b.A, b.ldA, b.ldA, b.dOut, b.ldD, b.ldD, b.part, b.ldp, b.part_stride, p.K + 1, p.N, B,
                             b.nsplit, nullptr, dws, &tn_lo, werr));
        if (split_here) {
            if (++m->start_epoch == 0) ++m->start_epoch;
            dw_split_epoch = m->start_epoch;
            PSCHK(launch_flag_set(m->start_flag + 11, dw_split_epoch, s0));
            dw_split_done = true;
        }
    }
    if (first_release) { PSCHK(fork2(m, st, sw, s0)); PSCHK(small_kernels()); first_release = false; }     // (no delta GEMM at all)
    // tail_dev: the dense update goes to the END OF SIDE CHAIN 1 and both of its edges are device-side flags (no event
    // wait anywhere in the tail): it starts when the embedding update has STARTED (that launch starts only after the
    // last delta GEMM, which reads W_0, has finished and after the main chain saw side chain 0's stop flag and slab
    // fold), and the main chain ends the step behind a spinner on "dense update done".  The update then runs beside
    // the embedding update instead of after it.
    const bool tail_dev = dev_flags && g_tail_dev && sw != st && !m->profile;
    if (!tail_dev && sw != st) HIPCHK(hipEventRecord(m->dw_ev, sw));      // the last dW GEMM
    // EmbeddingLayer.backward (twice): entries sorted by row (side chain 0, started in forward),
    // per-key run reduce in batch order, fused updater
    const int64_t nnz = m->cur_nnz;
    if (s0 != st && dev_flags && s0_joined) {}           // (the last delta GEMM's launch held the join)
    else if (sort_dev_wait) PSCHK(launch_spin_until(m->start_flag + 1, m->sort_epoch, st, werr, 1));
    else if (s0 != st && dev_flags) PSCHK(launch_spin_until(m->start_flag + 5, m->s0_epoch, st, werr, 5));
    else if (s0 != st) HIPCHK(hipStreamWaitEvent(st, m->s0_ev, 0));  // the sort (forward), the stop flag and the wide update
    if (sl != s0) HIPCHK(hipStreamWaitEvent(st, m->loss_ev, 0));
    m->side0_pending = false;
    EmbBwdArgs g;
    memset(&g, 0, sizeof g);
    g.nnz = nnz; g.F = c.F; g.D = c.D; g.grad_mode = c.emb_grad_mode; g.apply = apply ? 1 : 0;
    g.sorted_key = m->sorted_keys; g.sorted_ent = m->sorted_ents; g.seg_start = m->seg_start; g.seg_id = m->seg_id;
    g.nseg = m->sh.active ? m->nseg_cur : m->nseg_dev;
    g.long_list = m->long_list_valid ? m->long_list : nullptr;
    // (the sharded step's and the field sort's lists hold every run above PS_EMB_SEQ_TILE entries: any threshold of at least a chunk will do)
    g.list_min = (m->sh.active || m->field_sorted) ? ((g_emb_list_min > 0 && g_emb_list_min < PS_EMB_SUPER_MIN) ? g_emb_list_min : PS_EMB_SUPER_MIN) : m->list_min;
    g.nlong = m->nlong_ptr;
    g.ftab = (m->long_list_valid && m->field_sorted) ? reinterpret_cast<const uint32_t *>(m->fs_pub + PS_FS_TAB_OFF(c.F)) : nullptr;
    g.out_slot = (m->sh.active && m->field_sorted) ? m->sh.slot : nullptr;     // (runs field by field, gradients in send order)
    g.ent_bag = (m->cur_offsets && m->sh.active) ? m->ent_bag : nullptr;     // fused path: sorted_ent already holds bags
    g.delta = m->dx; g.ldd = m->ldx; g.partials = m->partials; g.partials2 = m->partials2; g.W = s->emb.W; g.state = s->emb.state;
    // a key's run is at most B entries when single-hot: no second level (and no extra launch) up to 128 chunks
    g.long_runs = (m->cur_offsets != nullptr || (int64_t)B > (int64_t)PS_EMB_CHUNK * PS_EMB_SUPER_MIN) ? 1 : 0;
    // the reference's own summation order wherever the reference's input domain reaches (single-hot: n <= B)
    g.seq_order = (c.emb_sum_order == PS_SUM_SEQUENTIAL || (c.emb_sum_order == PS_SUM_AUTO && m->cur_offsets == nullptr)) ? 1 : 0;
    g.upd = emb_upd; g.fu = emb_fu;
    // The per-key gradients as handed to the updater are an OUTPUT only of the split form (backward, then update), of the sharded
    // step (they are what is pushed) and of a model that was asked to keep them (ps_model_set_keep_grads: parity tests).  The fused
    // step consumes a key's gradient in the registers it was reduced in: KVStore.sum's map is cleared by update
    // (store/KVStore.java:268-276), nothing of it outlives the step.  Writing it anyway cost nnz * 4 D bytes per step -- 1.07 GB
    // beside 7.5 GB of algorithmic traffic on the 320 M-row table (configs[3]: 0.61 -> 0.69 of 8 TB/s without it).
    m->grads_kept = m->keep_grads || !apply || m->sh.active;
    g.grads_out = m->grads_kept ? m->grads_out : nullptr; g.uniq_row = m->uniq_row; g.uniq_cnt = m->uniq_cnt; g.skip = skip;
    // tail_dev: the dense update, on side chain 1, starts when the embedding update has STARTED (its first workgroup
    // raises start_flag[2]; the update's workgroups check that flag themselves, no spinner launch in front of it), a
    // flag-setter launch behind it raises start_flag[3], and the embedding update's first workgroup ENDS only once that
    // flag is up (normally long since) -- no spinner launch at the end of the main chain either (tail_fused).  The flag
    // setter stays a launch: the update's writes (W, Wt of every layer) are released by the END of its kernel, and a
    // release from inside it would have to write back every XCD's L2 (tried: a per-workgroup agent-scope release made
    // the update 29 us instead of 7).
    const bool tail_fused = tail_dev && g_tail_fused && nnz > 0;
    LaunchOpts emb_lo;
    if (tail_dev) { if (++m->start_epoch == 0) ++m->start_epoch; emb_lo.flag = m->start_flag + 2; emb_lo.flag_val = m->start_epoch; }
    // (sharded step in overlap mode: the flat gradient this launch's companion writes is consumed on side chain 1 itself --
    //  all-reduce, replicated update -- so the training stream, which goes on to the push, does not wait for it)
    const bool tail_join = !(m->sh.active && m->sh.ov_mode == 1);
    // tail_defer (fused step): the join with the dense update is not held by this launch either -- it is handed to
    // whatever uses the store's stream next (ps_store.h pending_ev): the next step's gather runs beside the update's tail
    const bool tail_defer = tail_fused && tail_join && g_tail_defer && !m->sh.active && apply;
    if (tail_fused && tail_join && !tail_defer) { emb_lo.wait = m->start_flag + 3; emb_lo.wait_val = m->start_epoch; }
    { Prof pf(m, "emb_bwd_update"); PSCHK(launch_emb_bwd(g, st, &emb_lo, werr)); }
    m->emb_started_valid = tail_dev && emb_lo.launched; m->emb_started_epoch = m->start_epoch;
    if (tail_dev && !emb_lo.launched)       // nothing was launched (an empty batch): announce the start ourselves
        PSCHK(launch_flag_set(m->start_flag + 2, m->start_epoch, st));
    // The dense update runs LAST ON THE MAIN CHAIN.  A stream that reaches a wait before its event has fired resumes
    // 10-20 us after it (tools/gpu_timeline.py), a wait that is already satisfied costs ~3: on a side chain the update
    // ended within a few microseconds of the embedding update, so the next step's first GEMM -- which must wait for
    // it -- sometimes hit the slow case and sometimes not (a bimodal step, 174 / 192 us, fixed per process).  Here the
    // dW GEMMs it needs ended ~15 us earlier on their side chain, the delta GEMM that reads W_0 is in order before
    // it, and nothing crosses a stream at the step boundary.
    if (!tail_dev && sw != st) HIPCHK(hipStreamWaitEvent(st, m->dw_ev, 0));
    if (m->sh.active && c.kind == PS_MODEL_WIDEDEEP) {
        // sharded worker: the wide part of the flat buffer ([fc | wide G | wide C | bias]) is filled by the same launch
        WideUpdArgs &w = d.wide;
        w.rows = s->wide.rows; w.touched = s->wide.touched; w.gbar = m->gbar_dev;
        w.mode = c.wide_grad_mode == PS_GRAD_INTENDED ? 5 : 1;
        w.G = m->sh.flat + m->dense_elems; w.C = w.G + s->wide.rows;
        if (m->sh.slot_world) {         // (ps_shard_step: the wide part as per-worker slots, kernels_emb.h WideUpdArgs.slots)
            w.mode = 6; w.slots = m->sh.flat + m->dense_elems; w.slot_words = (int)m->sh.slot_words; w.world = m->sh.slot_world; w.rank = m->sh.slot_rank;
        }
        d.wide_blocks = wide_update_blocks(w);
    }
    if (tail_dev) {
        // (every waiter is enqueued after the launch that releases it)
        if (tail_fused && emb_lo.launched) {
            d.wait_flag = m->start_flag + 2; d.wait_val = m->start_epoch; d.bound = wait_bound(werr, 2);
            if (dw_split_done) { d.wait_flag2 = m->start_flag + 11; d.wait_val2 = dw_split_epoch; }
            if (tail_defer) { d.started_flag = m->start_flag + 12; d.started_val = m->start_epoch; }
            // sharded step in overlap mode: nothing on the training stream waits for this launch, and the NEXT step's gather
            // overwrites the first layer's input while the last dW GEMM, in front of this launch on side chain 1, may still
            // read it (at configs[2]'s size 25 us lie between the two; at toy sizes they met: one run in ten of the pipelined
            // schedule tests, round 4).  This launch's START -- that GEMM is done then -- is what the next gather's workgroups
            // wait for at theirs (enqueue_forward)
            if (!tail_join) { d.started_flag = m->start_flag + 12; d.started_val = m->start_epoch; m->sh.flat_start_epoch = m->start_epoch; m->sh.flat_start_valid = true; }
            { Prof pf(m, "dense_update"); PSCHK(launch_dense_update(d, sw)); }
            m->flat_stream = sw;
            if (tail_join) PSCHK(launch_flag_set(m->start_flag + 3, m->start_epoch, sw));
            if (tail_defer) {
                HIPCHK(hipEventRecord(m->tail_ev, sw));
                s->pending_ev = m->tail_ev; s->pending_flag = m->start_flag + 3; s->pending_val = m->start_epoch;
                s->pending_start = m->start_flag + 12;
            }
            return PS_OK;
        }
        PSCHK(launch_spin_until(m->start_flag + 2, m->start_epoch, sw, werr, 2));
        { Prof pf(m, "dense_update"); PSCHK(launch_dense_update(d, sw)); }
        m->flat_stream = sw;
        PSCHK(launch_flag_set(m->start_flag + 3, m->start_epoch, sw));
        PSCHK(launch_spin_until(m->start_flag + 3, m->start_epoch, st, werr, 3));
        return PS_OK;
    }
    { Prof pf(m, "dense_update"); PSCHK(launch_dense_update(d, st)); }
    m->flat_stream = st;
    return PS_OK;
}

// KVStore.update for the split form: apply the gradients left by backward(apply = false)
static int enqueue_update(ps_model *m) {
    ps_store *s = m->s;
    const ps_model_config_t &c = m->cfg;
    hipStream_t st = s->stream;
    const int nfc = c.nfc;
    ps_updater_t u;
    UpdParams emb_upd;
    FieldUpd emb_fu;
    bool emb_stateful = false;
    PSCHK(store_fill_field_upd(s, &emb_upd, &emb_fu, &emb_stateful));
    if (!s->emb.state && emb_stateful)                            // before anything is enqueued
        return ps_set_err(PS_E_STATE, "the embedding table was created weights-only (state_slots = 0): Adam / Ftrl cannot train it");
    DenseUpdArgs d;
    memset(&d, 0, sizeof d);
    d.nlayers = nfc; d.B = m->cur_B; d.apply = 1; d.skip = m->skip_dev; d.flat_grad = m->dense_grad_flat;
    PSCHK(store_resolve_updater(s, "fc0.weights", &u));
    d.upd = make_upd_params(u);
    int64_t off = 0;
    for (int l = 0; l < nfc; ++l) {
        FcParams &p = s->fc[l];
        DenseLayer &L = d.L[l];
        L.W = p.W; L.Wt = p.Wt; L.Wp = p.Wp; L.S1 = p.S1; L.S2 = p.S2;
        L.K = p.K; L.N = p.N; L.ldw = p.ldw; L.ldwt = p.Kpad;
        L.elem_begin = off; off += (int64_t)(p.K + 1) * p.N; L.elem_end = off;
    }
    PSCHK(launch_dense_update(d, st));
    if (c.kind == PS_MODEL_WIDEDEEP) {
        WideUpdArgs w;
        memset(&w, 0, sizeof w);
        w.rows = s->wide.rows; w.W = s->wide.W; w.state = s->wide.state; w.touched = s->wide.touched;
        w.bias = s->wide.bias; w.bias_state = s->wide.bias_state; w.gbar = m->gbar_dev; w.skip = m->skip_dev;
        PSCHK(store_resolve_updater(s, "wide.weights", &u));
        w.upd = make_upd_params(u);
        if (c.wide_grad_mode == PS_GRAD_INTENDED) PSCHK(enqueue_wide_intended(m, w, st));
        else PSCHK(launch_wide_update(w, st));
    }
    {
        // the per-key gradients are unique already: one "push" per key
        RowsApplyArgs r;
        memset(&r, 0, sizeof r);
        r.D = c.D; r.is_async = 0; r.identity = 1;
        r.sorted_key = m->uniq_row; r.nseg = m->nseg_dev; r.grads = m->grads_out;
        r.W = s->emb.W; r.state = s->emb.state; r.skip = m->skip_dev;
        r.upd = emb_upd; r.fu = emb_fu;
        PSCHK(launch_rows_apply(r, m->cur_nnz, st));
    }
    return PS_OK;
}

int finish_step(ps_model *m, float *loss) {
    ps_store *s = m->s;
    if (!loss) return PS_OK;
    HIPCHK(hipMemcpyAsync(loss, m->loss_dev, sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return store_check_bad_ids(s);
}

// One training step as a replayed hipGraph: the kernel arguments bake in the batch pointers,
// so there is one instantiated graph per (pointers, B, nnz); host batches are staged into the
// model's fixed device buffers first (outside the graph) and share one graph.
static int train_graph(ps_model *m) {
    ps_store *s = m->s;
    const void *sig[5] = {m->cur_ids, m->cur_offsets, m->cur_dense, m->cur_labels, m->cur_wide};
    for (auto &g : m->graphs)
        if (g.B == m->cur_B && g.nnz == m->cur_nnz && memcmp(g.sig, sig, sizeof sig) == 0) {
            HIPCHK(hipGraphLaunch(g.exec, s->stream));
            return PS_OK;
        }
    if (m->graphs.size() >= 64) {        // do not grow without bound on ever-changing pointers
        for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
        m->graphs.clear();
    }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_forward(m, true, true);
    if (rc == PS_OK) rc = enqueue_backward(m, true);
    hipError_t e = hipStreamEndCapture(s->stream, &graph);
    if (rc != PS_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return ps_set_err(PS_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    ps_model::GraphEntry ge;
    memcpy(ge.sig, sig, sizeof sig);
    ge.B = m->cur_B; ge.nnz = m->cur_nnz; ge.exec = nullptr;
    e = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return ps_set_err(PS_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    m->graphs.push_back(ge);
    HIPCHK(hipGraphLaunch(ge.exec, s->stream));
    return PS_OK;
}

extern "C" int ps_model_train(ps_model_t *m, const ps_batch_t *batch, float *loss) {
    RoctxRange roctx_range("ps_model_train");
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    HIPCHK(hipSetDevice(m->s->device));
    PSCHK(stage_batch(m, batch, true));
    if (m->cfg.use_graph && !m->profile) {
        PSCHK(train_graph(m));
    } else {
        PSCHK(enqueue_forward(m, true, true));
        PSCHK(enqueue_backward(m, true));
    }
    m->fwd_done = true; m->bwd_done = true;
    m->s->global_step++;                               // Context.step / PServer.globalStep
    return finish_step(m, loss);
}

extern "C" int ps_model_forward(ps_model_t *m, const ps_batch_t *batch, float *loss) {
    RoctxRange roctx_range("ps_model_forward");
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    HIPCHK(hipSetDevice(m->s->device));
    PSCHK(stage_batch(m, batch, true));
    PSCHK(enqueue_forward(m, true, false));
    m->fwd_done = true; m->bwd_done = false;
    return finish_step(m, loss);
}

extern "C" int ps_model_backward(ps_model_t *m) {
    RoctxRange roctx_range("ps_model_backward");
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    if (!m->fwd_done) return ps_set_err(PS_E_STATE, "backward before forward");
    PSCHK(store_enter(m->s));
    PSCHK(enqueue_backward(m, false));
    m->bwd_done = true;
    return PS_OK;
}

extern "C" int ps_model_update(ps_model_t *m) {
    RoctxRange roctx_range("ps_model_update");
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    if (!m->bwd_done) return ps_set_err(PS_E_STATE, "update before backward");
    PSCHK(store_enter(m->s));
    PSCHK(enqueue_update(m));
    m->s->global_step++;
    m->bwd_done = false;
    return PS_OK;
}

extern "C" int ps_model_predict(ps_model_t *m, const ps_batch_t *batch, float *p_out) {
    RoctxRange roctx_range("ps_model_predict");
    if (!m || !p_out || !batch) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(m->s));
    ps_batch_t b = *batch;
    b.labels = nullptr;
    PSCHK(stage_batch(m, &b, false));
    PSCHK(enqueue_forward(m, false, false));
    m->fwd_done = false;
    HIPCHK(hipMemcpyAsync(p_out, m->P, sizeof(float) * b.B, hipMemcpyDeviceToHost, m->s->stream));
    HIPCHK(hipStreamSynchronize(m->s->stream));
    return PS_OK;
}

extern "C" int ps_model_sync(ps_model_t *m) {
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    PSCHK(store_enter(m->s));
    HIPCHK(hipStreamSynchronize(m->s->stream));
    return store_check_bad_ids(m->s);       // ps_model_train(loss = NULL) never waits: out-of-range ids surface here
}

extern "C" int ps_model_last_loss(ps_model_t *m, float *loss) {
    if (!m || !loss) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(m->s));
    return finish_step(m, loss);
}

// ---------------------------------------------------------------------------
// intermediates for parity tests
// ---------------------------------------------------------------------------
static int copy_out_2d(ps_model *m, const float *src, int ld, int rows, int cols, float *out, int64_t cap, int *r, int *c) {
    if (r) *r = cols;     // reference orientation: features x B
    if (c) *c = rows;
    if (!out) return PS_OK;
    if (cap < (int64_t)rows * cols) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
    HIPCHK(hipMemcpy2DAsync(out, sizeof(float) * cols, src, sizeof(float) * ld, sizeof(float) * cols, rows,
                            hipMemcpyDeviceToHost, m->s->stream));
    HIPCHK(hipStreamSynchronize(m->s->stream));
    return PS_OK;
}

extern "C" int ps_model_get_act(ps_model_t *m, int layer, float *out, int64_t cap, int *rows, int *cols) {
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    PSCHK(store_enter(m->s));
    const ps_model_config_t &c = m->cfg;
    const int B = m->cur_B;
    if (layer == 0) return copy_out_2d(m, m->fc[0].A, m->fc[0].ldA, B, c.F * c.D, out, cap, rows, cols);
    if (layer == 1) return copy_out_2d(m, m->fc[0].A, m->fc[0].ldA, B, c.F * c.D + c.X, out, cap, rows, cols);
    const int l = layer - 2;
    if (l < 0 || l >= c.nfc) return ps_set_err(PS_E_BAD_ARG, "no such layer %d", layer);
    if (l + 1 < c.nfc) return copy_out_2d(m, m->fc[l + 1].A, m->fc[l + 1].ldA, B, c.fc_dims[l], out, cap, rows, cols);
    return copy_out_2d(m, m->out_last, m->ld_last, B, c.fc_dims[l], out, cap, rows, cols);
}

extern "C" int ps_model_get_delta(ps_model_t *m, int layer, float *out, int64_t cap, int *rows, int *cols) {
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    PSCHK(store_enter(m->s));
    const ps_model_config_t &c = m->cfg;
    const int B = m->cur_B, l = layer - 2;
    if (l < 0 || l >= c.nfc) return ps_set_err(PS_E_BAD_ARG, "no such layer %d", layer);
    if (l == 0) return copy_out_2d(m, m->dx, m->ldx, B, c.F * c.D, out, cap, rows, cols);
    return copy_out_2d(m, m->fc[l - 1].dOut, m->fc[l - 1].ldD, B, m->s->fc[l].K, out, cap, rows, cols);
}

extern "C" int ps_model_get_p(ps_model_t *m, float *out, int cap) {
    if (!m || !out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    if (cap < m->cur_B) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
    PSCHK(store_enter(m->s));
    HIPCHK(hipMemcpyAsync(out, m->P, sizeof(float) * m->cur_B, hipMemcpyDeviceToHost, m->s->stream));
    HIPCHK(hipStreamSynchronize(m->s->stream));
    return PS_OK;
}

extern "C" int ps_model_get_emb_grads(ps_model_t *m, int field, int64_t *ids_out, float *grads_out,
                                      int64_t cap_rows, int64_t *n_out) {
    if (!m || !n_out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ps_store *s = m->s;
    PSCHK(store_enter(s));
    if (!m->grads_kept)
        return ps_set_err(PS_E_BAD_ARG, "the last step was a fused training step of a model that does not keep its per-key gradients: "
                                        "call ps_model_set_keep_grads(m, 1) first, or use ps_model_forward / ps_model_backward");
    HIPCHK(hipStreamSynchronize(s->stream));
    uint32_t nseg = 0;
    HIPCHK(hipMemcpyAsync(&nseg, m->nseg_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<uint32_t> rows(nseg);
    if (nseg) {
        HIPCHK(hipMemcpyAsync(rows.data(), m->uniq_row, sizeof(uint32_t) * nseg, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    const EmbTables &e = s->emb;
    if (field < 0 || field >= e.F) return ps_set_err(PS_E_BAD_ARG, "no field %d", field);
    int64_t lo = 0, hi = 0;
    for (uint32_t i = 0; i < nseg; ++i) {
        if ((int64_t)rows[i] < e.row_base[field]) lo = i + 1;
        if ((int64_t)rows[i] < e.row_base[field + 1]) hi = i + 1;
    }
    const int64_t n = hi - lo;
    *n_out = n;
    if (!ids_out || !grads_out) return PS_OK;
    if (cap_rows < n) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
    for (int64_t i = 0; i < n; ++i)
        ids_out[i] = e.java_route() ? (int64_t)e.ids_local_h[(size_t)rows[lo + i]] : e.shard + ((int64_t)rows[lo + i] - e.row_base[field]) * e.nshards;
    if (n) {
        HIPCHK(hipMemcpyAsync(grads_out, m->grads_out + (size_t)lo * e.D, sizeof(float) * n * e.D, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return PS_OK;
}

extern "C" int ps_model_get_fc_grad(ps_model_t *m, int layer, int bias, float *out, int cap) {
    if (!m || !out) return ps_set_err(PS_E_BAD_ARG, "null argument");
    ps_store *s = m->s;
    if (layer < 0 || layer >= m->cfg.nfc) return ps_set_err(PS_E_BAD_ARG, "no such layer");
    PSCHK(store_enter(s));
    HIPCHK(hipStreamSynchronize(s->stream));
    int64_t off = 0;
    for (int l = 0; l < layer; ++l) off += (int64_t)(s->fc[l].K + 1) * s->fc[l].N;
    const FcParams &p = s->fc[layer];
    const int n = bias ? p.N : p.K * p.N;
    if (cap < n) return ps_set_err(PS_E_BAD_ARG, "buffer too small");
    if (bias) off += (int64_t)p.K * p.N;
    HIPCHK(hipMemcpyAsync(out, m->dense_grad_flat + off, sizeof(float) * n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PS_OK;
}

// ---------------------------------------------------------------------------
// measurement hooks
// ---------------------------------------------------------------------------
static int prof_collect(ps_model *m) {
    HIPCHK(hipStreamSynchronize(m->s->stream));
    for (auto &e : m->prof_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            auto &acc = m->prof_acc[e.name];
            acc.first += 1; acc.second += ms;
        }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    m->prof_events.clear();
    return PS_OK;
}

extern "C" int ps_model_set_keep_grads(ps_model_t *m, int on) {
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    PSCHK(store_enter(m->s));
    if (m->keep_grads != (on != 0)) {           // (a captured step has the choice baked in)
        HIPCHK(hipStreamSynchronize(m->s->stream));
        for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
        m->graphs.clear();
    }
    m->keep_grads = on != 0;
    return PS_OK;
}

extern "C" int ps_model_set_profile(ps_model_t *m, int enabled) {
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    PSCHK(store_enter(m->s));
    PSCHK(prof_collect(m));
    if (enabled) m->prof_acc.clear();
    m->profile = enabled != 0;
    m->prof_filter.clear();
    return PS_OK;
}

extern "C" int ps_model_set_profile_filter(ps_model_t *m, const char *group) {
    if (!m) return ps_set_err(PS_E_BAD_ARG, "model is NULL");
    PSCHK(store_enter(m->s));
    PSCHK(prof_collect(m));
    m->prof_acc.clear();
    m->prof_filter = group ? group : "";
    m->profile = true;
    return PS_OK;
}

extern "C" int ps_model_profile_report(ps_model_t *m, char *report, int cap) {
    if (!m || !report || cap <= 0) return ps_set_err(PS_E_BAD_ARG, "null argument");
    PSCHK(store_enter(m->s));
    PSCHK(prof_collect(m));
    std::string out;
    char line[160];
    for (auto &kv : m->prof_acc) {
        snprintf(line, sizeof line, "%s:%ld:%.6f;", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    snprintf(report, cap, "%s", out.c_str());
    return PS_OK;
}

extern "C" int ps_model_time_steps(ps_model_t *m, const ps_batch_t *batch, int steps, double *ms_out) {
    if (!m || !batch || !ms_out || steps <= 0) return ps_set_err(PS_E_BAD_ARG, "bad argument");
    ps_store *s = m->s;
    PSCHK(store_enter(s));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipEventRecord(a, s->stream));
    for (int i = 0; i < steps; ++i) {
        int r = ps_model_train(m, batch, nullptr);
        if (r != PS_OK) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); return r; }
        if (m->profile && m->prof_events.size() > 4096) PSCHK(prof_collect(m));
    }
    HIPCHK(hipEventRecord(b, s->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_out = ms;
    return store_check_bad_ids(s);
}
