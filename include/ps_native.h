/*
 * ps_native.h -- C ABI of libps_amd.so: the MI355X-native replacement for the
 * wudikua/ps training hot path (EmbeddingField lookup -> FcLayer WX+B
 * forward/backward -> Adam/FTRL sparse update -> key-sharded push/pull).
 *
 * This is the drop-in boundary.  The reference has no
 * FFI layer; its seams are four plain Java types, and every entry point
 * below names the reference interface it replaces
 * (paths relative to /root/reference/src/main/java/):
 *
 *   store.KVStore            store/KVStore.java:33-299     -> ps_store_*
 *   update.Updater           update/Updater.java:6-11      -> ps_updater_*
 *   layer.Layer (+ fields)   layer/Layer.java:12-78        -> ps_emb_* / ps_fc_* / ps_model_*
 *   net.PSClient/Router      net/PSClient.java:47-186,
 *                            net/PSRouterClient.java:55-151,
 *                            net/Mod.java:13-15            -> ps_router_* / ps_shard_*
 *
 * Conventions
 *  - every function returns an int status: PS_OK (== Resp.ec 200,
 *    net/PServer.java:28), PS_MISSING (204, net/PServer.java:84),
 *    PS_NO_UPDATER (500, net/PServer.java:172), or a negative PS_E_* code.
 *    Nothing throws or aborts; ps_last_error() gives the text.
 *  - plain pointers and sizes only.  Host float buffers use the reference's
 *    byte layout: a FloatMatrix "features x B" (column-major) IS a row-major
 *    [B][features] array; a weight "out x in" IS row-major [in][out].
 *  - pointers named *_dev are HIP device pointers on the store's device;
 *    everything else is host memory owned by the caller.
 *  - ids are int64 (the reference carries them as float and formats them
 *    with Float.toString -- exact only below 2^24; string keys such as
 *    "emF13.28305.0" are accepted by ps_store_get/put for parity).
 *  - one ps_store_t per GPU; calls on one store are serialised by the
 *    caller (as the reference serialises on the KVStore monitor,
 *    store/KVStore.java:109,136,192,202).
 */
#ifndef PS_NATIVE_H
#define PS_NATIVE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------- */
#define PS_OK 0            /* Resp.ec 200 */
#define PS_MISSING 204     /* key / row absent            (net/PServer.java:84) */
#define PS_NO_UPDATER 500  /* updater name unknown        (net/PServer.java:172) */
#define PS_E_BAD_ARG (-1)
#define PS_E_HIP (-2)      /* a HIP call failed / no device */
#define PS_E_UNSUPPORTED (-3)
#define PS_E_STATE (-4)    /* call order violated */

const char *ps_last_error(void);
/* library / build identification: "ps_amd <ver> gfx950 hip" */
const char *ps_version(void);
int ps_device_count(int *count);

/* ---- update.Updater -------------------------------------------------- */
enum { PS_UPD_ADAM = 0, PS_UPD_FTRL = 1, PS_UPD_SIMPLE = 2 };
typedef struct ps_updater {
    int kind;
    float alfa;                 /* Adam / Ftrl learning rate ("alfa" sic) */
    float beta1, beta2, epsilon;/* update/AdamUpdater.java:25-30          */
    float beta, l1, l2;         /* update/FtrlUpdater.java:25-30          */
    float eta;                  /* update/SimpleUpdater.java:11           */
} ps_updater_t;
/* model/DNN.java:95 : Adam(0.005, 0.9, 0.999, 1e-8) */
void ps_updater_default_adam(ps_updater_t *u);
/* model/WideDeepNN.java:109 : Ftrl(0.005, 1, 0.001, 0.001) */
void ps_updater_default_ftrl(ps_updater_t *u);
/* Updater.getName(): "adam@alfa:0.005@beta1:0.9@beta2:0.999@epsilon:1.0E-8@",
 * Ftrl (also prefixed adam@, update/FtrlUpdater.java:78-80)
 * "adam@alfa:0.005@beta:1.0@l1:0.001@l2:0.001@", "simple@eta:..@".
 * The PS looks updaters up by exactly this string (net/PServer.java:169). */
int ps_updater_name(const ps_updater_t *u, char *buf, int cap);
/* AdamUpdater(String) / FtrlUpdater(String) / SimpleUpdater(String).
 * Unknown name -> PS_NO_UPDATER. */
int ps_updater_from_name(const char *name, ps_updater_t *out);

/* ---- net.Router / net.Mod -------------------------------------------- */
enum { PS_ROUTE_ID_MOD = 0,      /* native default: id mod n                      */
       PS_ROUTE_JAVA_STRING = 1  /* floorMod(String.hashCode("emF<f>.<id>.0"), n) */ };
/* java.lang.String.hashCode (net/Mod.java:14) */
int32_t ps_java_string_hash(const char *key);
/* net/Mod.java:13-15 with the floorMod fix (Java % can go negative). */
int ps_router_shard_key(const char *key, int nshards);
/* shard of embedding id `id` of field `field` under `route_mode` */
int ps_router_shard_id(int route_mode, int field, int64_t id, int nshards);

/* ---- store.KVStore : one GPU-resident shard -------------------------- */
typedef struct ps_store ps_store_t;

/* KVStore.ins() for device `device`; `seed` drives the counter-based row
 * init that replaces util/MatrixUtil.java:62-74 (unseeded RandomUtils). */
int ps_store_create(int device, uint64_t seed, ps_store_t **out);
int ps_store_destroy(ps_store_t *s);
int ps_store_device(const ps_store_t *s);

/* Embedding tables "emF0".."emF<F-1>" (layer/EmbeddingLayer.java:50-57):
 * F per-field tables of rows[f] x D floats, rows initialised
 * +-U(0, 4*sqrt(6)/sqrt(1+D)) (layer/EmbeddingField.java:40-46) as a pure
 * function of (seed, field, id).  Updater state lives beside the rows.
 * shard/nshards: this store holds the ids with
 * ps_router_shard_id(route_mode, f, id, nshards) == shard, densely packed.
 * state_slots: 2 for Adam {M,V} / Ftrl {Z,N}, 0 = weights only (gather runs). */
int ps_store_create_embedding(ps_store_t *s, int F, const int64_t *rows, int D,
                              int state_slots, int shard, int nshards, int route_mode);
/* Wide table "wide.weights.<id>" + "wide.bias" (layer/LRLayer.java:37-52): zeros. */
int ps_store_create_wide(ps_store_t *s, int64_t wide_size);
/* Dense tensors "fc<i>.weights" (out x in) / "fc<i>.bias" (out x 1),
 * init +-U(0, 4*sqrt(6)/sqrt(in+out)) / (in+1)  (layer/FcLayer.java:34-50). */
int ps_store_create_fc(ps_store_t *s, int layer, int in_dims, int out_dims);

/* "default" / "wide.weights" / "wide.bias" / "emF" / "emF3." ... -> updater
 * (the Map<String,Updater> of model/DNN.java:33, looked up exact-key, then
 * prefix, then "default": store/KVStore.java:242-252).  Embedding rows
 * resolve per FIELD (the lookup of "emF<f>.") and, in front of that, per ROW:
 * a key that IS one row's key ("emF3.17.0", Float.toString of the id -- the
 * exact match KVStore.update(Map) tries first, store/KVStore.java:242) wins
 * over its field's updater.  Hard limits of this ABI (the reference's
 * HashMap has none; kernels_emb.h PS_EMB_UPD_GROUPS / PS_EMB_ROW_OVERRIDES):
 * 8 distinct updaters over the fields of a table group, 16 exact-key rows,
 * per-field resolution for at most 64 fields.  Beyond them, or for a key
 * that ends INSIDE an id ("emF1.3": by String.startsWith a prefix of
 * emF1.3.0, emF1.30.0, emF1.31.0 ...), every update of embedding rows
 * fails with PS_E_UNSUPPORTED. */
int ps_store_set_updater(ps_store_t *s, const char *key_or_prefix, const ps_updater_t *u);

/* KVStore.get(key) / put(key,val) by reference-style string key:
 *   "emF<f>.<id>.0" (D floats), "wide.weights.<id>.0" (1), "wide.bias" (1),
 *   "fc<i>.weights" (out*in, reference layout [in][out]), "fc<i>.bias" (out).
 * get: *len receives the element count; PS_MISSING when the key is not
 * held by this shard (store/KVStore.java:129-134 returns null). */
int ps_store_get(ps_store_t *s, const char *key, float *out, int cap, int *len);
int ps_store_put(ps_store_t *s, const char *key, const float *val, int len);
/* Bulk row access: KVStore.get/put for n ids of one field
 * (= PSClient.getList / updateList, net/PSClient.java:72-98,128-151).
 * which: 0 weights, 1/2 = updater state slot 0/1 ({M,V} or {Z,N}). */
int ps_store_get_rows(ps_store_t *s, int field, const int64_t *ids, int64_t n, int which, float *out);
int ps_store_put_rows(ps_store_t *s, int field, const int64_t *ids, int64_t n, int which, const float *val);
int ps_store_get_wide(ps_store_t *s, const int64_t *ids, int64_t n, int which, float *out);
int ps_store_put_wide(ps_store_t *s, const int64_t *ids, int64_t n, int which, const float *val);
/* AtomicLong globalStep (net/PServer.java:40): one per applied update round. */
int64_t ps_store_global_step(const ps_store_t *s);
int ps_store_advance_global_step(ps_store_t *s, int64_t by);
/* The PS server's side of the wire, by string key (what the gRPC facade
 * ps_amd/ps_server.py binds):
 *   PServer.push(key, gradient, isAsync, updaterKey)  net/PServer.java:164-195
 *   PServer.psUpdate()                                net/PServer.java:197-214
 * n messages (keys[i], grads[i][lens[i]]) in ARRIVAL order, any of the key
 * kinds of ps_store_get.  is_async = 0: one BSP round -- every key's pushes
 * are summed in arrival order (KVStore.sum: addi), divided by their count
 * (KVStore.update: divi(sumCnt)) and given to the key's updater once;
 * is_async = 1: one updater step per message, in order.  All arithmetic on
 * the device (the kernels of the hot path); the updater of a key is the one
 * ps_store_set_updater resolves for it.  Does not touch globalStep. */
/* floats a push / upsert of `key` must carry on this shard; PS_MISSING for a key the store does not hold (the facade
 * validates a BSP push when it arrives, net/PServer.java:164-175, not when the round's last barrier applies it).
 * ps_store_push_update itself validates EVERY message before it touches the store: a bad one fails the call whole. */
int ps_store_key_length(const ps_store_t *s, const char *key, int *len_out);
int ps_store_push_update(ps_store_t *s, int n, const char *const *keys, const float *const *grads,
                         const int *lens, int is_async);
/* bytes of HBM held by the store */
int64_t ps_store_bytes(const ps_store_t *s);

/* ---- layer.Layer : the model graph on one GPU ------------------------- */
typedef struct ps_model ps_model_t;
enum { PS_MODEL_DNN = 0,       /* model/DNN.java:92-128            */
       PS_MODEL_WIDEDEEP = 1   /* model/WideDeepNN.java:105-161    */ };
enum { PS_GRAD_COMPAT = 0,     /* reference arithmetic incl. the double-backward
                                  factor (n+1)/(2n^2) and the
                                  "every wide key ever seen" gradient (A.10)   */
       PS_GRAD_INTENDED = 1    /* mean over occurrences / per-key presence sum */ };
enum { PS_ACT_NONE = 0, PS_ACT_RELU = 1, PS_ACT_SIGMOID = 2 };

typedef struct ps_model_config {
    int kind;            /* PS_MODEL_*                                        */
    int F, D, X;         /* embeddingFieldNum, embeddingSize, numberFieldNum  */
    int nfc;             /* fcLayerDims.length (<= 8)                         */
    int fc_dims[8];      /* e.g. {512,256,1}; last layer sigmoid (DNN) / none */
    int64_t wide_size;   /* CTR.wideSize (CTR.java:35); 0 for DNN            */
    int max_batch;       /* largest B ever passed                             */
    int64_t max_nnz;     /* largest number of ids per batch (B*F if single-hot) */
    int emb_grad_mode;   /* PS_GRAD_*                                         */
    int wide_grad_mode;  /* PS_GRAD_*                                         */
    int use_graph;       /* 1: replay the step as a hipGraph.  A graph bakes the batch's device pointers in: one instantiated graph per
                            (pointers, B, nnz), at most 64 kept -- host batches (staged into the model's own buffers) and a few
                            recycled device batches replay; a stream of ever-new device batches re-instantiates every step (~20 ms).
                            Measured slower than the eager multi-stream step on MI355X: off by default. */
    int emb_sum_order;   /* PS_SUM_*: order in which a key's per-sample
                            gradients are added (layer/EmbeddingField.java:86-104) */
} ps_model_config_t;
/* EmbeddingField.backward adds a key's per-sample gradients strictly in
 * sample order.  PS_SUM_SEQUENTIAL reproduces that order for every key
 * (bit-exact with the oracle's orc_emb_geff(chunk = 0)); a key seen n times
 * costs a chain of n (compat mode: 2n) dependent f32 adds on one wave.
 * PS_SUM_CHUNKED adds runs above 32 entries as 32-entry chunks folded in two
 * levels (orc_emb_geff(chunk = 32)): same sum, different rounding, no long
 * chain.  PS_SUM_AUTO: sequential for single-hot batches (everything the
 * reference can express: n <= B), chunked for multi-hot bags, where one key
 * can occur tens of thousands of times per step. */
enum { PS_SUM_AUTO = 0, PS_SUM_SEQUENTIAL = 1, PS_SUM_CHUNKED = 2 };

/* One minibatch.  Single-hot (the reference): offsets == NULL and ids is
 * [B][F] (the bytes of the F x B matrix "E", CTR.java:47-68).  Multi-hot:
 * offsets[B*F+1] is a CSR over bags in (sample, field) order, sum pooling.
 * on_device != 0: every pointer is a device pointer (inputs resident in HBM).
 * A device batch must be COMPLETE when it is handed over (written by a
 * host-synchronous copy such as ps_dev_upload, or by ps_ingest_next, which
 * synchronises its own copies) and stay unchanged until the step that uses
 * it has run: the step reads its ids on side streams that are not ordered
 * behind earlier work of the store's stream (the sharded step's next plan
 * since round 3; the multi-hot step's key/sort kernels since round 5). */
typedef struct ps_batch {
    int B;
    const int64_t *ids;       /* nnz ids, bag-major                          */
    const int64_t *offsets;   /* NULL or [B*F+1]                             */
    const float *dense;       /* [B][X]   ("X")                              */
    const float *labels;      /* [B]      ("Y"); NULL for predict            */
    const int64_t *wide_ids;  /* [B][F]   ("W" = E mod wideSize); WideDeep   */
    int on_device;
    int64_t nnz;              /* multi-hot + on_device: the id count (= offsets[B*F]) when the caller knows it;
                                 0 = read it back from the device (one host wait in front of the step)          */
} ps_batch_t;

/* DNN.buildModel / WideDeepNN.buildModel over the store's parameters
 * (creates any of them that do not exist yet, as the lazy
 * kvStore.get(key, init) does). */
int ps_model_create(ps_store_t *s, const ps_model_config_t *cfg, ps_model_t **out);
int ps_model_destroy(ps_model_t *m);

/* One training step for thread = 1:
 *   TrainerThread.call (train/TrainerThread.java:29-39): pullWeights + Model.train
 *   Trainer.train tail (train/Trainer.java:90-100): KVStore.update(updaters), clear, step++
 * Asynchronous on the store's stream; *loss is written when loss != NULL
 * (that forces a sync).  Training stops backward when loss <= 0.01 or NaN
 * (model/DNN.java:58-63), exactly as the reference. */
int ps_model_train(ps_model_t *m, const ps_batch_t *batch, float *loss);
/* Split form of the same step, for hosts that drive Layer.forward /
 * Layer.backward / KVStore.update themselves (and for the sharded path):
 *   forward  = layers forward + loss            (model/DNN.java:44-49)
 *   backward = loss.backward + layers backward  (model/DNN.java:50,64-68);
 *              gradients are left in the store's pending set (= KVStore.sum)
 *   update   = KVStore.update(Map) + clear      (store/KVStore.java:240-277) */
int ps_model_forward(ps_model_t *m, const ps_batch_t *batch, float *loss);
int ps_model_backward(ps_model_t *m);
int ps_model_update(ps_model_t *m);
/* Model.predict (model/DNN.java:78-90): forward only, P[B] to host. */
int ps_model_predict(ps_model_t *m, const ps_batch_t *batch, float *p_out);
int ps_model_sync(ps_model_t *m);
/* last loss computed on device (syncs) */
int ps_model_last_loss(ps_model_t *m, float *loss);

/* Intermediates of the last step, copied to host in the reference layout
 * (for parity tests: Layer.A / Layer.delta, layer/Layer.java:16-18).
 * layer: 0 embedding A [B][F*D]; 1 concat A [B][F*D+X]; 2+i fc_i A [B][out_i].
 * delta: 2+i = fc_i.delta = W^T delta [B][in_i] (layer/FcLayer.java:108);
 *        for i = 0 the embedding columns are already masked by relu'. */
int ps_model_get_act(ps_model_t *m, int layer, float *out, int64_t cap, int *rows, int *cols);
int ps_model_get_delta(ps_model_t *m, int layer, float *out, int64_t cap, int *rows, int *cols);
int ps_model_get_p(ps_model_t *m, float *out, int cap);
/* Per-key gradient handed to the updater in the last step (after /cnt):
 * unique embedding rows of field `field`: ids_out[n], grads_out[n][D];
 * n_out receives the count (call with NULL outputs to size). */
int ps_model_get_emb_grads(ps_model_t *m, int field, int64_t *ids_out, float *grads_out,
                           int64_t cap_rows, int64_t *n_out);
/* The gradients above exist after ps_model_backward (the split form) and after a sharded step (they are what is pushed).  A FUSED
 * training step (ps_model_train) consumes every key's gradient in the registers it was reduced in -- KVStore.sum's map does not
 * outlive update either (store/KVStore.java:268-276) -- unless the model was asked to keep them: on = 1 makes ps_model_train write
 * them as well (nnz x D floats more per step), for parity tests.  Default 0; ps_model_get_emb_grads after a fused step of a model
 * that does not keep them returns PS_E_BAD_ARG. */
int ps_model_set_keep_grads(ps_model_t *m, int on);
/* dense gradient of "fc<i>.weights" ([in][out]) / "fc<i>.bias" as given to the updater */
int ps_model_get_fc_grad(ps_model_t *m, int layer, int bias, float *out, int cap);

/* ---- stand-alone hot-path operators (what a GpuEmbeddingLayer /
 *      GpuFcLayer JNI shim binds one-to-one) ------------------------------ */
/* Device buffers for hosts without their own HIP binding (the JNI shim keeps
 * these as long handles). */
int ps_dev_alloc(ps_store_t *s, size_t bytes, void **out_dev);
int ps_dev_free(ps_store_t *s, void *p_dev);
int ps_dev_upload(ps_store_t *s, void *dst_dev, const void *src_host, size_t bytes);
int ps_dev_download(ps_store_t *s, void *dst_host, const void *src_dev, size_t bytes);
/* EmbeddingLayer.forward (layer/EmbeddingLayer.java:25-48) for all fields:
 * out_dev[b][f*D + d] = act(sum over the bag of table_f[id][d]); ld = row
 * stride of out_dev in floats.  ids/offsets as in ps_batch_t (device). */
int ps_emb_forward(ps_store_t *s, const int64_t *ids_dev, const int64_t *offsets_dev,
                   int B, int act, float *out_dev, int ld);
/* FcLayer.forward (layer/FcLayer.java:74-91): y = act(W x + b).
 * x_dev [B][ldx]: ldx must be a multiple of 16 with ldx > in and
 * x_dev[b][in] == 1.0 (the bias rides in the GEMM as a ones column; columns
 * in+1..ldx-1 zero).  y_dev [B][ldy], ldy >= out. */
int ps_fc_forward(ps_store_t *s, int layer, int act, const float *x_dev, int ldx,
                  int B, float *y_dev, int ldy);
/* FcLayer.backward (layer/FcLayer.java:93-110) of layer `layer`, stand-alone:
 *   delta_dev [B][ldd], ldd = out rounded up to 16, padding columns zero: on
 *     entry the gradient wrt the layer's output (next.delta, or the loss
 *     gradient when the layer is last, :95-99); act' is applied IN PLACE from
 *     y_dev, the layer's output as ps_fc_forward left it (:100-102; y_dev may
 *     be NULL with PS_ACT_NONE);
 *   x_dev [B][ldx]: the layer's input exactly as given to ps_fc_forward;
 *   biasGradient = rowMeans(delta) and weightsGradient = delta * x^T / B are
 *     ADDED to the store's pending gradient of "fc<i>.bias" / "fc<i>.weights"
 *     (kvStore.sum, :104,:106) -- ps_dense_update applies and clears them;
 *   dx_dev [B][lddx] (NULL: skip): weights^T * delta (:108), the layer's
 *     `delta` field = the next.delta of the layer below (no mask applied:
 *     that layer's backward applies its own act').
 * ps_fc_pending_grad reads the pending SUM ([in][out] / [out]) and its count. */
int ps_fc_backward(ps_store_t *s, int layer, int act, const float *x_dev, int ldx,
                   const float *y_dev, int ldy, float *delta_dev, int ldd, int B,
                   float *dx_dev, int lddx);
int ps_fc_pending_grad(ps_store_t *s, int layer, int bias, float *out, int cap, int *count);
/* KVStore.update(Map<String,Updater>) + clear for the dense tensors
 * (store/KVStore.java:240-277): g = pending sum / count, the updater resolved
 * for "fc<i>.weights" (exact key, prefix, "default"), W' and its transpose
 * rewritten, pending cleared, globalStep++.  layer < 0: every layer with a
 * pending gradient; a named layer without one: PS_MISSING. */
int ps_dense_update(ps_store_t *s, int layer);
/* EmbeddingLayer.backward (layer/EmbeddingLayer.java:59-69) =
 * EmbeddingField.backward of every field (layer/EmbeddingField.java:86-104),
 * run TWICE per step by the reference (grad_mode
 * PS_GRAD_COMPAT reproduces the resulting factor (n+1)/(2n^2), PS_GRAD_INTENDED
 * is the mean), + KVStore.sum + KVStore.update fused (apply != 0: the "emF"
 * updater runs on every touched row in place; apply == 0: gradients only).
 *   ids_dev / offsets_dev / nnz as in ps_batch_t (device); a_dev [B][lda] the
 *   layer's output (relu' mask, act = PS_ACT_RELU; NULL with PS_ACT_NONE);
 *   delta_dev [B][ldd] the gradient wrt that output (columns f*D..f*D+D-1);
 *   sum_order PS_SUM_* (see ps_model_config_t.emb_sum_order).
 * ps_emb_last_grads: the per-key gradients it handed to the updater -- n unique
 * table rows (row = first row of the field + id, fields back to back) and
 * [n][D] gradients; call with NULL outputs to size. */
int ps_emb_backward_update(ps_store_t *s, const int64_t *ids_dev, const int64_t *offsets_dev,
                           int64_t nnz, int B, int act, const float *a_dev, int lda,
                           const float *delta_dev, int ldd, int grad_mode, int sum_order, int apply);
int ps_emb_last_grads(ps_store_t *s, int64_t *rows_out, float *grads_out, int64_t cap_rows, int64_t *n_out);
int ps_store_sync(ps_store_t *s);
/* Host wait for one HIP stream (NULL = the store's): for ps_comm_ops_t callbacks that stage through the host. */
int ps_stream_sync(ps_store_t *s, void *hip_stream);

/* ---- data.LibsvmParser + CTR.parseFeature + DataSource/DataSet -----------
 * The step in FRONT of the hot path: libsvm text -> the
 * arrays Model.train consumes.  One sample per line:
 *   "<label> <idx>:<v> ... "   data/LibsvmParser.java:13-25
 * columns 1..F give the sparse ids (the idx; CTR.java:55-57), columns
 * F+1..F+X the dense values (CTR.java:58-60), W = E mod wideSize
 * (MatrixUtil.hash, util/MatrixUtil.java:27-33).  ids_via_float = 1 keeps the
 * reference's long -> float -> id path (exact below 2^24); 0 keeps int64.
 * offset/step: this reader takes the RAW lines offset, offset+step, ... (blank
 * ones counted, as DataSource.readLine does), then drops the blank ones
 * (data/DataSource.java:25-46 worker sharding).  Blank lines are skipped
 * (LibsvmParser returns an empty list); runs of spaces are tolerated. */
typedef struct ps_ingest_config {
    int F, X;            /* sparse fields, dense features                     */
    int batch;           /* samples per batch (CTR.java:84: 1000)             */
    int threads;         /* host parser threads                               */
    int offset, step;    /* DataSource offset / step                          */
    int ids_via_float;   /* 1 = reference compat                              */
    int64_t wide_size;   /* 0: no W                                           */
} ps_ingest_config_t;
typedef struct ps_ingest ps_ingest_t;
/* Host-only (no GPU): number of lines of this reader; parse lines
 * [first_line, first_line+max_lines) of this reader into caller arrays
 * ids[n][F] (int64), dense[n][X], labels[n], wide_ids[n][F] (may be NULL). */
int ps_libsvm_count(const char *text, size_t len, int offset, int step, int64_t *n_lines);
int ps_libsvm_parse(const char *text, size_t len, const ps_ingest_config_t *cfg, int64_t first_line,
                    int64_t max_lines, int64_t *ids, float *dense, float *labels, int64_t *wide_ids,
                    int64_t *n_parsed);
/* The pipeline (= DataSet's reader threads + queue): batch k+1 is parsed by
 * the host pool into pinned memory and copied to HBM on its own stream while
 * batch k trains.  ps_ingest_next returns device pointers (on_device = 1),
 * valid until the call after the next one; PS_MISSING at the end of the data
 * (FileSource returns null); ps_ingest_reset rewinds (DataSource.reset). */
int ps_ingest_create(ps_store_t *s, const ps_ingest_config_t *cfg, ps_ingest_t **out);
int ps_ingest_destroy(ps_ingest_t *g);
int ps_ingest_open_file(ps_ingest_t *g, const char *path);
int ps_ingest_open_memory(ps_ingest_t *g, const char *text, size_t len);
int ps_ingest_lines(ps_ingest_t *g, int64_t *n_lines);
int ps_ingest_next(ps_ingest_t *g, ps_batch_t *out);
int ps_ingest_reset(ps_ingest_t *g);
int ps_ingest_stats(ps_ingest_t *g, double *parse_seconds, int64_t *lines, int64_t *bytes);
/* CTR.java:84-100's loop (dataSet.next -> trainer.train until the source is dry) for one reader and one model: up to
 * max_batches batches (< 0: to the end of the data) trained as they arrive, no host wait in between; *trained = their
 * number.  PS_OK at the end of the data too. */
int ps_ingest_train(ps_ingest_t *g, ps_model_t *m, int64_t max_batches, int64_t *trained);

/* ---- shard checkpoint / resume (absent in the reference) --
 * One file per shard: the store's arrays raw (embedding rows + updater state
 * of THIS shard, the wide table, every FC tensor + state, the registered
 * updaters, globalStep).  ps_store_load needs a store of the same geometry
 * (create the tables / the model first); resuming is exact. */
int ps_store_save(ps_store_t *s, const char *path);
int ps_store_load(ps_store_t *s, const char *path);

/* ---- evaluate.AUC (evaluate/AUC.java:32-82; train/Trainer.java:44-68) ------
 * AUC of predictions p[n] against labels y[n] (y > 0 = positive) exactly as
 * AUC.calculate() defines it: stable ascending sort by p, walk from the top,
 * add (x - prev) * y at every negative -- i.e. the fraction of (positive,
 * negative) pairs ranked correctly, ties resolved by input order.  The pair
 * count is computed exactly on the device (stable radix sort + prefix
 * counts) and divided once in double.  on_device: p, y are device pointers.
 * No negatives -> 0.0, no positives -> NaN (what the reference's arithmetic
 * yields). */
int ps_auc_compute(ps_store_t *s, const float *p, const float *y, int64_t n, int on_device,
                   double *auc, int64_t *pos_num, int64_t *neg_num);

/* ---- net.PSClient / PSRouterClient / PServer over a sharded store ------
 * One ps_store_t per GPU holds the embedding rows with id mod N == shard
 * (net/Mod.java routing, PS_ROUTE_ID_MOD); dense FC tensors and the wide
 * table are replicated and kept identical by an all-reduce.  These are the
 * device-side halves of the exchange; the host moves the buffers between
 * ranks (RCCL all-to-all-v / all-reduce over xGMI: torch.distributed in
 * bench.py, any binding in a Java host).  Per step and rank:
 *
 *   ps_shard_plan            = PSRouterClient.getList fan-out  (net/PSRouterClient.java:60-85)
 *     all-to-all counts, all-to-all-v send_rows -> recv_rows
 *   ps_shard_serve_pull      = PServer.getList                 (net/PServer.java:102-117)
 *     all-to-all-v rows back  -> cache [U][D]  (the worker cache, store/KVStore.java:96)
 *   ps_shard_forward_backward= Model.train on the cached rows  (model/DNN.java:35-70)
 *   ps_shard_grads           = the per-key gradients PSClient.push sends (net/PSClient.java:154-174)
 *     all-to-all-v grads      -> recv_grads (same order as recv_rows)
 *   ps_shard_apply_push      = PServer.push + barrier + psUpdate (net/PServer.java:164-283)
 *   ps_shard_flat_grad / all-reduce(sum) / ps_shard_apply_flat
 *                            = push + psUpdate of every dense tensor and wide key
 *
 * The step never stops early on loss <= 0.01 in this mode (a worker that
 * skipped its backward would stall the collective). */
/* Adopt the host framework's HIP stream (hipStream_t) for everything this
 * store enqueues, so kernels and the host's collectives are stream-ordered. */
int ps_store_set_stream(ps_store_t *s, void *hip_stream);
/* Worker: stage the batch, find its unique (field,id) keys grouped by owner
 * shard.  counts_out[nshards] = keys per owner; *send_rows_dev = the
 * owner-local row of every unique key (uint32, owner-major, ascending row).
 * Nothing here depends on the weights, so the host may run it ahead of time:
 * `hip_stream` (NULL = the store's stream) is the stream the plan kernels run
 * on and the one this call synchronises to read the counts back.  A second
 * ps_model_t on the same store gives the plan of step t+1 its own buffers
 * while step t still trains (ps_amd/sharded.py alternates two models). */
int ps_shard_plan(ps_model_t *m, const ps_batch_t *batch, int nshards, void *hip_stream,
                  int64_t *counts_out, uint32_t **send_rows_dev, int64_t *n_unique);
/* The same in two halves: _launch only enqueues (no host wait), _finish waits
 * for that plan's event and returns the counts -- so the host can enqueue a
 * whole training step between the two. */
int ps_shard_plan_launch(ps_model_t *m, const ps_batch_t *batch, int nshards, void *hip_stream);
int ps_shard_plan_finish(ps_model_t *m, int64_t *counts_out, uint32_t **send_rows_dev, int64_t *n_unique);
/* Owner: rows_out_dev[i][0..D) = weights of local row rows_dev[i]. */
int ps_shard_serve_pull(ps_store_t *s, const uint32_t *rows_dev, int64_t n, float *rows_out_dev);
/* Worker: forward + loss + backward reading embedding rows from cache_dev
 * [n_unique][D] (order of send_rows).  Leaves the per-key gradients, the
 * flat dense/wide gradient, and writes *loss when non-NULL (syncs). */
int ps_shard_forward_backward(ps_model_t *m, const float *cache_dev, float *loss);
/* Worker: *grads_dev = [n_unique][D] gradients in the order of send_rows
 * (each already carries this worker's double-backward factor, App. A.6). */
int ps_shard_grads(ps_model_t *m, float **grads_dev, int64_t *n_unique);
/* Owner: rows_dev[n] / grads_dev[n][D] as received from all workers
 * (worker-major).  BSP (is_async = 0): mean over the workers that pushed a
 * key, one updater step per key.  Async (-DisPsAsync=1,
 * net/PServer.java:176-184): every push applied on its own, in arrival
 * (= worker) order.  globalStep++ either way.  peer_counts[npeers] (host) =
 * entries received from every worker, in buffer order; with it (and rows
 * unique inside one worker's range, as ps_shard_plan sends them) the update
 * needs no sort.  NULL: any order / duplicates allowed, stable sort by row. */
int ps_shard_apply_push(ps_store_t *s, const uint32_t *rows_dev, const float *grads_dev,
                        int64_t n, const int64_t *peer_counts, int npeers, int is_async);
/* ---- the same exchange driven by the library: ONE call per step ------------
 * ps_shard_step = plan, the fixed-size exchange of the key lists with their
 * counts, all-to-all-v rows / gradients, the dense + wide all-reduce, owner
 * push + updater, replicated update -- everything above, enqueued from C.  The collectives are reached through a table of callbacks:
 * ps_comm_rccl_create fills it with RCCL (ncclSend/ncclRecv groups,
 * ncclAllGather, ncclAllReduce over xGMI; librccl is dlopen'ed on first use;
 * rank 0 makes the ids (3 x 128 bytes) with ps_comm_rccl_unique_id and the host
 * hands it to every rank); a host may plug in its own.  All pointers are
 * device pointers; every callback enqueues on `stream` (hipStream_t).
 * counts are in elements of elem_bytes, ordered by peer rank. */
typedef struct ps_comm_ops {
    void *ctx;
    int nranks, rank;
    int (*all_gather)(void *ctx, const void *send, void *recv, size_t bytes_per_rank, void *stream);
    int (*all_to_all_v)(void *ctx, const void *send, const int64_t *send_counts, void *recv,
                        const int64_t *recv_counts, size_t elem_bytes, void *stream);
    int (*all_reduce_sum_f32)(void *ctx, float *buf, int64_t n, void *stream);
    int flags;          /* PS_COMM_* bits; 0 for a table that moves every part of every exchange */
} ps_comm_ops_t;
/* flags of a plugged-in table.  PS_COMM_OWN_IN_PLACE: the step reads this rank's OWN part of the rows and gradient
 * exchanges where it was produced (its own packed key list, the owner-side gather's output, its gradient buffer) and
 * never looks at the self part of the receive buffer -- the table may leave that part unwritten.  The RCCL table made
 * by ps_comm_rccl_create always works this way (a rank's own keys never touch the wire). */
#define PS_COMM_OWN_IN_PLACE 1
/* At most PS_COMM_MAX_RANKS ranks in one table (a hard limit of this ABI: the owner-side push keeps one bit per
 * pushing worker and row; net/PServer.java has no such limit, its workers are gRPC clients). */
#define PS_COMM_MAX_RANKS 32
int ps_comm_rccl_unique_id(char *out384);   /* three 128-byte ids: the main, the key-list and the all-reduce communicator */
/* nranks == 1: no wire is needed and none is loaded (every collective is a device copy) -- unless
 * ps_tune_set("rccl_force", 1 | 2) (or PS_RCCL_FORCE in the environment) asks for it: then librccl is loaded, the three
 * 1-rank communicators are created (id384 may be NULL: the ids are made here) and EVERY collective of the step goes
 * through RCCL on its own stream -- ncclAllGather, grouped ncclSend / ncclRecv to this rank itself, ncclAllReduce.
 * 1: everything the step reads came off the wire (own keys too); 2: the wire runs and a rank's own keys are still read
 * in place, as at N > 1.  This is how the wire is exercised on a one-GPU box (tests/test_gpu_rccl_wire.py). */
int ps_comm_rccl_create(ps_store_t *s, int nranks, int rank, const char *id384, ps_comm_ops_t *out);
int ps_comm_rccl_destroy(ps_comm_ops_t *ops);
/* RCCL's own view of a table made by ps_comm_rccl_create: ncclCommCount / ncclCommUserRank of the main communicator
 * and whether the side communicator exists (what bench.py reports as evidence of the wire a multi-GPU run used). */
int ps_comm_rccl_info(const ps_comm_ops_t *ops, int *comm_count, int *user_rank, int *has_side);
/* RCCL operations the table has issued so far: out5[0] ncclAllGather, [1] grouped ncclSend/ncclRecv exchanges, [2]
 * ncclAllReduce, [3] single ncclSend + ncclRecv calls, [4] 1 when the table reaches a wire (N > 1 or rccl_force). */
int ps_comm_rccl_calls(const ps_comm_ops_t *ops, int64_t *out5);
/* One-shot wire check (collective: every rank calls it): each callback of the
 * table once on known patterns -- uneven all-to-all-v counts, all-gather slot
 * order, a float all-reduce -- verified on the host.  PS_E_STATE names the
 * first wrong word.  bench.py runs it before the first timed step. */
int ps_comm_selfcheck(ps_store_t *s, const ps_comm_ops_t *comm);
/* MAPPED PEER (ps_tune_set("mapped_peer", 1), or PS_MAPPED_PEER=1 in the environment; off by default): the two exchanges on
 * the step's critical chain -- rows back (PServer.getList's reply, net/PServer.java:102-117) and gradients out (PSClient.push,
 * net/PSClient.java:154-174) -- do not go through the table's all_to_all_v but through the peers' memory, mapped with
 * hipIpcOpenMemHandle at the model's first ps_shard_step_begin (the handles travel through the table's all_gather; every rank
 * must have asked for it and every mapping must succeed on every rank, otherwise all ranks stay on all_to_all_v): one launch
 * per exchange stores every peer's part into that peer's buffer, raises this rank's flag there and waits -- bounded -- for the
 * peers' flags here.  The id blocks and the all-reduce still use the table.  A wait that runs into its bound is reported like
 * every other device-side wait: the next host-side check returns PS_E_STATE (a peer died: restart the ranks without the knob).
 * Needs D % 4 == 0 and the sort-free owner push.  2: a 1-rank table too (its own part through the same launch; measurement).
 * The set-up ends with a wire check of its own (three rounds of both exchanges on a pattern that names round, sender, receiver, row
 * and column, verified by a kernel of the receiving rank): a wrong word on any rank and every rank stays on all_to_all_v.
 * ps_shard_mapped_info: out5[0] 1 when this model's exchanges use it (-1: the wire check failed), [1] a rank's own part too, [2] / [3] launches so far
 * (rows, gradients), [4] 1 when the flag words are fine-grained memory. */
int ps_shard_mapped_info(const ps_model_t *m, int64_t *out5);
int ps_shard_step(ps_model_t *m, const ps_batch_t *batch, const ps_comm_ops_t *comm, int is_async, float *loss);
/* The step in two halves.  _begin enqueues what reads no weight without a host wait: the plan, the exchange of the key
 * lists -- FIXED-SIZE blocks [count | overflow flag | owner-local rows | padding], one per peer, so the exchange needs
 * no split sizes and carries the counts of the two exchanges that do (rows back, gradients out) -- and the publication
 * of those counts to pinned host memory.  A wire block holds 2 * max_nnz / nranks rows (ps_tune_set("blk_factor")): a
 * worker whose list for some owner is longer raises the overflow flag in every block it sends, every rank sees every
 * flag, and _finish of that step first exchanges the full-size blocks (max_nnz rows: what every step sent before
 * round 4) -- same decision on every rank, no host round trip to find out.  The very first _begin of a model
 * all-gathers [block sizes | stream-join mode] once and fails with PS_E_BAD_ARG on every rank when the ranks were
 * configured differently (max_batch / max_nnz), instead of posting mismatched receives; use_side = 1 runs it on the store's prefetch stream so that step t+1 can begin -- on
 * another model of the same store -- before step t finishes.  _finish waits for the counts (the step's one host wait)
 * and enqueues the rest: owner-side gather, rows back, forward / backward, gradients out, owner update, the flat
 * reduction and the replicated update.  A rank's own keys are read where they are, never copied. */
int ps_shard_step_begin(ps_model_t *m, const ps_batch_t *batch, const ps_comm_ops_t *comm, int use_side);
int ps_shard_step_finish(ps_model_t *m, const ps_comm_ops_t *comm, int is_async, float *loss);
/* _finish of step t with _begin of step t+1 (next_batch; NULL = plain _finish) issued between "gradients ready"
 * and "push": the next step's plan runs on a side stream while step t trains, its id blocks are exchanged (own
 * communicator) right behind it, and its counts are on the host long before step t's push has run -- the host's one
 * wait per step finds them there.  With device-side joins (ps_store_join_mode = 1) the all-reduce and the replicated
 * update run on another side stream with a third communicator, beside the push; every rank issues the operations
 * of each communicator in one fixed order.  ps_tune_set("shard_overlap", 0): everything on the training stream with
 * one communicator.  One model suffices (the plan only overwrites key lists that step t no longer reads). */
/* Wire accounting of this model's ps_shard_step calls so far: out[0] steps, [1] id-block bytes sent, [2] row bytes
 * received, [3] gradient bytes sent, [4] all-reduce payload bytes, [5] unique keys requested, [6] keys served as an
 * owner, [7] words of one id block on the wire (n >= 8); n >= 10: [8] steps whose key lists did not fit the wire
 * blocks and were sent again at full size, [9] words of a full-size block. */
int ps_shard_exchange_stats(const ps_model_t *m, int64_t *out, int n);
int ps_shard_step_finish_begin(ps_model_t *m, const ps_comm_ops_t *comm, int is_async,
                               const ps_batch_t *next_batch, float *loss);
/* Measurement: under ps_tune_set("comm_timing", 1) every collective of ps_shard_step is bracketed by HIP events on the
 * stream it is enqueued on.  out8[2 k] = calls, out8[2 k + 1] = total ms of kind k (0 id blocks, 1 rows back, 2 gradients
 * out, 3 all-reduce) since the last call; waits for the model's streams and resets the sums. */
int ps_shard_collective_times(ps_model_t *m, double *out8);

/* Replicated tensors: one flat device buffer to all-reduce(sum), *nfloats floats.
 * Before the model's first ps_shard_step_begin (a host that drives the exchange itself through this pair):
 *   [fc weights+biases | wide G | wide C | wide.bias g], G[k] = this worker's mean delta if it ever touched key k, C[k] = 1 if so.
 * From its first ps_shard_step_begin on (the worker then knows its rank) and with the reference's wide gradient (wide_grad_mode =
 * compat): [fc | wide.bias g | nranks slots of (gbar_w, touched_w packed 24 bits to a float)] -- every worker fills its own slot and
 * zeroes the others, so the sum is an all-gather in any order, and every rank rebuilds G[k] / C[k] over the workers whose bit is
 * set, in rank order (net/PServer.java:164-214: mean over the workers that pushed the key).  1.54 instead of 2.21 MB at
 * configs[2]; ps_tune_set("wide_slots", 0) keeps the dense form. */
int ps_shard_flat_grad(ps_model_t *m, float **flat_dev, int64_t *nfloats);
/* Apply the all-reduced flat buffer: dense g = sum / nworkers (every worker
 * pushes every dense tensor), wide g[k] = G[k] / C[k] over the touching workers. */
int ps_shard_apply_flat(ps_model_t *m, int nworkers);

/* ---- measurement hooks ------------------------------------------------ */
/* Large-table gather run (BASELINE config 4): one table of rows x D floats
 * filled on device (no host copy), n bags of `bag` uniform-random ids,
 * `iters` launches of the forward gather timed with HIP events on the
 * store's stream.  Reports the average kernel time and the algorithmic
 * bytes per launch (read = nnz*(4D+8) [+ 8*(n+1) offsets],
 * written = 4*n*D). */
int ps_bench_gather(ps_store_t *s, int64_t rows, int D, int64_t n, int bag, int iters,
                    uint64_t seed, double *avg_ms_out, double *bytes_read_out,
                    double *bytes_written_out);
/* The same table and lookups (same rows, D, n, bag, seed), ONE launch, then
 * n_sample of the n output rows spread over the launch are copied back with
 * their bag index and ids, so a test can check full-size outputs against
 * the table's definition: float4 i of the table is the k_fill_table hash of
 * (seed, i) (ps_ops.hip), id j is splitmix64((seed ^ 0xABCDEF) + j*golden)
 * mod rows; out = relu(sum of the bag's rows in order).
 * bag_index_out[n_sample], ids_out[n_sample][bag], out_rows[n_sample][D]. */
int ps_bench_gather_check(ps_store_t *s, int64_t rows, int D, int64_t n, int bag, uint64_t seed,
                          int64_t n_sample, int64_t *bag_index_out, int64_t *ids_out, float *out_rows);
/* GEMM micro-benchmark: kind 0 = C[M][N] = A[M][K] * Bt[N][K]^T (FcLayer forward / delta),
 * kind 1 = split-K dW[K][N] = A[M][K]^T * D[M][N].  Average ms per launch (HIP events). */
int ps_bench_gemm(ps_store_t *s, int kind, int M, int N, int K, int nsplit, int iters, double *avg_ms_out);
/* How this store's models join their streams: 1 = device-side flags (every wait bounded: a waiter that does not see
 * its flag within the timeout -- ps_tune_set("spin_timeout_ms"), 2000 by default -- is counted, the next call that
 * waits on the store's stream returns PS_E_STATE and the store switches to events), 0 = events.  The event form is
 * chosen up front under ROCPROF_COUNTER_COLLECTION, HIP_LAUNCH_BLOCKING, AMD_SERIALIZE_KERNEL, HSA_ENABLE_DEBUG,
 * GPU_MAX_HW_QUEUES < 4, and while more than one model of this process drives the device.  why: the reason ("" for 1). */
int ps_store_join_mode(const ps_store_t *s, char *why, int why_cap);
int64_t ps_store_wait_timeouts(const ps_store_t *s);     /* device-side waits that timed out on this store so far */
/* tuning knobs for experiments ("gemm_nt_cfg": 0 auto, 1 128x128, 2 64x128, 3 64x64, 4 128x32 tiles) */
int ps_tune_set(const char *knob, int value);
/* Run `steps` training steps on `batch` back to back, timed with HIP events
 * on the store's stream (ms for all steps). */
int ps_model_time_steps(ps_model_t *m, const ps_batch_t *batch, int steps, double *ms_out);
/* Per-kernel-group HIP-event timing: while enabled every kernel group of the
 * step is bracketed by events on the store's stream; the report is a
 * ';'-separated list of name:launches:total_ms. */
int ps_model_set_profile(ps_model_t *m, int enabled);
/* bracket only the named kernel group (two events per step: cheap enough to
 * leave on inside a timed region); NULL/"" = all groups */
int ps_model_set_profile_filter(ps_model_t *m, const char *group);
int ps_model_profile_report(ps_model_t *m, char *report, int cap);

#ifdef __cplusplus
}
#endif
#endif
