// ps_jni.cpp -- JNI glue between store.NativeKVStore (java/ps_native/NativeKVStore.java) and the C ABI of
// include/ps_native.h: one native per method declared there, nothing else.  The image this repository is developed in
// has no JDK, so the shim is not linked here; tests/test_jni_shim.py compiles it with -fsyntax-only against a minimal
// declaration-only jni.h (tests/jni_mock/) and checks that every `native` method of the Java sources has its
// Java_store_NativeKVStore_* definition.  Build on a host with a JDK:
//   g++ -std=c++17 -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude
//       java/ps_native/ps_jni.cpp -Lps_amd/lib -lps_amd -o libps_amd_jni.so          (one command line)
#if __has_include(<jni.h>)
#include <jni.h>

#include <string.h>

#include <vector>

#include "ps_native.h"

namespace {

ps_store_t *S(JNIEnv *env, jobject self) {
    jclass c = env->GetObjectClass(self);
    return reinterpret_cast<ps_store_t *>(env->GetLongField(self, env->GetFieldID(c, "handle", "J")));
}
// the reference never throws on this path (it prints and returns null); a native failure is not recoverable there, so
// surface it loudly.  PS_MISSING (Resp 204) is a value, not an error.
bool fail(JNIEnv *env, int rc) {
    if (rc == PS_OK || rc == PS_MISSING) return false;
    env->ThrowNew(env->FindClass("java/lang/RuntimeException"), ps_last_error());
    return true;
}
struct Utf {
    JNIEnv *env; jstring s; const char *c;
    Utf(JNIEnv *e, jstring str) : env(e), s(str), c(str ? e->GetStringUTFChars(str, nullptr) : nullptr) {}
    ~Utf() { if (c) env->ReleaseStringUTFChars(s, c); }
};
struct Floats {
    JNIEnv *env; jfloatArray a; jfloat *p; jint mode;
    Floats(JNIEnv *e, jfloatArray arr, jint m = JNI_ABORT) : env(e), a(arr), p(arr ? e->GetFloatArrayElements(arr, nullptr) : nullptr), mode(m) {}
    ~Floats() { if (p) env->ReleaseFloatArrayElements(a, p, mode); }
};
struct Longs {
    JNIEnv *env; jlongArray a; jlong *p;
    Longs(JNIEnv *e, jlongArray arr) : env(e), a(arr), p(arr ? e->GetLongArrayElements(arr, nullptr) : nullptr) {}
    ~Longs() { if (p) env->ReleaseLongArrayElements(a, p, JNI_ABORT); }
    const int64_t *i64() const { return reinterpret_cast<const int64_t *>(p); }
};
jfloatArray to_java(JNIEnv *env, const float *v, int n) {
    jfloatArray out = env->NewFloatArray(n);
    if (out && n) env->SetFloatArrayRegion(out, 0, n, v);
    return out;
}
ps_batch_t batch_of(int B, const Longs &E, const Floats &X, const Longs &W, const Floats *Y) {
    ps_batch_t b;
    memset(&b, 0, sizeof b);
    b.B = B; b.ids = E.i64(); b.dense = X.p; b.wide_ids = W.i64(); b.labels = Y ? Y->p : nullptr;
    return b;
}

}  // namespace

extern "C" {

// ---- life cycle -------------------------------------------------------------------------------------------------
JNIEXPORT jlong JNICALL Java_store_NativeKVStore_create(JNIEnv *env, jclass, jint device, jlong seed) {
    ps_store_t *s = nullptr;
    fail(env, ps_store_create(device, (uint64_t)seed, &s));
    return reinterpret_cast<jlong>(s);
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_destroy(JNIEnv *, jclass, jlong h) { ps_store_destroy(reinterpret_cast<ps_store_t *>(h)); }

// ---- store.KVStore: get / put / rows / updaters / globalStep ---------------------------------------------------------
JNIEXPORT jfloatArray JNICALL Java_store_NativeKVStore_get(JNIEnv *env, jobject self, jstring key) {
    Utf k(env, key);
    int want = 0;                                         // the key's own length (a row: D floats; a tensor: in x out) -- no 16 MB scratch per call
    int rc = ps_store_key_length(S(env, self), k.c, &want);
    if (rc == PS_MISSING) return nullptr;                 // KVStore.get: null when absent (store/KVStore.java:129-134)
    if (fail(env, rc)) return nullptr;
    std::vector<float> buf((size_t)(want > 0 ? want : 1));
    int len = 0;
    rc = ps_store_get(S(env, self), k.c, buf.data(), (int)buf.size(), &len);
    if (rc == PS_MISSING) return nullptr;
    if (fail(env, rc)) return nullptr;
    return to_java(env, buf.data(), len);
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_put(JNIEnv *env, jobject self, jstring key, jfloatArray val) {
    Utf k(env, key);
    Floats v(env, val);
    fail(env, ps_store_put(S(env, self), k.c, v.p, env->GetArrayLength(val)));
}
JNIEXPORT jfloatArray JNICALL Java_store_NativeKVStore_getRows(JNIEnv *env, jobject self, jint field, jlongArray ids, jint which, jint dim) {
    Longs id(env, ids);
    const jsize n = env->GetArrayLength(ids);
    std::vector<float> buf((size_t)n * (size_t)dim);
    const int rc = ps_store_get_rows(S(env, self), field, id.i64(), n, which, buf.data());
    if (rc == PS_MISSING) return nullptr;                 // a key this shard does not hold (PSClient.getList: empty data)
    if (fail(env, rc)) return nullptr;
    return to_java(env, buf.data(), (int)buf.size());
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_putRows(JNIEnv *env, jobject self, jint field, jlongArray ids, jint which, jfloatArray rows) {
    Longs id(env, ids);
    Floats r(env, rows);
    fail(env, ps_store_put_rows(S(env, self), field, id.i64(), env->GetArrayLength(ids), which, r.p));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_setUpdater(JNIEnv *env, jobject self, jstring key, jstring name) {
    Utf k(env, key), n(env, name);
    ps_updater_t u;
    int rc = ps_updater_from_name(n.c, &u);               // "adam@alfa:0.005@beta1:0.9@..." (Updater.getName())
    if (rc == PS_OK) rc = ps_store_set_updater(S(env, self), k.c, &u);
    fail(env, rc);
}
JNIEXPORT jlong JNICALL Java_store_NativeKVStore_globalStep(JNIEnv *env, jobject self) { return ps_store_global_step(S(env, self)); }

// ---- tables ---------------------------------------------------------------------------------------------------------
JNIEXPORT void JNICALL Java_store_NativeKVStore_createEmbedding(JNIEnv *env, jobject self, jlongArray rowsPerField, jint dim, jint stateSlots,
                                                                jint shard, jint nshards, jint routeMode) {
    Longs r(env, rowsPerField);
    fail(env, ps_store_create_embedding(S(env, self), env->GetArrayLength(rowsPerField), r.i64(), dim, stateSlots, shard, nshards, routeMode));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_createWide(JNIEnv *env, jobject self, jlong wideSize) { fail(env, ps_store_create_wide(S(env, self), wideSize)); }
JNIEXPORT void JNICALL Java_store_NativeKVStore_createFc(JNIEnv *env, jobject self, jint layer, jint in, jint out) {
    fail(env, ps_store_create_fc(S(env, self), layer, in, out));
}

// ---- model/DNN.java, model/WideDeepNN.java: the whole step ---------------------------------------------------------
JNIEXPORT jlong JNICALL Java_store_NativeKVStore_buildModel(JNIEnv *env, jobject self, jint kind, jint F, jint D, jint X, jintArray fcDims,
                                                            jlong wideSize, jint maxBatch) {
    ps_model_config_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.kind = kind; cfg.F = F; cfg.D = D; cfg.X = X; cfg.wide_size = wideSize; cfg.max_batch = maxBatch;
    cfg.emb_grad_mode = PS_GRAD_COMPAT; cfg.wide_grad_mode = PS_GRAD_COMPAT; cfg.emb_sum_order = PS_SUM_AUTO;
    cfg.nfc = env->GetArrayLength(fcDims);
    if (cfg.nfc > 8) { env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "at most 8 FC layers"); return 0; }
    jint *d = env->GetIntArrayElements(fcDims, nullptr);
    for (int i = 0; i < cfg.nfc; ++i) cfg.fc_dims[i] = d[i];
    env->ReleaseIntArrayElements(fcDims, d, JNI_ABORT);
    ps_model_t *m = nullptr;
    fail(env, ps_model_create(S(env, self), &cfg, &m));
    return reinterpret_cast<jlong>(m);
}
JNIEXPORT jfloat JNICALL Java_store_NativeKVStore_train(JNIEnv *env, jobject, jlong model, jlongArray E, jfloatArray X,
                                                        jlongArray W, jfloatArray Y, jint B) {
    Longs e(env, E), w(env, W);
    Floats x(env, X), y(env, Y);
    const ps_batch_t b = batch_of(B, e, x, w, &y);
    float loss = 0.f;
    fail(env, ps_model_train(reinterpret_cast<ps_model_t *>(model), &b, &loss));   // copies in, syncs for the loss
    return loss;
}
JNIEXPORT jfloatArray JNICALL Java_store_NativeKVStore_predict(JNIEnv *env, jobject, jlong model, jlongArray E, jfloatArray X, jlongArray W, jint B) {
    Longs e(env, E), w(env, W);
    Floats x(env, X);
    const ps_batch_t b = batch_of(B, e, x, w, nullptr);
    std::vector<float> p((size_t)B);
    if (fail(env, ps_model_predict(reinterpret_cast<ps_model_t *>(model), &b, p.data()))) return nullptr;
    return to_java(env, p.data(), B);
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_destroyModel(JNIEnv *, jobject, jlong model) { ps_model_destroy(reinterpret_cast<ps_model_t *>(model)); }

// ---- -Dmode=dist: PSRouterClient / PServer as collectives (one process per GPU) -------------------------------------
JNIEXPORT jbyteArray JNICALL Java_store_NativeKVStore_commUniqueId(JNIEnv *env, jclass) {
    char id[384];       // three 128-byte communicator ids
    if (fail(env, ps_comm_rccl_unique_id(id))) return nullptr;
    jbyteArray out = env->NewByteArray(384);
    if (out) env->SetByteArrayRegion(out, 0, 384, reinterpret_cast<const jbyte *>(id));
    return out;
}
JNIEXPORT jlong JNICALL Java_store_NativeKVStore_commCreate(JNIEnv *env, jobject self, jint nranks, jint rank, jbyteArray id) {
    // three 128-byte ids (ps_comm_rccl_unique_id): a shorter array (e.g. the 256 bytes of round 2's two communicators) would be
    // read past its end by ps_comm_rccl_create
    if (id && env->GetArrayLength(id) < 384) {
        env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "commCreate: the RCCL id must be the 384 bytes of commUniqueId()");
        return 0;
    }
    ps_comm_ops_t *ops = new ps_comm_ops_t();
    jbyte *b = id ? env->GetByteArrayElements(id, nullptr) : nullptr;
    const int rc = ps_comm_rccl_create(S(env, self), nranks, rank, reinterpret_cast<const char *>(b), ops);
    if (b) env->ReleaseByteArrayElements(id, b, JNI_ABORT);
    if (fail(env, rc)) { delete ops; return 0; }
    return reinterpret_cast<jlong>(ops);
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_commSelfCheck(JNIEnv *env, jobject self, jlong comm) {
    fail(env, ps_comm_selfcheck(S(env, self), reinterpret_cast<const ps_comm_ops_t *>(comm)));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_commDestroy(JNIEnv *, jobject, jlong comm) {
    ps_comm_ops_t *ops = reinterpret_cast<ps_comm_ops_t *>(comm);
    if (ops) { ps_comm_rccl_destroy(ops); delete ops; }
}
JNIEXPORT jfloat JNICALL Java_store_NativeKVStore_shardStep(JNIEnv *env, jobject, jlong model, jlong comm, jlongArray E, jfloatArray X,
                                                            jlongArray W, jfloatArray Y, jint B, jboolean isPsAsync) {
    Longs e(env, E), w(env, W);
    Floats x(env, X), y(env, Y);
    const ps_batch_t b = batch_of(B, e, x, w, &y);
    float loss = 0.f;       // pull + train + push + psUpdate + barrier of one minibatch, all ranks in step
    fail(env, ps_shard_step(reinterpret_cast<ps_model_t *>(model), &b, reinterpret_cast<const ps_comm_ops_t *>(comm), isPsAsync ? 1 : 0, &loss));
    return loss;
}

// ---- layer.Layer granularity: device buffers + one native per Layer.forward / Layer.backward ---------------------------
JNIEXPORT jlong JNICALL Java_store_NativeKVStore_devAlloc(JNIEnv *env, jobject self, jlong bytes) {
    void *p = nullptr;
    fail(env, ps_dev_alloc(S(env, self), (size_t)bytes, &p));
    return reinterpret_cast<jlong>(p);
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_devFree(JNIEnv *env, jobject self, jlong p) { ps_dev_free(S(env, self), reinterpret_cast<void *>(p)); }
JNIEXPORT void JNICALL Java_store_NativeKVStore_uploadFloats(JNIEnv *env, jobject self, jlong dst, jfloatArray src, jint n) {
    Floats s(env, src);
    fail(env, ps_dev_upload(S(env, self), reinterpret_cast<void *>(dst), s.p, sizeof(float) * (size_t)n));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_uploadLongs(JNIEnv *env, jobject self, jlong dst, jlongArray src, jint n) {
    Longs s(env, src);
    fail(env, ps_dev_upload(S(env, self), reinterpret_cast<void *>(dst), s.p, sizeof(int64_t) * (size_t)n));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_downloadFloats(JNIEnv *env, jobject self, jfloatArray dst, jlong src, jint n) {
    Floats d(env, dst, 0);                                // mode 0: copy back and release
    fail(env, ps_dev_download(S(env, self), d.p, reinterpret_cast<const void *>(src), sizeof(float) * (size_t)n));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_embForward(JNIEnv *env, jobject self, jlong idsDev, jlong offsetsDev, jint B, jint act, jlong outDev, jint ld) {
    fail(env, ps_emb_forward(S(env, self), reinterpret_cast<const int64_t *>(idsDev), reinterpret_cast<const int64_t *>(offsetsDev), B, act,
                             reinterpret_cast<float *>(outDev), ld));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_embBackwardUpdate(JNIEnv *env, jobject self, jlong idsDev, jlong offsetsDev, jlong nnz, jint B, jint act,
                                                                  jlong aDev, jint lda, jlong deltaDev, jint ldd, jint gradMode, jint sumOrder, jboolean apply) {
    fail(env, ps_emb_backward_update(S(env, self), reinterpret_cast<const int64_t *>(idsDev), reinterpret_cast<const int64_t *>(offsetsDev), nnz, B, act,
                                     reinterpret_cast<const float *>(aDev), lda, reinterpret_cast<const float *>(deltaDev), ldd, gradMode, sumOrder, apply ? 1 : 0));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_fcForward(JNIEnv *env, jobject self, jint layer, jint act, jlong xDev, jint ldx, jint B, jlong yDev, jint ldy) {
    fail(env, ps_fc_forward(S(env, self), layer, act, reinterpret_cast<const float *>(xDev), ldx, B, reinterpret_cast<float *>(yDev), ldy));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_fcBackward(JNIEnv *env, jobject self, jint layer, jint act, jlong xDev, jint ldx, jlong yDev, jint ldy,
                                                           jlong deltaDev, jint ldd, jint B, jlong dxDev, jint lddx) {
    fail(env, ps_fc_backward(S(env, self), layer, act, reinterpret_cast<const float *>(xDev), ldx, reinterpret_cast<const float *>(yDev), ldy,
                             reinterpret_cast<float *>(deltaDev), ldd, B, reinterpret_cast<float *>(dxDev), lddx));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_denseUpdate(JNIEnv *env, jobject self, jint layer) {
    const int rc = ps_dense_update(S(env, self), layer);
    if (rc != PS_MISSING) fail(env, rc);                  // nothing pending: KVStore.update over an empty sum map is a no-op
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_sync(JNIEnv *env, jobject self) { fail(env, ps_store_sync(S(env, self))); }

// ---- the rows either side of the path -------------------------------------------------------------------------------
JNIEXPORT jdouble JNICALL Java_store_NativeKVStore_auc(JNIEnv *env, jobject self, jfloatArray p, jfloatArray y) {
    Floats pp(env, p), yy(env, y);
    double auc = 0.0;
    int64_t pos = 0, neg = 0;
    fail(env, ps_auc_compute(S(env, self), pp.p, yy.p, env->GetArrayLength(p), 0, &auc, &pos, &neg));
    return auc;
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_save(JNIEnv *env, jobject self, jstring path) {
    Utf p(env, path);
    fail(env, ps_store_save(S(env, self), p.c));
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_load(JNIEnv *env, jobject self, jstring path) {
    Utf p(env, path);
    fail(env, ps_store_load(S(env, self), p.c));
}

}  // extern "C"
#endif
