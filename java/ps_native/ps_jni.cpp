// ps_jni.cpp -- JNI glue between store.NativeKVStore and the C ABI of include/ps_native.h.
// NOT built here (no jni.h in this image):  g++ -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux \
//     -Iinclude java/ps_native/ps_jni.cpp -Lps_amd/lib -lps_amd -o libps_amd_jni.so
#if __has_include(<jni.h>)
#include <jni.h>
#include <vector>
#include "ps_native.h"

static ps_store_t *S(JNIEnv *env, jobject self) {
    jclass c = env->GetObjectClass(self);
    return reinterpret_cast<ps_store_t *>(env->GetLongField(self, env->GetFieldID(c, "handle", "J")));
}
static void fail(JNIEnv *env, int rc) {       // the reference never throws on this path; surface native errors loudly
    if (rc != PS_OK && rc != PS_MISSING) env->ThrowNew(env->FindClass("java/lang/RuntimeException"), ps_last_error());
}

extern "C" {
JNIEXPORT jlong JNICALL Java_store_NativeKVStore_create(JNIEnv *env, jclass, jint device, jlong seed) {
    ps_store_t *s = nullptr;
    fail(env, ps_store_create(device, (uint64_t)seed, &s));
    return reinterpret_cast<jlong>(s);
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_destroy(JNIEnv *, jclass, jlong h) { ps_store_destroy(reinterpret_cast<ps_store_t *>(h)); }

JNIEXPORT jfloatArray JNICALL Java_store_NativeKVStore_get(JNIEnv *env, jobject self, jstring key) {
    const char *k = env->GetStringUTFChars(key, nullptr);
    std::vector<float> buf(1 << 22);
    int len = 0;
    const int rc = ps_store_get(S(env, self), k, buf.data(), (int)buf.size(), &len);
    env->ReleaseStringUTFChars(key, k);
    if (rc == PS_MISSING) return nullptr;                 // KVStore.get: null when absent
    fail(env, rc);
    jfloatArray out = env->NewFloatArray(len);
    env->SetFloatArrayRegion(out, 0, len, buf.data());
    return out;
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_put(JNIEnv *env, jobject self, jstring key, jfloatArray val) {
    const char *k = env->GetStringUTFChars(key, nullptr);
    jfloat *v = env->GetFloatArrayElements(val, nullptr);
    fail(env, ps_store_put(S(env, self), k, v, env->GetArrayLength(val)));
    env->ReleaseFloatArrayElements(val, v, JNI_ABORT);
    env->ReleaseStringUTFChars(key, k);
}
JNIEXPORT void JNICALL Java_store_NativeKVStore_setUpdater(JNIEnv *env, jobject self, jstring key, jstring name) {
    const char *k = env->GetStringUTFChars(key, nullptr), *n = env->GetStringUTFChars(name, nullptr);
    ps_updater_t u;
    int rc = ps_updater_from_name(n, &u);                 // "adam@alfa:0.005@beta1:0.9@..." (Updater.getName())
    if (rc == PS_OK) rc = ps_store_set_updater(S(env, self), k, &u);
    env->ReleaseStringUTFChars(key, k); env->ReleaseStringUTFChars(name, n);
    fail(env, rc);
}
JNIEXPORT jfloat JNICALL Java_store_NativeKVStore_train(JNIEnv *env, jobject, jlong model, jlongArray E, jfloatArray X,
                                                        jlongArray W, jfloatArray Y, jint B) {
    ps_batch_t b = {};
    b.B = B;
    jlong *e = env->GetLongArrayElements(E, nullptr);
    jfloat *x = env->GetFloatArrayElements(X, nullptr), *y = env->GetFloatArrayElements(Y, nullptr);
    jlong *w = W ? env->GetLongArrayElements(W, nullptr) : nullptr;
    b.ids = reinterpret_cast<const int64_t *>(e); b.dense = x; b.labels = y; b.wide_ids = reinterpret_cast<const int64_t *>(w);
    float loss = 0.f;
    const int rc = ps_model_train(reinterpret_cast<ps_model_t *>(model), &b, &loss);   // copies in, syncs for the loss
    env->ReleaseLongArrayElements(E, e, JNI_ABORT); env->ReleaseFloatArrayElements(X, x, JNI_ABORT);
    env->ReleaseFloatArrayElements(Y, y, JNI_ABORT); if (w) env->ReleaseLongArrayElements(W, w, JNI_ABORT);
    fail(env, rc);
    return loss;
}
JNIEXPORT jfloat JNICALL Java_store_NativeKVStore_shardStep(JNIEnv *env, jobject, jlong model, jlong comm, jlongArray E, jfloatArray X,
                                                            jlongArray W, jfloatArray Y, jint B, jboolean isPsAsync) {
    ps_batch_t b = {};
    b.B = B;
    jlong *e = env->GetLongArrayElements(E, nullptr);
    jfloat *x = env->GetFloatArrayElements(X, nullptr), *y = env->GetFloatArrayElements(Y, nullptr);
    jlong *w = W ? env->GetLongArrayElements(W, nullptr) : nullptr;
    b.ids = reinterpret_cast<const int64_t *>(e); b.dense = x; b.labels = y; b.wide_ids = reinterpret_cast<const int64_t *>(w);
    float loss = 0.f;       // pull + train + push + psUpdate + barrier of one minibatch, all ranks in step
    const int rc = ps_shard_step(reinterpret_cast<ps_model_t *>(model), &b, reinterpret_cast<const ps_comm_ops_t *>(comm), isPsAsync ? 1 : 0, &loss);
    env->ReleaseLongArrayElements(E, e, JNI_ABORT); env->ReleaseFloatArrayElements(X, x, JNI_ABORT);
    env->ReleaseFloatArrayElements(Y, y, JNI_ABORT); if (w) env->ReleaseLongArrayElements(W, w, JNI_ABORT);
    fail(env, rc);
    return loss;
}
// commUniqueId / commCreate: ps_comm_rccl_unique_id into a byte[256]; a heap ps_comm_ops_t filled by ps_comm_rccl_create.
// auc / save / load: ps_auc_compute(on_device = 0), ps_store_save, ps_store_load.
// getRows / putRows / createEmbedding / createWide / createFc / buildModel / predict / destroyModel / globalStep:
// the same pattern over ps_store_get_rows, ps_store_put_rows, ps_store_create_*, ps_model_create, ps_model_predict,
// ps_model_destroy, ps_store_global_step.
}
#endif
