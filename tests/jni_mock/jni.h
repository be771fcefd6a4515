/* Declaration-only subset of the Java Native Interface (JNI specification, chapter 4), written for ONE purpose:
 * letting tests/test_jni_shim.py run `g++ -fsyntax-only` over java/ps_native/ps_jni.cpp in an image without a JDK.
 * Types and member signatures follow the specification; nothing here is ever linked or executed. */
#ifndef PS_AMD_TEST_JNI_MOCK_H
#define PS_AMD_TEST_JNI_MOCK_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;
class _jobject {};
typedef _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jfloatArray;
typedef jarray jlongArray;
typedef jarray jintArray;
typedef jarray jbyteArray;
typedef jobject jthrowable;
struct _jfieldID;
typedef _jfieldID *jfieldID;

#define JNI_ABORT 2
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

struct JNIEnv {
    jclass GetObjectClass(jobject obj);
    jclass FindClass(const char *name);
    jfieldID GetFieldID(jclass clazz, const char *name, const char *sig);
    jlong GetLongField(jobject obj, jfieldID id);
    jint ThrowNew(jclass clazz, const char *msg);
    const char *GetStringUTFChars(jstring str, jboolean *isCopy);
    void ReleaseStringUTFChars(jstring str, const char *chars);
    jsize GetArrayLength(jarray array);
    jfloatArray NewFloatArray(jsize len);
    jbyteArray NewByteArray(jsize len);
    void SetFloatArrayRegion(jfloatArray array, jsize start, jsize len, const jfloat *buf);
    void SetByteArrayRegion(jbyteArray array, jsize start, jsize len, const jbyte *buf);
    jfloat *GetFloatArrayElements(jfloatArray array, jboolean *isCopy);
    jlong *GetLongArrayElements(jlongArray array, jboolean *isCopy);
    jint *GetIntArrayElements(jintArray array, jboolean *isCopy);
    jbyte *GetByteArrayElements(jbyteArray array, jboolean *isCopy);
    void ReleaseFloatArrayElements(jfloatArray array, jfloat *elems, jint mode);
    void ReleaseLongArrayElements(jlongArray array, jlong *elems, jint mode);
    void ReleaseIntArrayElements(jintArray array, jint *elems, jint mode);
    void ReleaseByteArrayElements(jbyteArray array, jbyte *elems, jint mode);
};
#endif
