/* tests/jni_stub/jni.h — NOT the JDK header and never shipped or linked: a declarations-only stand-in with the handful of
 * JNI 1.6 names jni/plaac_jni.cpp uses, so that tests/test_abi.py can run `g++ -fsyntax-only` on the shim in an image that
 * has no JDK. Signatures follow the JNI specification (C++ flavour: JNIEnv is a struct with member functions). The real
 * build (`make jni`) uses $(JAVA_HOME)/include/jni.h. */
#ifndef PLAAC_TEST_JNI_STUB_H
#define PLAAC_TEST_JNI_STUB_H
#include <cstdint>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef int8_t jbyte;
typedef jint jsize;
class _jobject {};
typedef _jobject *jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;
typedef jarray jobjectArray;
typedef jarray jbyteArray;
struct JNIEnv_ {
    jclass FindClass(const char *name);
    jint ThrowNew(jclass cls, const char *msg);
    void *GetDirectBufferAddress(jobject buf);
    jlong GetDirectBufferCapacity(jobject buf);
    jsize GetArrayLength(jarray a);
    void GetIntArrayRegion(jintArray a, jsize start, jsize len, jint *buf);
    void GetDoubleArrayRegion(jdoubleArray a, jsize start, jsize len, jdouble *buf);
    void SetLongArrayRegion(jlongArray a, jsize start, jsize len, const jlong *buf);
    void GetLongArrayRegion(jlongArray a, jsize start, jsize len, jlong *buf);
    jbyteArray NewByteArray(jsize len);
    void SetByteArrayRegion(jbyteArray a, jsize start, jsize len, const jbyte *buf);
    jobject GetObjectArrayElement(jobjectArray a, jsize index);
};
typedef JNIEnv_ JNIEnv;
struct JavaVM_;
typedef JavaVM_ JavaVM;
#define JNI_VERSION_1_6 0x00010006
#define JNI_ERR (-1)
#endif
