// TEST-ONLY declarations, NOT a JDK header: just enough of the JNI vocabulary for `g++ -fsyntax-only` to parse
// jni/zstd/Wrapper.cpp, jni/zstd/BatchWrapper.cpp and jni/zlib/Wrapper.cpp on machines without a JDK (tests/test_abi_and_host.py).  Nothing is
// ever linked or run against this file; kompressor_amd/build.py builds the shims only against a real jni.h.
#pragma once
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean; typedef jint jsize;
class _jobject {}; typedef _jobject* jobject; typedef jobject jclass; typedef jobject jstring; typedef jobject jarray; typedef jarray jbyteArray; typedef jarray jlongArray; typedef jarray jintArray;
struct _jfieldID; typedef _jfieldID* jfieldID;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_OK 0
#define JNI_ERR (-1)
#define JNI_ABORT 2
#define JNI_VERSION_1_4 0x00010004
struct JNIEnv {
    jclass FindClass(const char*); jfieldID GetFieldID(jclass, const char*, const char*); jint GetVersion();
    jbyte* GetByteArrayElements(jbyteArray, jboolean*); void ReleaseByteArrayElements(jbyteArray, jbyte*, jint);
    jlong* GetLongArrayElements(jlongArray, jboolean*); void ReleaseLongArrayElements(jlongArray, jlong*, jint);
    jint* GetIntArrayElements(jintArray, jboolean*); void ReleaseIntArrayElements(jintArray, jint*, jint);
    void* GetDirectBufferAddress(jobject);
    jlong GetDirectBufferCapacity(jobject);
    jsize GetArrayLength(jarray); void SetIntField(jobject, jfieldID, jint); jstring NewStringUTF(const char*);
};
struct JavaVM { jint GetEnv(void**, jint); };
