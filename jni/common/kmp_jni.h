// kmp_jni.h -- what the two JNI shims (jni/zstd/Wrapper.cpp, jni/zlib/Wrapper.cpp) share.
//
// The shims are the drop-in replacements of the reference's JNI libraries
//   libzstd-jni.so  kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:10-196
//   libz-jni.so     kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:10-153
// with the same exported Java_... names (so the Kotlin side, ZstdWrapper.kt:24-60 / ZlibWrapper.kt:24-54, stays as it
// is), forwarding to the C ABI of libkompressor_hip.so (include/kompressor_hip.h) instead of libzstd / zlib.
// Each shim is one shared library and includes this header once: it defines that library's JNI_OnLoad, which caches the
// two ByteArraySlice field IDs the stream calls write back (the reference does the same in jni/common/src/
// DefaultLoad.cpp:11-26 + SliceClass.cpp:4-7).  Built by kompressor_amd/build.py only where a JDK's jni.h exists.
#pragma once
#include <jni.h>
#include <stddef.h>

namespace kmpjni {

// ByteArraySlice.readStart / writeStart (kompressor-core/.../ByteArraySlice.kt:14-25, both @JvmField Int)
struct SliceFields { jfieldID readStart; jfieldID writeStart; };
inline SliceFields& fields() { static SliceFields f = { nullptr, nullptr }; return f; }

// A Java byte[] borrowed for the duration of one call.  commit = false releases with JNI_ABORT (the input side:
// nothing to copy back), commit = true copies back (the output side), as Wrapper.cpp:117-118 does.
class Borrowed {
public:
    Borrowed(JNIEnv* env, jbyteArray array, bool commit) : env_(env), array_(array), commit_(commit),
        bytes_(env->GetByteArrayElements(array, nullptr)) {}
    ~Borrowed() { if (bytes_) env_->ReleaseByteArrayElements(array_, bytes_, commit_ ? 0 : JNI_ABORT); }
    Borrowed(const Borrowed&) = delete;
    Borrowed& operator=(const Borrowed&) = delete;
    bool ok() const { return bytes_ != nullptr; }
    void* data() const { return bytes_; }
private:
    JNIEnv* env_; jbyteArray array_; bool commit_; jbyte* bytes_;
};

// the cursor write-back of a stream call: absolute indices into the whole arrays
inline void store_cursors(JNIEnv* env, jobject inputSlice, size_t readStart, jobject outputSlice, size_t writeStart)
{
    env->SetIntField(inputSlice, fields().readStart, static_cast<jint>(readStart));
    env->SetIntField(outputSlice, fields().writeStart, static_cast<jint>(writeStart));
}

}  // namespace kmpjni

extern "C" JNIEXPORT jint JNI_OnLoad(JavaVM* vm, void*)
{
    JNIEnv* env = nullptr;
    if (vm->GetEnv(reinterpret_cast<void**>(&env), JNI_VERSION_1_4) != JNI_OK) return JNI_ERR;
    jclass const slice = env->FindClass("com/ensody/kompressor/core/ByteArraySlice");
    if (!slice) return JNI_ERR;
    kmpjni::fields().readStart = env->GetFieldID(slice, "readStart", "I");
    kmpjni::fields().writeStart = env->GetFieldID(slice, "writeStart", "I");
    if (!kmpjni::fields().readStart || !kmpjni::fields().writeStart) return JNI_ERR;
    return env->GetVersion();
}
