// BatchWrapper.cpp -- the batch for the JVM: five exports of libzstd-jni.so that the reference does not have.
//
// The reference binds one slice per call (ZstdWrapper.kt:35-46 -> Wrapper.cpp:75-121); a GPU wants thousands.  A Kotlin
// maintainer adds ("The batch from Kotlin"):
//
//     internal object ZstdBatchWrapper {
//         external fun compressBatch(device: Int, level: Int, src: ByteBuffer, inOff: LongArray, inLen: IntArray,
//                                    dst: ByteBuffer, outOff: LongArray, outCap: IntArray, outLen: IntArray): Int
//         external fun decompressBatch(device: Int, src: ByteBuffer, inOff: LongArray, inLen: IntArray,
//                                      dst: ByteBuffer, outOff: LongArray, outCap: IntArray, outLen: IntArray, status: IntArray): Int
//         external fun registerBuffer(buffer: ByteBuffer): Int      // a direct buffer that will be reused: pinned + mapped once
//         external fun unregisterBuffer(buffer: ByteBuffer): Int    // before it is let go
//         external fun releaseEngines(device: Int): Int             // give the pinned staging and device memory back
//     }
//
// src / dst are DIRECT ByteBuffers (ByteBuffer.allocateDirect: stable native memory, no copy at the JNI boundary); slice i is
// src[inOff[i] .. + inLen[i]), its frame goes to dst[outOff[i] ..) with outCap[i] bytes of room and outLen[i] receives its
// size.  Both forward to include/kompressor_hip.h: kmp_zstd_compress_host_batch / kmp_zstd_decompress_host_batch (pinned
// staging + the device batch).  The return value is a KMP_* code (0 = fine).  Compiled into the same library as
// Wrapper.cpp (kompressor_amd/build.py), which provides JNI_OnLoad.
#include <jni.h>
#include <stdint.h>
#include "../../include/kompressor_hip.h"

namespace {
// primitive arrays borrowed for one call; `commit` copies the elements back (the output arrays)
struct Longs {
    JNIEnv* env; jlongArray a; jlong* p;
    Longs(JNIEnv* e, jlongArray arr) : env(e), a(arr), p(e->GetLongArrayElements(arr, nullptr)) {}
    ~Longs() { if (p) env->ReleaseLongArrayElements(a, p, JNI_ABORT); }
};
struct Ints {
    JNIEnv* env; jintArray a; jint* p; bool commit;
    Ints(JNIEnv* e, jintArray arr, bool c) : env(e), a(arr), p(e->GetIntArrayElements(arr, nullptr)), commit(c) {}
    ~Ints() { if (p) env->ReleaseIntArrayElements(a, p, commit ? 0 : JNI_ABORT); }
};
constexpr jint kBadArgument = KMP_ERR_ARG;
// every (offset, length) pair must lie inside its direct buffer: the values come from Kotlin and go to native memcpy / DMA
bool inside(const jlong* off, const jint* len, jsize n, jlong capacity)
{
    for (jsize i = 0; i < n; i++) {
        if (off[i] < 0 || len[i] < 0 || off[i] > capacity || static_cast<jlong>(len[i]) > capacity - off[i]) return false;
    }
    return true;
}
}

extern "C" {

JNIEXPORT jint JNICALL Java_com_ensody_kompressor_zstd_ZstdBatchWrapper_compressBatch(
    JNIEnv* env, jobject, jint device, jint level, jobject src, jlongArray inOff, jintArray inLen,
    jobject dst, jlongArray outOff, jintArray outCap, jintArray outLen)
{
    void* const s = env->GetDirectBufferAddress(src); void* const d = env->GetDirectBufferAddress(dst);
    if (!s || !d) return kBadArgument;                                   // not direct buffers
    jsize const n = env->GetArrayLength(inLen);
    if (env->GetArrayLength(inOff) < n || env->GetArrayLength(outOff) < n || env->GetArrayLength(outCap) < n || env->GetArrayLength(outLen) < n) return kBadArgument;
    Longs io(env, inOff), oo(env, outOff); Ints il(env, inLen, false), oc(env, outCap, false), ol(env, outLen, true);
    if (!io.p || !oo.p || !il.p || !oc.p || !ol.p) return kBadArgument;
    if (!inside(io.p, il.p, n, env->GetDirectBufferCapacity(src)) || !inside(oo.p, oc.p, n, env->GetDirectBufferCapacity(dst))) { ol.commit = false; return kBadArgument; }
    // (jlong / jint and uint64_t / uint32_t have the same size and the values are non-negative offsets and lengths: checked above)
    return kmp_zstd_compress_host_batch(device, level, s, reinterpret_cast<const uint64_t*>(io.p), reinterpret_cast<const uint32_t*>(il.p),
                                        static_cast<uint32_t>(n), d, reinterpret_cast<const uint64_t*>(oo.p), reinterpret_cast<const uint32_t*>(oc.p),
                                        reinterpret_cast<uint32_t*>(ol.p));
}

JNIEXPORT jint JNICALL Java_com_ensody_kompressor_zstd_ZstdBatchWrapper_decompressBatch(
    JNIEnv* env, jobject, jint device, jobject src, jlongArray inOff, jintArray inLen,
    jobject dst, jlongArray outOff, jintArray outCap, jintArray outLen, jintArray status)
{
    void* const s = env->GetDirectBufferAddress(src); void* const d = env->GetDirectBufferAddress(dst);
    if (!s || !d) return kBadArgument;
    jsize const n = env->GetArrayLength(inLen);
    if (env->GetArrayLength(inOff) < n || env->GetArrayLength(outOff) < n || env->GetArrayLength(outCap) < n || env->GetArrayLength(outLen) < n || env->GetArrayLength(status) < n) return kBadArgument;
    Longs io(env, inOff), oo(env, outOff); Ints il(env, inLen, false), oc(env, outCap, false), ol(env, outLen, true), st(env, status, true);
    if (!io.p || !oo.p || !il.p || !oc.p || !ol.p || !st.p) return kBadArgument;
    if (!inside(io.p, il.p, n, env->GetDirectBufferCapacity(src)) || !inside(oo.p, oc.p, n, env->GetDirectBufferCapacity(dst))) { ol.commit = false; st.commit = false; return kBadArgument; }
    return kmp_zstd_decompress_host_batch(device, s, reinterpret_cast<const uint64_t*>(io.p), reinterpret_cast<const uint32_t*>(il.p), static_cast<uint32_t>(n),
                                          d, reinterpret_cast<const uint64_t*>(oo.p), reinterpret_cast<const uint32_t*>(oc.p),
                                          reinterpret_cast<uint32_t*>(ol.p), reinterpret_cast<uint32_t*>(st.p));
}

// a direct buffer that is reused from call to call: pinned and mapped for the device once (kmp_host_register), so that the batch
// calls move its bytes over PCIe without a staging copy
JNIEXPORT jint JNICALL Java_com_ensody_kompressor_zstd_ZstdBatchWrapper_registerBuffer(JNIEnv* env, jobject, jobject buffer)
{
    void* const p = env->GetDirectBufferAddress(buffer); jlong const cap = env->GetDirectBufferCapacity(buffer);
    if (!p || cap <= 0) return kBadArgument;
    return kmp_host_register(p, static_cast<size_t>(cap));
}
JNIEXPORT jint JNICALL Java_com_ensody_kompressor_zstd_ZstdBatchWrapper_unregisterBuffer(JNIEnv* env, jobject, jobject buffer)
{
    void* const p = env->GetDirectBufferAddress(buffer);
    return p ? kmp_host_unregister(p) : kBadArgument;
}
JNIEXPORT jint JNICALL Java_com_ensody_kompressor_zstd_ZstdBatchWrapper_releaseEngines(JNIEnv*, jobject, jint device)
{
    return kmp_host_engines_release(device);
}

}  // extern "C"
