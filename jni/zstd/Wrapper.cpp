// libzstd-jni.so re-pointed at the MI355X backend: the ten exports of
// kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:10-196 (Kotlin side: ZstdWrapper.kt:24-60), each
// forwarding to the function of include/kompressor_hip.h that replaces the libzstd call the reference makes there.
// Contexts travel as jlong exactly as ZSTD_CCtx* / ZSTD_DCtx* do; results are libzstd-style size_t codes, so
// ZstdCompressor.jvm.kt:41,45-51 and ZstdDecompressor.jvm.kt:38 work unchanged.
#include "../common/kmp_jni.h"
#include "../../include/kompressor_hip.h"

namespace {
constexpr jlong kGeneric = -1;                       // -ZSTD_error_GENERIC, what the reference returns when a byte[] cannot be borrowed
inline kmp_zstd_cctx* cctx_of(jlong p) { return reinterpret_cast<kmp_zstd_cctx*>(p); }
inline kmp_zstd_dctx* dctx_of(jlong p) { return reinterpret_cast<kmp_zstd_dctx*>(p); }
}

extern "C" {

// ZSTD_createCCtx (Wrapper.cpp:10-17)
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_createCompressor(JNIEnv*, jobject)
{
    return reinterpret_cast<jlong>(kmp_zstd_create_cctx());
}

// ZSTD_freeCCtx (Wrapper.cpp:19-27); called from the cleaner thread
JNIEXPORT void JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_freeCompressor(JNIEnv*, jobject, jlong cctx)
{
    kmp_zstd_free_cctx(cctx_of(cctx));
}

// ZSTD_CCtx_setParameter (Wrapper.cpp:29-39); the reference only sets parameter 100 = compressionLevel
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_setParameter(JNIEnv*, jobject, jlong cctx, jint parameter, jint value)
{
    return static_cast<jlong>(kmp_zstd_cctx_set_parameter(cctx_of(cctx), parameter, value));
}

// ZSTD_CCtx_loadDictionary (Wrapper.cpp:41-56): the bytes are copied by the callee
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_loadCompressorDictionary(JNIEnv* env, jobject, jlong cctx, jbyteArray dictionary)
{
    kmpjni::Borrowed dict(env, dictionary, false);
    if (!dict.ok()) return kGeneric;
    return static_cast<jlong>(kmp_zstd_cctx_load_dictionary(cctx_of(cctx), dict.data(), static_cast<size_t>(env->GetArrayLength(dictionary))));
}

// ZSTD_DCtx_loadDictionary (Wrapper.cpp:58-73)
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_loadDecompressorDictionary(JNIEnv* env, jobject, jlong dctx, jbyteArray dictionary)
{
    kmpjni::Borrowed dict(env, dictionary, false);
    if (!dict.ok()) return kGeneric;
    return static_cast<jlong>(kmp_zstd_dctx_load_dictionary(dctx_of(dctx), dict.data(), static_cast<size_t>(env->GetArrayLength(dictionary))));
}

// ZSTD_compressStream2(cctx, &out, &in, finish ? ZSTD_e_end : ZSTD_e_continue) (Wrapper.cpp:75-121): sizes are
// end-exclusive indices into the whole arrays, positions absolute; the new positions go back into the slices.
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_compressStream(
    JNIEnv* env, jobject, jlong cctx,
    jobject inputSlice, jbyteArray inputByteArray, jint inputStart, jint inputEndExclusive,
    jobject outputSlice, jbyteArray outputByteArray, jint outputStart, jint outputEndExclusive, jboolean finish)
{
    kmpjni::Borrowed out(env, outputByteArray, true);
    if (!out.ok()) return kGeneric;
    kmpjni::Borrowed in(env, inputByteArray, false);
    if (!in.ok()) return kGeneric;
    size_t srcPos = static_cast<size_t>(inputStart), dstPos = static_cast<size_t>(outputStart);
    size_t const result = kmp_zstd_compress_stream(cctx_of(cctx),
        out.data(), static_cast<size_t>(outputEndExclusive), &dstPos,
        in.data(), static_cast<size_t>(inputEndExclusive), &srcPos,
        finish ? KMP_ZSTD_e_end : KMP_ZSTD_e_continue);
    kmpjni::store_cursors(env, inputSlice, srcPos, outputSlice, dstPos);
    return static_cast<jlong>(result);
}

// ZSTD_createDCtx (Wrapper.cpp:123-130)
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_createDecompressor(JNIEnv*, jobject)
{
    return reinterpret_cast<jlong>(kmp_zstd_create_dctx());
}

// ZSTD_freeDCtx (Wrapper.cpp:132-140)
JNIEXPORT void JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_freeDecompressor(JNIEnv*, jobject, jlong dctx)
{
    kmp_zstd_free_dctx(dctx_of(dctx));
}

// ZSTD_decompressStream(dctx, &out, &in) (Wrapper.cpp:142-187)
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_decompressStream(
    JNIEnv* env, jobject, jlong dctx,
    jobject inputSlice, jbyteArray inputByteArray, jint inputStart, jint inputEndExclusive,
    jobject outputSlice, jbyteArray outputByteArray, jint outputStart, jint outputEndExclusive)
{
    kmpjni::Borrowed out(env, outputByteArray, true);
    if (!out.ok()) return kGeneric;
    kmpjni::Borrowed in(env, inputByteArray, false);
    if (!in.ok()) return kGeneric;
    size_t srcPos = static_cast<size_t>(inputStart), dstPos = static_cast<size_t>(outputStart);
    size_t const result = kmp_zstd_decompress_stream(dctx_of(dctx),
        out.data(), static_cast<size_t>(outputEndExclusive), &dstPos,
        in.data(), static_cast<size_t>(inputEndExclusive), &srcPos);
    kmpjni::store_cursors(env, inputSlice, srcPos, outputSlice, dstPos);
    return static_cast<jlong>(result);
}

// ZSTD_isError(code) ? ZSTD_getErrorName(code) : null (Wrapper.cpp:189-196)
JNIEXPORT jstring JNICALL Java_com_ensody_kompressor_zstd_ZstdWrapper_getErrorName(JNIEnv* env, jobject, jlong code)
{
    size_t const c = static_cast<size_t>(code);
    if (!kmp_zstd_is_error(c)) return nullptr;
    return env->NewStringUTF(kmp_zstd_get_error_name(c));
}

}  // extern "C"
