// libz-jni.so re-pointed at the MI355X backend: the six exports of
// kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:10-153 (Kotlin side: ZlibWrapper.kt:24-54) over
// include/kompressor_hip.h.  Results are zlib's codes (Z_OK 0, Z_STREAM_END 1, Z_BUF_ERROR -5 are the benign ones:
// ZlibCompressor.jvm.kt:49-56, ZlibResult.kt:3-13); a stream the GPU path does not serve makes create return 0, which the
// Kotlin side reports as a failed allocation, exactly as it does when deflateInit2 refuses its arguments.
#include "../common/kmp_jni.h"
#include "../../include/kompressor_hip.h"

namespace {
constexpr jint kBufError = -5;                       // Z_BUF_ERROR, what the reference returns when a byte[] cannot be borrowed
inline kmp_zlib_cstream* cstream_of(jlong p) { return reinterpret_cast<kmp_zlib_cstream*>(p); }
inline kmp_zlib_dstream* dstream_of(jlong p) { return reinterpret_cast<kmp_zlib_dstream*>(p); }
}

extern "C" {

// deflateInit2(stream, level, Z_DEFLATED, windowBits, memLevel, strategy) (Wrapper.cpp:10-26); 0 = failed
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zlib_ZlibWrapper_createCompressor(JNIEnv*, jobject, jint level, jint windowBits, jint memLevel, jint strategy)
{
    return reinterpret_cast<jlong>(kmp_zlib_create_compressor(level, windowBits, memLevel, strategy));
}

// deflateEnd (Wrapper.cpp:28-38)
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zlib_ZlibWrapper_freeCompressor(JNIEnv*, jobject, jlong stream)
{
    return static_cast<jlong>(kmp_zlib_free_compressor(cstream_of(stream)));
}

// deflate(stream, finish ? Z_FINISH : Z_NO_FLUSH) (Wrapper.cpp:40-82): next_in / next_out start at the absolute
// indices, avail_* run to the end-exclusive ones; the consumed / produced positions go back into the slices.
JNIEXPORT jint JNICALL Java_com_ensody_kompressor_zlib_ZlibWrapper_compressStream(
    JNIEnv* env, jobject, jlong stream,
    jobject inputSlice, jbyteArray inputByteArray, jint inputStart, jint inputEndExclusive,
    jobject outputSlice, jbyteArray outputByteArray, jint outputStart, jint outputEndExclusive, jboolean finish)
{
    kmpjni::Borrowed out(env, outputByteArray, true);
    if (!out.ok()) return kBufError;
    kmpjni::Borrowed in(env, inputByteArray, false);
    if (!in.ok()) return kBufError;
    size_t srcPos = static_cast<size_t>(inputStart), dstPos = static_cast<size_t>(outputStart);
    int const result = kmp_zlib_compress_stream(cstream_of(stream),
        out.data(), static_cast<size_t>(outputEndExclusive), &dstPos,
        in.data(), static_cast<size_t>(inputEndExclusive), &srcPos, finish ? 1 : 0);
    kmpjni::store_cursors(env, inputSlice, srcPos, outputSlice, dstPos);
    return static_cast<jint>(result);
}

// inflateInit2(stream, windowBits) (Wrapper.cpp:84-97); 0 = failed
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zlib_ZlibWrapper_createDecompressor(JNIEnv*, jobject, jint windowBits)
{
    return reinterpret_cast<jlong>(kmp_zlib_create_decompressor(windowBits));
}

// inflateEnd (Wrapper.cpp:99-109)
JNIEXPORT jlong JNICALL Java_com_ensody_kompressor_zlib_ZlibWrapper_freeDecompressor(JNIEnv*, jobject, jlong stream)
{
    return static_cast<jlong>(kmp_zlib_free_decompressor(dstream_of(stream)));
}

// inflate(stream, finish ? Z_FINISH : Z_NO_FLUSH) (Wrapper.cpp:111-153)
JNIEXPORT jint JNICALL Java_com_ensody_kompressor_zlib_ZlibWrapper_decompressStream(
    JNIEnv* env, jobject, jlong stream,
    jobject inputSlice, jbyteArray inputByteArray, jint inputStart, jint inputEndExclusive,
    jobject outputSlice, jbyteArray outputByteArray, jint outputStart, jint outputEndExclusive, jboolean finish)
{
    kmpjni::Borrowed out(env, outputByteArray, true);
    if (!out.ok()) return kBufError;
    kmpjni::Borrowed in(env, inputByteArray, false);
    if (!in.ok()) return kBufError;
    size_t srcPos = static_cast<size_t>(inputStart), dstPos = static_cast<size_t>(outputStart);
    int const result = kmp_zlib_decompress_stream(dstream_of(stream),
        out.data(), static_cast<size_t>(outputEndExclusive), &dstPos,
        in.data(), static_cast<size_t>(inputEndExclusive), &srcPos, finish ? 1 : 0);
    kmpjni::store_cursors(env, inputSlice, srcPos, outputSlice, dstPos);
    return static_cast<jint>(result);
}

}  // extern "C"
