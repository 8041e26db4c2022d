"""ZstdCompressor / ZstdDecompressor with the reference's call pattern, over the
C ABI of libkompressor_hip.so instead of JNI + libzstd:

  ZstdCompressorImpl    kompressor-zstd--nativelib/src/jvmCommonMain/kotlin/com/ensody/kompressor/zstd/ZstdCompressor.jvm.kt:10-51
  ZstdDecompressorImpl  .../zstd/ZstdDecompressor.jvm.kt:10-40
  ZstdWrapper externals .../zstd/ZstdWrapper.kt:24-60 (-> jni/Wrapper.cpp)
"""
import ctypes
import weakref

from . import _lib
from .slice_transform import ByteArraySlice, SliceTransform

COMPRESSION_LEVEL = 100          # ZstdParameter.compressionLevel (ZstdParameter.kt:4)


def _check_error_result(lib, result):
    """checkErrorResult (ZstdCompressor.jvm.kt:45-51): IllegalStateException -> RuntimeError."""
    if result != 0 and lib.kmp_zstd_is_error(result):
        name = lib.kmp_zstd_get_error_name(result).decode()
        signed = result - (1 << 64) if result >= (1 << 63) else result
        raise RuntimeError(f"Bad zstd result code {signed}: {name}")


def _buf(ba):
    return (ctypes.c_char * len(ba)).from_buffer(ba) if len(ba) else None


class ZstdCompressor(SliceTransform):
    def __init__(self, compression_level=3, dictionary=None):
        lib = self._lib = _lib.load()
        self._cctx = lib.kmp_zstd_create_cctx()
        if not self._cctx:
            raise RuntimeError("Failed allocating zstd cctx")
        # createCleaner(cctx, freeCompressor): freed by the GC, possibly on another thread
        self._cleaner = weakref.finalize(self, lib.kmp_zstd_free_cctx, self._cctx)
        _check_error_result(lib, lib.kmp_zstd_cctx_set_parameter(self._cctx, COMPRESSION_LEVEL, compression_level))
        if dictionary is not None:
            _check_error_result(lib, lib.kmp_zstd_cctx_load_dictionary(self._cctx, bytes(dictionary), len(dictionary)))

    def transform(self, input, output, finish):            # noqa: A002
        lib = self._lib
        src_pos = ctypes.c_size_t(input.read_start)
        dst_pos = ctypes.c_size_t(output.write_start)
        result = lib.kmp_zstd_compress_stream(
            self._cctx,
            ctypes.cast(_buf(output.data), ctypes.c_void_p), output.write_limit, ctypes.byref(dst_pos),
            ctypes.cast(_buf(input.data), ctypes.c_void_p), input.write_start, ctypes.byref(src_pos),
            2 if finish else 0)
        input.read_start = src_pos.value                   # Wrapper.cpp:114-115 SetIntField
        output.write_start = dst_pos.value
        _check_error_result(lib, result)
        output.insufficient = input.has_data or (finish and result != 0)


class ZstdDecompressor(SliceTransform):
    def __init__(self, dictionary=None):
        lib = self._lib = _lib.load()
        self._dctx = lib.kmp_zstd_create_dctx()
        if not self._dctx:
            raise RuntimeError("Failed allocating zstd dctx")
        self._cleaner = weakref.finalize(self, lib.kmp_zstd_free_dctx, self._dctx)
        if dictionary is not None:
            _check_error_result(lib, lib.kmp_zstd_dctx_load_dictionary(self._dctx, bytes(dictionary), len(dictionary)))

    def transform(self, input, output, finish):            # noqa: A002
        lib = self._lib
        src_pos = ctypes.c_size_t(input.read_start)
        dst_pos = ctypes.c_size_t(output.write_start)
        result = lib.kmp_zstd_decompress_stream(
            self._dctx,
            ctypes.cast(_buf(output.data), ctypes.c_void_p), output.write_limit, ctypes.byref(dst_pos),
            ctypes.cast(_buf(input.data), ctypes.c_void_p), input.write_start, ctypes.byref(src_pos))
        input.read_start = src_pos.value
        output.write_start = dst_pos.value
        _check_error_result(lib, result)
        output.insufficient = output.is_full and result != 0
