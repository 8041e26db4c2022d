"""ZlibCompressor with the reference's call pattern over the C ABI:

  ZlibCompressorImpl  kompressor-zlib--nativelib/src/jvmCommonMain/kotlin/com/ensody/kompressor/zlib/ZlibCompressor.jvm.kt:19-56
  ZlibFormat          .../commonMain/.../ZlibFormat.kt:32-57 (Raw negates windowBits)
  ZlibWrapper externs .../zlib/ZlibWrapper.kt:24-54 (-> jni/Wrapper.cpp)

The GPU path covers ZlibFormat.Raw, Zlib and Gzip at zlib's levels 1 .. 9 (6, the default, is BASELINE
configs[4] and the reference's deflate KAT) and inflate for all of them plus AutoDetectZlibGzip."""
import ctypes
import weakref

from . import _lib
from .slice_transform import SliceTransform

Z_OK, Z_STREAM_END, Z_BUF_ERROR = 0, 1, -5
_RESULT_NAMES = {0: "Z_OK", 1: "Z_STREAM_END", 2: "Z_NEED_DICT", -1: "Z_ERRNO", -2: "Z_STREAM_ERROR",
                 -3: "Z_DATA_ERROR", -4: "Z_MEM_ERROR", -5: "Z_BUF_ERROR", -6: "Z_VERSION_ERROR"}   # ZlibResult.kt:3-13


class ZlibFormat:
    """ZlibFormat.kt:7-57: how windowBits selects the wrapper."""
    Zlib, Gzip, Raw, AutoDetectZlibGzip, UnmodifiedWindowBits = "zlib", "gzip", "raw", "auto", "unmodified"

    @staticmethod
    def adjust_window_bits_for_compression(fmt, window_bits):
        if fmt == ZlibFormat.AutoDetectZlibGzip:
            raise RuntimeError("Compression can't be used with auto-detection")       # ZlibFormat.kt:28
        return ZlibFormat.adjust_window_bits_for_decompression(fmt, window_bits)

    @staticmethod
    def adjust_window_bits_for_decompression(fmt, window_bits):
        if fmt == ZlibFormat.Zlib:
            if not 9 <= window_bits <= 15:
                raise ValueError("windowBits must be between 9..15")
            return window_bits
        if fmt == ZlibFormat.Gzip:
            if not 8 <= window_bits <= 15:
                raise ValueError("windowBits must be between 8..15")
            return window_bits + 16          # ZlibFormat.kt:39-42
        if fmt == ZlibFormat.Raw:
            if not 8 <= window_bits <= 15:
                raise ValueError("windowBits must be between 8..15")
            return -window_bits              # ZlibFormat.kt:44-47
        if fmt == ZlibFormat.AutoDetectZlibGzip:
            if not 9 <= window_bits <= 15:
                raise ValueError("windowBits must be between 8..15")
            return window_bits + 32          # ZlibFormat.kt:52-55
        return window_bits


def _check_error_result(result):
    """checkErrorResult (ZlibCompressor.jvm.kt:49-56)."""
    if result not in (Z_OK, Z_STREAM_END, Z_BUF_ERROR):
        raise RuntimeError(f"Bad zlib result code {result}: {_RESULT_NAMES.get(result)}")


def _buf(ba):
    return (ctypes.c_char * len(ba)).from_buffer(ba) if len(ba) else None


class ZlibCompressor(SliceTransform):
    def __init__(self, format=ZlibFormat.Zlib, compression_level=-1, window_bits=15, mem_level=8):   # noqa: A002
        lib = self._lib = _lib.load()
        wb = ZlibFormat.adjust_window_bits_for_compression(format, window_bits)
        self._stream = lib.kmp_zlib_create_compressor(compression_level, wb, mem_level, 0)
        if not self._stream:
            raise RuntimeError("Failed allocating zlib stream")
        self._cleaner = weakref.finalize(self, lib.kmp_zlib_free_compressor, self._stream)

    def transform(self, input, output, finish):            # noqa: A002
        lib = self._lib
        src_pos = ctypes.c_size_t(input.read_start)
        dst_pos = ctypes.c_size_t(output.write_start)
        result = lib.kmp_zlib_compress_stream(
            self._stream,
            ctypes.cast(_buf(output.data), ctypes.c_void_p), output.write_limit, ctypes.byref(dst_pos),
            ctypes.cast(_buf(input.data), ctypes.c_void_p), input.write_start, ctypes.byref(src_pos),
            1 if finish else 0)
        input.read_start = src_pos.value
        output.write_start = dst_pos.value
        _check_error_result(result)
        output.insufficient = input.has_data or (finish and result != Z_STREAM_END)


class ZlibDecompressor(SliceTransform):
    """ZlibDecompressorImpl (kompressor-zlib--nativelib/.../zlib/ZlibDecompressor.jvm.kt) over kmp_zlib_decompress_stream."""

    def __init__(self, format=ZlibFormat.Zlib, window_bits=15):   # noqa: A002
        lib = self._lib = _lib.load()
        wb = ZlibFormat.adjust_window_bits_for_decompression(format, window_bits)
        self._stream = lib.kmp_zlib_create_decompressor(wb)
        if not self._stream:
            raise RuntimeError("Failed allocating zlib stream")
        self._cleaner = weakref.finalize(self, lib.kmp_zlib_free_decompressor, self._stream)

    def transform(self, input, output, finish):            # noqa: A002
        lib = self._lib
        src_pos = ctypes.c_size_t(input.read_start)
        dst_pos = ctypes.c_size_t(output.write_start)
        result = lib.kmp_zlib_decompress_stream(
            self._stream,
            ctypes.cast(_buf(output.data), ctypes.c_void_p), output.write_limit, ctypes.byref(dst_pos),
            ctypes.cast(_buf(input.data), ctypes.c_void_p), input.write_start, ctypes.byref(src_pos),
            1 if finish else 0)
        input.read_start = src_pos.value
        output.write_start = dst_pos.value
        _check_error_result(result)
        output.insufficient = output.is_full and result != Z_STREAM_END
