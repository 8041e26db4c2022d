"""TEST INFRASTRUCTURE -- loader for a binary libzstd 1.5.7, the third-party
library the reference binds (gradle/libs.versions.toml:9,46; call site
kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:112,178).

It is not part of /root/reference and not part of this repo: it is looked up
on the machine (the Pillow wheel bundles one).  Used to (a) pin the C
restatement in oracle/zstd_l3_ref.c, (b) generate tests/golden/, and (c) as
bench.py's cpu_baseline (kind "reference") when present on the GPU box.
Never imported by the product package.
"""
import ctypes
import glob
import os

_CANDIDATES = [
    "/usr/local/lib/python3.10/dist-packages/pillow.libs/libzstd-*.so.1.5.7",
    "/usr/local/lib/python3*/dist-packages/pillow.libs/libzstd*.so*",
    "/usr/lib/python3/dist-packages/pillow.libs/libzstd*.so*",
]


def find_libzstd_157():
    """Return a ctypes handle to a libzstd reporting version 10507, or None."""
    for pat in _CANDIDATES:
        for path in sorted(glob.glob(pat)):
            try:
                # RTLD_DEEPBIND: the library must bind its internal calls to itself even when another libzstd
                # (e.g. the system 1.4.8 pulled in by a profiler's preloaded tool) is already in the process
                lib = ctypes.CDLL(path, mode=os.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
                lib.ZSTD_versionNumber.restype = ctypes.c_uint
                if lib.ZSTD_versionNumber() == 10507:
                    lib._path = path
                    return lib
            except OSError:
                continue
    return None


class _ZBuf(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


class LibZstd:
    """Mirror of what Kompressor's JNI layer does with libzstd (level param id 100)."""

    def __init__(self):
        lib = find_libzstd_157()
        if lib is None:
            raise RuntimeError("no libzstd 1.5.7 on this machine")
        self.lib = lib
        self.path = lib._path
        lib.ZSTD_createCCtx.restype = ctypes.c_void_p
        lib.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
        lib.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
        lib.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_compress2.restype = ctypes.c_size_t
        lib.ZSTD_compressStream2.argtypes = [ctypes.c_void_p, ctypes.POINTER(_ZBuf), ctypes.POINTER(_ZBuf), ctypes.c_int]
        lib.ZSTD_compressStream2.restype = ctypes.c_size_t
        lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        lib.ZSTD_compressBound.restype = ctypes.c_size_t
        lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_decompress.restype = ctypes.c_size_t
        lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
        lib.ZSTD_getErrorName.argtypes = [ctypes.c_size_t]
        lib.ZSTD_getErrorName.restype = ctypes.c_char_p
        self.cctx = lib.ZSTD_createCCtx()
        self.level = None

    def compress(self, data: bytes, level: int = 3) -> bytes:
        lib = self.lib
        if self.level != level:
            lib.ZSTD_CCtx_setParameter(self.cctx, 100, level)
            self.level = level
        cap = lib.ZSTD_compressBound(len(data))
        out = ctypes.create_string_buffer(cap)
        n = lib.ZSTD_compress2(self.cctx, out, cap, data, len(data))
        if lib.ZSTD_isError(n):
            raise RuntimeError(lib.ZSTD_getErrorName(n).decode())
        return out.raw[:n]

    def compress_with_dict(self, data: bytes, dictionary: bytes, level: int = 3) -> bytes:
        """What Kompressor's ZstdCompressor(level, dictionary) does: ZSTD_CCtx_loadDictionary on a fresh context
        (Wrapper.cpp:41-56), then the one-shot compress."""
        lib = self.lib
        lib.ZSTD_CCtx_loadDictionary.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_CCtx_loadDictionary.restype = ctypes.c_size_t
        cctx = lib.ZSTD_createCCtx()
        try:
            lib.ZSTD_CCtx_setParameter(cctx, 100, level)
            r = lib.ZSTD_CCtx_loadDictionary(cctx, dictionary, len(dictionary))
            if lib.ZSTD_isError(r):
                raise RuntimeError(lib.ZSTD_getErrorName(r).decode())
            cap = lib.ZSTD_compressBound(len(data))
            out = ctypes.create_string_buffer(cap)
            n = lib.ZSTD_compress2(cctx, out, cap, data, len(data))
            if lib.ZSTD_isError(n):
                raise RuntimeError(lib.ZSTD_getErrorName(n).decode())
            return out.raw[:n]
        finally:
            lib.ZSTD_freeCCtx(cctx)

    def compress_streaming(self, data: bytes, cuts, out_chunk: int = 8192, level: int = 3) -> bytes:
        """What the reference's streaming callers do (SliceTransformRawSource.kt:32-55): data[cuts[i]:cuts[i+1]] is fed with
        ZSTD_e_continue (finish = false), the last piece with ZSTD_e_end; output drained through out_chunk-byte buffers."""
        lib = self.lib
        Buf = _ZBuf                      # (one type and one prototype for all callers: several threads may be in here)
        cctx = lib.ZSTD_createCCtx()
        lib.ZSTD_CCtx_setParameter(cctx, 100, level)
        src = ctypes.create_string_buffer(data, len(data) + 1)
        out = ctypes.create_string_buffer(out_chunk)
        res = bytearray()
        pieces = list(zip(cuts[:-1], cuts[1:]))
        try:
            for j, (a, b) in enumerate(pieces):
                end = j == len(pieces) - 1
                ib = Buf(ctypes.cast(src, ctypes.c_void_p).value, b, a)
                while True:
                    ob = Buf(ctypes.cast(out, ctypes.c_void_p).value, out_chunk, 0)
                    r = lib.ZSTD_compressStream2(cctx, ctypes.byref(ob), ctypes.byref(ib), 2 if end else 0)
                    if lib.ZSTD_isError(r):
                        raise RuntimeError(lib.ZSTD_getErrorName(r).decode())
                    res += out.raw[:ob.pos]
                    if (end and r == 0) or (not end and ib.pos == ib.size and ob.pos < out_chunk):
                        break
        finally:
            lib.ZSTD_freeCCtx(cctx)
        return bytes(res)

    def decompress(self, frame: bytes, out_size: int) -> bytes:
        lib = self.lib
        out = ctypes.create_string_buffer(max(out_size, 1))
        n = lib.ZSTD_decompress(out, out_size, frame, len(frame))
        if lib.ZSTD_isError(n):
            raise RuntimeError(lib.ZSTD_getErrorName(n).decode())
        return out.raw[:n]

    def decompress_with_dict(self, frame: bytes, out_size: int, dictionary: bytes) -> bytes:
        """ZSTD_decompress_usingDict on a fresh context: what ZstdDecompressor(dictionary) comes to for whole frames."""
        lib = self.lib
        lib.ZSTD_createDCtx.restype = ctypes.c_void_p
        lib.ZSTD_freeDCtx.argtypes = [ctypes.c_void_p]
        lib.ZSTD_decompress_usingDict.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                                  ctypes.c_void_p, ctypes.c_size_t]
        lib.ZSTD_decompress_usingDict.restype = ctypes.c_size_t
        dctx = lib.ZSTD_createDCtx()
        try:
            out = ctypes.create_string_buffer(max(out_size, 1))
            n = lib.ZSTD_decompress_usingDict(dctx, out, out_size, frame, len(frame), dictionary, len(dictionary))
            if lib.ZSTD_isError(n):
                raise RuntimeError(lib.ZSTD_getErrorName(n).decode())
            return out.raw[:n]
        finally:
            lib.ZSTD_freeDCtx(dctx)
