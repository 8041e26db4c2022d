"""ctypes binding of libkompressor_hip.so (include/kompressor_hip.h).

There is no fallback: if the HIP library is missing or does not export what
the header declares, importing the codecs fails loudly."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMP_LIB_PATH") or os.path.join(HERE, "libkompressor_hip.so")    # KMP_LIB_PATH: another build of the same library (A/B timing)

# every symbol include/kompressor_hip.h declares: (name, restype, argtypes)
_c = ctypes
_P = _c.c_void_p
SIGNATURES = [
    ("kmp_zstd_create_cctx", _P, []),
    ("kmp_zstd_free_cctx", _c.c_size_t, [_P]),
    ("kmp_zstd_cctx_set_parameter", _c.c_size_t, [_P, _c.c_int, _c.c_int]),
    ("kmp_zstd_cctx_load_dictionary", _c.c_size_t, [_P, _P, _c.c_size_t]),
    ("kmp_zstd_compress_stream", _c.c_size_t,
     [_P, _P, _c.c_size_t, _c.POINTER(_c.c_size_t), _P, _c.c_size_t, _c.POINTER(_c.c_size_t), _c.c_int]),
    ("kmp_zstd_create_dctx", _P, []),
    ("kmp_zstd_free_dctx", _c.c_size_t, [_P]),
    ("kmp_zstd_dctx_load_dictionary", _c.c_size_t, [_P, _P, _c.c_size_t]),
    ("kmp_zstd_decompress_stream", _c.c_size_t,
     [_P, _P, _c.c_size_t, _c.POINTER(_c.c_size_t), _P, _c.c_size_t, _c.POINTER(_c.c_size_t)]),
    ("kmp_zstd_is_error", _c.c_uint, [_c.c_size_t]),
    ("kmp_zstd_get_error_name", _c.c_char_p, [_c.c_size_t]),
    ("kmp_zstd_compress_bound", _c.c_size_t, [_c.c_size_t]),
    ("kmp_batch_create", _c.c_int, [_c.POINTER(_P), _c.c_int, _c.c_uint32, _c.c_uint32, _c.c_int]),
    ("kmp_batch_create_ex", _c.c_int, [_c.POINTER(_P), _c.c_int, _c.c_uint32, _c.c_uint32, _P]),
    ("kmp_batch_memory", _c.c_int, [_P, _P]),
    ("kmp_batch_destroy", None, [_P]),
    ("kmp_batch_status", _c.c_int, [_P, _c.POINTER(_c.c_uint32), _P]),
    ("kmp_zstd_compress_batch", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P]),
    ("kmp_batch_piece_range", None, [_c.c_uint32, _c.c_uint32, _c.c_uint32, _c.POINTER(_c.c_uint32), _c.POINTER(_c.c_uint32)]),
    ("kmp_zstd_compress_batch_pieces", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _c.c_uint32, _P]),
    ("kmp_zstd_compress_batch_stream", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _c.c_int, _P]),
    ("kmp_zstd_compress_batch_stream_level", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _c.c_int, _c.c_int, _P]),
    ("kmp_zstd_compress_batch_level", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _c.c_int, _P]),
    ("kmp_zstd_compress_batch_reference", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _c.c_int, _c.c_uint32, _P]),
    ("kmp_zstd_compress_batch_dict", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _c.c_char_p, _c.c_uint32, _P]),
    ("kmp_zstd_decompress_batch", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P, _P, _P]),
    ("kmp_zstd_decompress_batch_dict", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P, _P, _P, _c.c_uint32, _P]),
    ("kmp_deflate_bound", _c.c_size_t, [_c.c_size_t]),
    ("kmp_deflate_compress_batch", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P]),
    ("kmp_zlib_compress_batch", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P]),
    ("kmp_gzip_compress_batch", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P]),
    ("kmp_deflate_compress_batch_level", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _c.c_int, _c.c_int, _P]),
    ("kmp_deflate_compress_batch_params", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P]),
    ("kmp_deflate_bound_params", _c.c_size_t, [_c.c_size_t, _c.c_int, _c.c_int]),
    ("kmp_inflate_batch", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P, _P, _c.c_int, _P]),
    ("kmp_deflate_last_kernel_ms", _c.c_int, [_P, _c.POINTER(_c.c_float)]),
    ("kmp_zlib_create_compressor", _P, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    ("kmp_zlib_free_compressor", _c.c_int, [_P]),
    ("kmp_zlib_compress_stream", _c.c_int,
     [_P, _P, _c.c_size_t, _c.POINTER(_c.c_size_t), _P, _c.c_size_t, _c.POINTER(_c.c_size_t), _c.c_int]),
    ("kmp_zlib_create_decompressor", _P, [_c.c_int]),
    ("kmp_zlib_free_decompressor", _c.c_int, [_P]),
    ("kmp_zlib_decompress_stream", _c.c_int,
     [_P, _P, _c.c_size_t, _c.POINTER(_c.c_size_t), _P, _c.c_size_t, _c.POINTER(_c.c_size_t), _c.c_int]),
    ("kmp_compact_batch", _c.c_int, [_P, _P, _P, _P, _c.c_uint32, _P, _P, _P]),
    ("kmp_batch_set_profiling", _c.c_int, [_P, _c.c_int]),
    ("kmp_batch_last_kernel_ms", _c.c_int, [_P, _c.c_int, _c.POINTER(_c.c_float)]),
    ("kmp_batch_last_chunks", _c.c_int, [_P]),
    ("kmp_batch_last_rounds", _c.c_int, [_P]),
    ("kmp_zstd_compress_host_batch", _c.c_int, [_c.c_int, _c.c_int, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P]),
    ("kmp_zstd_decompress_host_batch", _c.c_int, [_c.c_int, _P, _P, _P, _c.c_uint32, _P, _P, _P, _P, _P]),
    ("kmp_host_register", _c.c_int, [_P, _c.c_size_t]),
    ("kmp_host_unregister", _c.c_int, [_P]),
    ("kmp_host_engines_release", _c.c_int, [_c.c_int]),
    ("kmp_batch_table_rates", _c.c_int, [_P, _c.POINTER(_c.c_float), _c.POINTER(_c.c_float)]),
    ("kmp_debug_copy_meta", _c.c_int, [_P, _P, _c.c_uint32]),
    ("kmp_debug_probe_region", _c.c_int, [_P, _c.c_size_t, _c.c_uint32, _c.c_uint32, _c.POINTER(_c.c_float), _P]),
    ("kmp_last_error", _c.c_char_p, []),
    ("kmp_version", _c.c_char_p, []),
]

ABL_LIB_PATH = os.path.join(HERE, "libkompressor_hip_abl.so")    # the same sources built with -DKMP_ABLATIONS (kompressor_amd/build.py)



class BatchOptions(_c.Structure):          # kmp_batch_options
    _fields_ = [("struct_bytes", _c.c_uint32), ("team_lanes", _c.c_int), ("table_span_gib", _c.c_int), ("table_retry", _c.c_int)]


class BatchMemoryInfo(_c.Structure):       # kmp_batch_memory_info
    _fields_ = [("struct_bytes", _c.c_uint32), ("reserved", _c.c_uint32)] + [(k, _c.c_size_t) for k in
                ("arena", "arena_used", "workspace", "other_tables", "block_chain", "decode_staging", "deflate_workspace", "total")]


_lib = None
_abl = None


def load_ablations():
    """The ablation build of the same library (the second level-3 parser, the fused kernel, every experiment knob as an
    environment variable) as a second handle in this process: what the tests of those paths and the A/B tools use
    (ZstdBatch(..., ablations=True)).  Never what the package's codecs run on."""
    global _abl
    if _abl is None:
        _abl = _open(ABL_LIB_PATH)
    return _abl


def load():
    """Load (once) and return the HIP backend. Raises if it is not built."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def _open(path):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m kompressor_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch ships its own ROCm runtime (torch/lib/libamdhip64.so).  Two HIP runtimes in one process do not share devices:
    # if this library pulled in /opt/rocm's copy first, torch would come up later with its own and kmp_batch_create
    # would find "no ROCm-capable device".  So when torch is importable it is loaded first and this library binds to
    # the runtime already in the process (same soname); without torch (a C caller's process) /opt/rocm's is used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(path)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def last_error():
    return load().kmp_last_error().decode()
