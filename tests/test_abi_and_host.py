"""C-ABI surface + host logic, no GPU: the library loads and exports every
symbol include/kompressor_hip.h declares; error names/numbers follow libzstd;
the SliceTransform mirror obeys the reference's contract (its tests use a fake
XOR codec: kompressor-kotlinx-io/src/jvmTest/.../XorSliceTransformTest.kt:67-77)."""
import ctypes
import os
import re

import pytest

import helpers
from kompressor_amd import _lib, build
from kompressor_amd.slice_transform import ByteArraySlice, SliceTransform


@pytest.fixture(scope="module")
def lib():
    build.build_all()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(helpers.ROOT, "include", "kompressor_hip.h")).read()
    declared = set(re.findall(r"KMP_API[^;(]*?\b(kmp_\w+)\s*\(", hdr))
    assert len(declared) >= 20
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert getattr(lib, name) is not None


def test_error_convention_matches_libzstd(lib):
    # numbering / names of libzstd 1.5.7 (ZSTD_getErrorName), so the reference's
    # "Bad zstd result code ...: name" messages (ZstdCompressor.jvm.kt:45-51) are unchanged
    names = {1: "Error (generic)", 10: "Unknown frame descriptor", 14: "Unsupported frame parameter",
             20: "Data corruption detected", 22: "Restored data doesn't match checksum", 40: "Unsupported parameter",
             42: "Parameter is out of bound", 60: "Operation not authorized at current processing stage",
             64: "Allocation error : not enough memory", 70: "Destination buffer is too small", 72: "Src size is incorrect"}
    for code, name in names.items():
        v = (1 << 64) - code
        assert lib.kmp_zstd_is_error(v) == 1
        assert lib.kmp_zstd_get_error_name(v).decode() == name
    for ok in (0, 1, 65536, (1 << 64) - 121):
        assert lib.kmp_zstd_is_error(ok) == 0
        assert lib.kmp_zstd_get_error_name(ok).decode() == "No error detected"
    z = helpers.live_libzstd()
    if z is not None:
        for code in range(0, 125):
            v = (1 << 64) - code
            assert lib.kmp_zstd_get_error_name(v) == z.lib.ZSTD_getErrorName(ctypes.c_size_t(v)), code


def test_compress_bound(lib):
    for n, b in [(0, 64), (1, 64), (65536, 65824), (131072, 131584), (1 << 20, (1 << 20) + 4096)]:
        assert lib.kmp_zstd_compress_bound(n) == b
    z = helpers.live_libzstd()
    if z is not None:
        for n in [0, 5, 1000, 65536, 100000, 131072, 500000]:
            assert lib.kmp_zstd_compress_bound(n) == z.lib.ZSTD_compressBound(n)


def test_parameter_and_argument_errors_need_no_gpu(lib):
    c = lib.kmp_zstd_create_cctx()
    assert c
    assert lib.kmp_zstd_cctx_set_parameter(c, 100, 3) == 0
    assert lib.kmp_zstd_cctx_set_parameter(c, 100, 0) == 0              # 0 = default level = 3
    assert lib.kmp_zstd_get_error_name(lib.kmp_zstd_cctx_set_parameter(c, 100, 19)).decode() == "Unsupported parameter"
    assert lib.kmp_zstd_get_error_name(lib.kmp_zstd_cctx_set_parameter(c, 101, 20)).decode() == "Unsupported parameter"
    assert lib.kmp_zstd_cctx_load_dictionary(c, None, 0) == 0
    assert lib.kmp_zstd_is_error(lib.kmp_zstd_cctx_load_dictionary(c, b"abc", 3))
    # cursor sanity (positions are absolute indices, Wrapper.cpp:101-110)
    buf = ctypes.create_string_buffer(16)
    dp, sp = ctypes.c_size_t(17), ctypes.c_size_t(0)
    r = lib.kmp_zstd_compress_stream(c, buf, 16, ctypes.byref(dp), buf, 16, ctypes.byref(sp), 2)
    assert lib.kmp_zstd_get_error_name(r).decode() == "Destination buffer is too small"
    lib.kmp_zstd_free_cctx(c)
    assert lib.kmp_batch_create(None, 0, 1, 65536, 8) == -2
    h = ctypes.c_void_p()
    assert lib.kmp_batch_create(ctypes.byref(h), 0, 1, (1 << 30) + 1, 8) == -3   # slices up to 1 GiB
    assert b"1 GiB" in lib.kmp_last_error()


class XorSliceTransform(SliceTransform):
    """The reference's fake codec (XorSliceTransformTest.kt:67-77)."""

    def transform(self, input, output, finish):     # noqa: A002
        n = min(input.remaining_read, output.remaining_write)
        for i in range(n):
            output.data[output.write_start + i] = input.data[input.read_start + i] ^ 0x5A
        input.read_start += n
        output.write_start += n
        output.insufficient = input.has_data


def test_slice_transform_contract_with_fake_codec():
    data = bytes((i * 31) & 0xFF for i in range(16 * 1024 + 3))
    x = XorSliceTransform()
    enc = x.transform_bytes(data)
    assert enc == bytes(b ^ 0x5A for b in data)
    assert x.transform_bytes(enc) == data
    s = ByteArraySlice(8)
    assert s.remaining_read == 0 and s.remaining_write == 8 and not s.has_data and not s.is_full
    i = ByteArraySlice(bytearray(b"abcdef"))
    x.transform(i, s, True)
    assert i.read_start == 6 and s.write_start == 6 and not s.insufficient
    i = ByteArraySlice(bytearray(b"0123456789"))
    s = ByteArraySlice(4)
    x.transform(i, s, True)
    assert s.is_full and s.insufficient and i.remaining_read == 6
    assert XorSliceTransform().transform_bytes(b"") == b""


def test_product_modules_never_touch_the_oracle():
    # the oracle (and any libzstd) is test infrastructure: nothing under kompressor_amd/ may load or call it
    pkg = os.path.join(helpers.ROOT, "kompressor_amd")
    banned = ("kref_", "libkref", "libzstd_ref", "import oracle", "from oracle", "dlopen", "find_library", "pillow.libs")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for b in banned:
                    assert b not in txt, (f, b)


def test_jni_shims_export_what_the_kotlin_side_binds():
    """jni/zstd/Wrapper.cpp and jni/zlib/Wrapper.cpp carry every Java_... export of the reference's JNI libraries
    (ZstdWrapper.kt:24-60: ten, ZlibWrapper.kt:24-54: six), call only functions the C ABI header declares, and parse
    as C++ (against test-only JNI declarations: the image has no JDK; with one, build.py builds the real libraries)."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "kompressor_hip.h")).read()
    declared = set(re.findall(r"KMP_API[^;(]*?\b(kmp_\w+)\s*\(", hdr))
    want = {
        "zstd": ("com_ensody_kompressor_zstd_ZstdWrapper",
                 {"createCompressor", "freeCompressor", "setParameter", "loadCompressorDictionary", "loadDecompressorDictionary",
                  "compressStream", "createDecompressor", "freeDecompressor", "decompressStream", "getErrorName"}),
        "zlib": ("com_ensody_kompressor_zlib_ZlibWrapper",
                 {"createCompressor", "freeCompressor", "compressStream", "createDecompressor", "freeDecompressor", "decompressStream"}),
    }
    for lib, (prefix, names) in want.items():
        path = os.path.join(root, "jni", lib, "Wrapper.cpp")
        src = open(path).read()
        assert set(re.findall(r"Java_" + prefix + r"_(\w+)\s*\(", src)) == names
        called = set(re.findall(r"\b(kmp_(?:zstd|zlib)_\w+)\s*\(", src))
        assert called and called <= declared, called - declared
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(root, "tests", "jni_stub"), path], check=True)
    # the batch for the JVM (no reference counterpart): two more exports of libzstd-jni.so over the host-batch calls
    path = os.path.join(root, "jni", "zstd", "BatchWrapper.cpp")
    src = open(path).read()
    assert set(re.findall(r"Java_com_ensody_kompressor_zstd_ZstdBatchWrapper_(\w+)\s*\(", src)) == {"compressBatch", "decompressBatch", "registerBuffer", "unregisterBuffer", "releaseEngines"}
    called = set(re.findall(r"\b(kmp_zstd_\w+)\s*\(", src))
    assert called == {"kmp_zstd_compress_host_batch", "kmp_zstd_decompress_host_batch"} and called <= declared
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(root, "tests", "jni_stub"), path], check=True)
    from kompressor_amd import build
    if build.find_jni_include() is None:
        assert build.build_jni() == []


def test_cpu_bench_harness_times_the_oracle_port(tmp_path):
    """oracle/cpu_bench.c (bench.py's cpu_baseline leg): pthreads over a few slices with the oracle's restatement as the
    compressor -- frame bytes equal the oracle's own, every pass is timed, and the host-core accounting is sane."""
    import ctypes
    import numpy as np
    import bench
    from kompressor_amd import corpus
    hc = bench.host_cores()
    assert 1 <= hc["threads_used"] <= hc["nproc"] and hc["affinity"] >= 1
    lib = ctypes.CDLL(bench.build_cpu_bench())
    for sym in ("cpubench_zstd_l3", "cpubench_codec", "cpubench_set_zstd_level"):       # everything bench.py calls (built with hidden visibility: an entry point has to ask to be exported)
        assert hasattr(lib, sym), sym
    lib.cpubench_zstd_l3.restype = ctypes.c_int
    lib.cpubench_zstd_l3.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    n, S = 12, 65536
    buf = corpus.make(300, n, S)
    secs = (ctypes.c_double * 2)(); fb, err = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = lib.cpubench_zstd_l3(None, helpers.build_oracle().encode(), buf.ctypes.data, n, S, 3, 2, secs, ctypes.byref(fb), ctypes.byref(err))
    assert rc == 0 and err.value == 0 and secs[0] > 0 and secs[1] > 0
    o = helpers.oracle()
    assert fb.value == sum(len(o.compress(buf[i * S:(i + 1) * S].tobytes())) for i in range(n))
    assert bench.count_gpus_without_hip() in (-1, 0) or bench.count_gpus_without_hip() > 0


def test_product_build_has_one_level3_parser_and_a_short_environment_surface():
    """The product library carries one level-3 parser and reads a short, documented list of environment variables; the second
    parser, the fused kernel and the experiment switches exist only in the ablation build (libkompressor_hip_abl.so, -DKMP_ABLATIONS),
    which exports the same C ABI."""
    import re
    import subprocess
    from kompressor_amd import _lib, build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build.build_all()
    prod = open(build.HIP_LIB, "rb").read()
    abl = open(build.HIP_LIB_ABL, "rb").read()
    names = lambda blob: set(m.decode() for m in re.findall(rb"KMP_[A-Z][A-Z0-9_]+(?=\x00)", blob)) - {"KMP_MAX_CHUNKS"}      # noqa: E731
    env_prod, env_abl = names(prod), names(abl)
    assert len(env_prod) <= 20, sorted(env_prod)
    assert env_prod < env_abl and {"KMP_MATCH_V2", "KMP_FUSE", "KMP_ZSTD_AUTOTUNE"} <= env_abl - env_prod
    # every variable the product reads is in INTEGRATION.md's table
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert all(v in doc for v in env_prod), sorted(v for v in env_prod if v not in doc)
    assert b"k_zstd_match2" not in prod and b"k_zstd_l3_fused" not in prod
    assert b"k_zstd_match2" in abl and b"k_zstd_l3_fused" in abl
    # same exports
    for lib in (_lib.load(), _lib.load_ablations()):
        for name, _, _ in _lib.SIGNATURES:
            assert hasattr(lib, name)
