"""Seeded synthetic slice corpus (csrc/corpus.c), generated on the host cores."""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkmpcorpus.so")
MIX_CONFIG1 = 0      # BASELINE.json configs[1]: 16-class "Silesia-like" mix
MIX_TEXT_BINARY = 1  # configs[3]: text / binary alternating
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `python -m kompressor_amd.build`")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.kmp_corpus_fill.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int]
        _lib.kmp_corpus_fill.restype = None
        _lib.kmp_corpus_class.argtypes = [ctypes.c_uint64, ctypes.c_int]
    return _lib


def slice_class(index, mix=MIX_CONFIG1):
    return chr(_load().kmp_corpus_class(index, mix))


def fill(out: np.ndarray, first_index, count, slice_size, mix=MIX_CONFIG1, threads=None):
    """Fill out[count * slice_size] (uint8, C-contiguous) with slices first_index .. first_index+count-1."""
    lib = _load()
    assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size >= count * slice_size
    if slice_size == 0 or count == 0:
        return out
    threads = threads or min(os.cpu_count() or 1, 16)
    base = out.ctypes.data
    if threads <= 1 or count < 64:
        lib.kmp_corpus_fill(base, first_index, count, slice_size, mix)
        return out
    per = (count + threads - 1) // threads

    def work(t):
        lo = t * per
        hi = min(count, lo + per)
        if lo < hi:
            lib.kmp_corpus_fill(base + lo * slice_size, first_index + lo, hi - lo, slice_size, mix)

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    return out


def make(first_index, count, slice_size, mix=MIX_CONFIG1, threads=None):
    return fill(np.empty(count * slice_size, dtype=np.uint8), first_index, count, slice_size, mix, threads)
