#!/usr/bin/env python3
"""Generates tests/golden/zstd_lazy_big_golden.json with a binary libzstd 1.5.7: length + sha256 of the frames ZSTD_compress2 writes at
levels 4 .. 10 for the seeded inputs of tests/helpers.py lazy_big_inputs() (128 KiB < size <= 2 MiB: frames of several blocks), the sizes
of their blocks' contents (read by feeding ZSTD_decompressStream one block at a time), and libzstd's own parameter table for these
levels and sizes (ZSTD_getCParams).  What ZstdCompressor(level) of the reference returns there through ZSTD_compress2 on a known size
(Wrapper.cpp:112); the product does not serve these sizes at these levels yet -- this pins the oracle of that row.  Run in the build
container only:

    python tests/golden/make_golden_lazy_big.py
"""
import ctypes, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import helpers
from libzstd_ref import LibZstd, find_libzstd_157


class CP(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint) for n in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength", "strategy")]


class Buf(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


def block_sizes(lib, frame, total):
    """the regenerated size of every block of a frame: the streaming decoder is fed one block at a time"""
    lib.ZSTD_createDStream.restype = ctypes.c_void_p
    lib.ZSTD_decompressStream.argtypes = [ctypes.c_void_p, ctypes.POINTER(Buf), ctypes.POINTER(Buf)]; lib.ZSTD_decompressStream.restype = ctypes.c_size_t
    lib.ZSTD_freeDStream.argtypes = [ctypes.c_void_p]
    fhd = frame[4]; ss = (fhd >> 5) & 1; fcs = fhd >> 6; did = fhd & 3
    p = 5 + (0 if ss else 1) + (0, 1, 2, 4)[did] + ((1 if ss else 0) if fcs == 0 else (2, 4, 8)[fcs - 1])
    ds = lib.ZSTD_createDStream(); out = ctypes.create_string_buffer(total + 64); src = ctypes.create_string_buffer(frame, len(frame))
    ob = Buf(ctypes.cast(out, ctypes.c_void_p).value, total + 64, 0); sizes = []; fed = p
    ib = Buf(ctypes.cast(src, ctypes.c_void_p).value, fed, 0); lib.ZSTD_decompressStream(ds, ctypes.byref(ob), ctypes.byref(ib))
    while True:
        h = frame[fed] | (frame[fed + 1] << 8) | (frame[fed + 2] << 16)
        last, typ, sz = h & 1, (h >> 1) & 3, h >> 3
        fed += 3 + (1 if typ == 1 else sz)
        before = ob.pos
        ib = Buf(ctypes.cast(src, ctypes.c_void_p).value, fed, ib.pos)
        for _ in range(4):
            lib.ZSTD_decompressStream(ds, ctypes.byref(ob), ctypes.byref(ib))
        sizes.append(ob.pos - before)
        if last:
            break
    lib.ZSTD_freeDStream(ds)
    return sizes


def main():
    lib = find_libzstd_157(); z = LibZstd()
    lib.ZSTD_getCParams.restype = CP; lib.ZSTD_getCParams.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_size_t]
    params = {}
    for lvl in range(4, 11):
        for sz in (131073, 200000, 262144, 262145, 600000, 1 << 20, (1 << 20) + 1, 2 << 20):
            c = lib.ZSTD_getCParams(lvl, sz, 0)
            params[f"{lvl}:{sz}"] = [c.windowLog, c.chainLog, c.hashLog, c.searchLog, c.minMatch, c.strategy]
    inputs = helpers.lazy_big_inputs()
    rows = {}
    for lvl in range(4, 11):
        rows[str(lvl)] = []
        for p in inputs:
            f = z.compress(p, lvl)
            rows[str(lvl)].append([len(f), hashlib.sha256(f).hexdigest(), block_sizes(lib, f, len(p))])
    path = os.path.join(HERE, "zstd_lazy_big_golden.json")
    json.dump({"libzstd": "1.5.7", "generator": "tests/golden/make_golden_lazy_big.py", "params": params, "frames": rows}, open(path, "w"), indent=0)
    print(path, len(inputs), "inputs x 7 levels", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
