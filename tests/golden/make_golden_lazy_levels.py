#!/usr/bin/env python3
"""Generates tests/golden/zstd_lazy_levels_golden.json with a binary libzstd 1.5.7: length + sha256 of the frames ZSTD_compress2
writes at levels 4 .. 10 for the seeded inputs of tests/helpers.py lazy_level_inputs() -- the levels libzstd runs as strategy "greedy",
"lazy" or "lazy2" at these sizes (level 4 up to 16 KiB; 5 .. 8 at every size up to 128 KiB; 9 and 10 above 16 KiB), with the row-based
match finder above 16 KiB and the hash-chain finder below, as a library built for a machine with 128-bit vectors does (the reference's
JNI library on x86-64 / arm64).  Also libzstd's own parameter table for those levels.  Run in the build container only:

    python tests/golden/make_golden_lazy_levels.py
"""
import ctypes, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import helpers
from libzstd_ref import LibZstd, find_libzstd_157


class CP(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint) for n in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength", "strategy")]


def main():
    lib = find_libzstd_157(); z = LibZstd()
    lib.ZSTD_getCParams.restype = CP; lib.ZSTD_getCParams.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_size_t]
    params = {}
    for lvl in range(4, 11):
        for sz in (3000, 16384, 16385, 40000, 65536, 131072):
            c = lib.ZSTD_getCParams(lvl, sz, 0)
            params[f"{lvl}:{sz}"] = [c.windowLog, c.chainLog, c.hashLog, c.searchLog, c.minMatch, c.strategy]
    inputs = helpers.lazy_level_inputs()
    rows = {}
    for lvl in range(4, 11):
        rows[str(lvl)] = [[len(f), hashlib.sha256(f).hexdigest()] for f in (z.compress(p, lvl) for p in inputs)]
    path = os.path.join(HERE, "zstd_lazy_levels_golden.json")
    json.dump({"libzstd": "1.5.7", "generator": "tests/golden/make_golden_lazy_levels.py", "params": params, "frames": rows}, open(path, "w"), indent=0)
    print(path, len(inputs), "inputs x 7 levels", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
