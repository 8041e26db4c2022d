#!/usr/bin/env python3
"""Generates tests/golden/zstd_level4_golden.json with a binary libzstd 1.5.7: frame length + sha256 at level 4 where that
level is the double-fast parse with one block (ZSTD_getCParams(4, n, 0): slices above 16 KiB up to 128 KiB) -- the size ladder
of make_golden.py above 16 KiB (8 slices per size, indices 1000..1007) and the first 256 slices of the 64 KiB mix -- plus
the parameters libzstd reports per size class and level (what decides which levels this backend can serve).
Run in the build container only:

    python tests/golden/make_golden_level4.py
"""
import ctypes
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from libzstd_ref import LibZstd          # noqa: E402
from kompressor_amd import corpus        # noqa: E402

LADDER = [16385, 20000, 32768, 40959, 40960, 40961, 65535, 65536, 65537, 90000, 131071, 131072]


class CP(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint) for n in ("windowLog", "chainLog", "hashLog", "searchLog", "minMatch", "targetLength", "strategy")]


def main():
    z = LibZstd()
    assert z.lib.ZSTD_versionNumber() == 10507
    out = {"libzstd": "1.5.7", "level": 4, "ladder": [], "config1": [], "cparams": []}
    f = z.lib.ZSTD_getCParams
    f.restype = CP
    f.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_size_t]
    for size in (4096, 16384, 16385, 32768, 65536, 131072, 131073, 262144, 262145, 1 << 20, 4 << 20):
        for lvl in (1, 2, 3, 4, 5, 6):
            c = f(lvl, size, 0)
            out["cparams"].append([size, lvl] + [getattr(c, n) for n, _ in CP._fields_])
    for S in LADDER:
        buf = corpus.make(1000, 8, S)
        for k in range(8):
            fr = z.compress(buf[k * S:(k + 1) * S].tobytes(), 4)
            out["ladder"].append([S, k, len(fr), hashlib.sha256(fr).hexdigest()])
    S = 65536
    buf = corpus.make(0, 256, S)
    for i in range(256):
        fr = z.compress(buf[i * S:(i + 1) * S].tobytes(), 4)
        out["config1"].append([i, len(fr), hashlib.sha256(fr).hexdigest()])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_level4_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
