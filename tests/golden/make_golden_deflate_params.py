#!/usr/bin/env python3
"""Generates tests/golden/deflate_params_golden.json with the zlib of this machine (Python's zlib module, zlib 1.2.11): the streams
deflateInit2(level, Z_DEFLATED, windowBits, memLevel, 0) + deflate(Z_FINISH) writes for the cases of helpers.deflate_params_cases()
-- the windowBits / memLevel arguments of the reference's ZlibCompressor (kompressor-zlib--nativelib/src/jvmCommonMain/kotlin/com/
ensody/kompressor/zlib/ZlibCompressor.jvm.kt:7-17 -> jni/Wrapper.cpp:20).  gzip streams are kept with MTIME 0 and OS 3, which is what
this zlib writes on Linux.  One row per case: [len, sha256].

    python tests/golden/make_golden_deflate_params.py
"""
import hashlib
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rows = []
    for case in helpers.deflate_params_cases():
        level, wb, ml, fmt, seed, size, cls = case
        d = helpers.deflate_params_input(case)
        c = zlib.compressobj(level, zlib.DEFLATED, (-wb, wb, wb + 16)[fmt], ml, 0)
        f = c.compress(d) + c.flush()
        if fmt == 2:
            assert f[4:8] == bytes(4) and f[9] == 3, f[:10].hex()
        rows.append([len(f), hashlib.sha256(f).hexdigest()])
    out = {"zlib": zlib.ZLIB_RUNTIME_VERSION, "cases": "helpers.deflate_params_cases()", "rows": rows}
    with open(os.path.join(HERE, "deflate_params_golden.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print(len(rows), "rows", zlib.ZLIB_RUNTIME_VERSION)


if __name__ == "__main__":
    main()
