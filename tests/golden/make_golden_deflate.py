#!/usr/bin/env python3
"""Generates tests/golden/deflate_l6_golden.json with the zlib of this machine
(Python's zlib module, zlib 1.2.11): raw DEFLATE, level 6, windowBits 15, memLevel 8,
strategy 0 -- the settings of the reference's ZlibCompressor(ZlibFormat.Raw, 6)
(kompressor-zlib--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:20,73).  The reference
pins zlib 1.3.1, which is not available here; see DESIGN.md.

    python tests/golden/make_golden_deflate.py
"""
import base64
import hashlib
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kompressor_amd import corpus        # noqa: E402
from helpers import special_inputs      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LADDER = [0, 1, 2, 3, 4, 5, 10, 100, 258, 262, 263, 1000, 4096, 16383, 16385, 32768, 40000, 65274, 65275, 65276, 65500, 65535, 65536]


def raw6(d):
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0)
    return c.compress(d) + c.flush()


def main():
    out = {"zlib": zlib.ZLIB_RUNTIME_VERSION, "settings": "level 6, windowBits -15, memLevel 8, strategy 0",
           "config4": [], "ladder": [], "special": []}
    S, n = 65536, 1024
    buf = corpus.make(0, n, S)
    for i in range(n):
        f = raw6(buf[i * S:(i + 1) * S].tobytes())
        out["config4"].append([i, corpus.slice_class(i), len(f), hashlib.sha256(f).hexdigest()])
    for S in LADDER:
        buf = corpus.make(1000, 8, S) if S else np.zeros(0, dtype=np.uint8)
        for k in range(8):
            f = raw6(buf[k * S:(k + 1) * S].tobytes())
            row = {"index": 1000 + k, "size": S, "len": len(f), "sha256": hashlib.sha256(f).hexdigest()}
            if len(f) <= 400:
                row["stream"] = base64.b64encode(f).decode()
            out["ladder"].append(row)
    for name, d in special_inputs().items():
        if len(d) > 65536:
            continue
        f = raw6(d)
        row = {"name": name, "size": len(d), "len": len(f), "sha256": hashlib.sha256(f).hexdigest()}
        if len(f) <= 400:
            row["stream"] = base64.b64encode(f).decode()
        out["special"].append(row)
    # slices above 64 KiB: zlib's window slides (helpers.deflate_long_inputs)
    from helpers import deflate_long_inputs
    out["long"] = []
    for name, d in deflate_long_inputs():
        f = raw6(d)
        out["long"].append({"name": name, "size": len(d), "input_sha256": hashlib.sha256(d).hexdigest(), "len": len(f), "sha256": hashlib.sha256(f).hexdigest()})
    # the reference's only compress-side known-answer vector (ZlibTest.kt:66-84): zlib format, default level;
    # its raw DEFLATE body is what level 6 raw must produce for the same text
    kat = base64.b64decode("eJzLSM3JyVdIzs8tKEotLs7Mz1Mozy/KSQEAbW0JLw==")
    out["reference_kat"] = {"plain": "hello compression world", "zlib_b64": "eJzLSM3JyVdIzs8tKEotLs7Mz1Mozy/KSQEAbW0JLw==",
                            "raw_body_b64": base64.b64encode(kat[2:-4]).decode()}
    assert raw6(b"hello compression world") == kat[2:-4]
    path = os.path.join(HERE, "deflate_l6_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
