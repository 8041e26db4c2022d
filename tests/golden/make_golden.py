#!/usr/bin/env python3
"""Generates tests/golden/*.json with a binary libzstd 1.5.7 (the library the
reference binds: gradle/libs.versions.toml:9,46; JNI call site
kompressor-zstd--nativelib/src/jvmCommonMain/jni/Wrapper.cpp:112).

Run in the build container only (needs oracle/libzstd_ref.py to find a libzstd
reporting 10507).  Inputs come from the repo's seeded generator
(kompressor_amd/csrc/corpus.c), so every machine can regenerate them; outputs
are frame length + sha256 (and whole frames for small / edge inputs).

    python tests/golden/make_golden.py
"""
import base64
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from libzstd_ref import LibZstd          # noqa: E402
from kompressor_amd import corpus        # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import special_inputs      # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LADDER = [0, 1, 6, 7, 8, 9, 23, 40, 63, 64, 65, 200, 255, 256, 257, 300, 600, 1023, 1024, 1025, 2000, 2048, 4095, 4096,
          5000, 9000, 16383, 16384, 16385, 20000, 32768, 40959, 40960, 40961, 65535, 65536, 65537, 90000, 131071, 131072]


def main():
    z = LibZstd()
    assert z.lib.ZSTD_versionNumber() == 10507
    out = {"libzstd": "1.5.7", "level": 3, "generator": "kompressor_amd/csrc/corpus.c seed 0x4B6F6D70 ^ index",
           "config1": [], "ladder": [], "special": []}
    # configs[1]-shaped: first 2048 slices of the 64 KiB mix
    S = 65536
    n = 2048
    buf = corpus.make(0, n, S)
    for i in range(n):
        d = buf[i * S:(i + 1) * S].tobytes()
        f = z.compress(d)
        out["config1"].append([i, corpus.slice_class(i), len(f), hashlib.sha256(f).hexdigest()])
    # size ladder: 8 slices per size starting at index 1000 (classes T,X,S,B,...)
    for S in LADDER:
        buf = corpus.make(1000, 8, S) if S else np.zeros(0, dtype=np.uint8)
        for k in range(8):
            d = buf[k * S:(k + 1) * S].tobytes()
            f = z.compress(d)
            row = {"index": 1000 + k, "size": S, "len": len(f), "sha256": hashlib.sha256(f).hexdigest()}
            if len(f) <= 700:
                row["frame"] = base64.b64encode(f).decode()
            out["ladder"].append(row)
    # hand-made edge inputs (tests/helpers.py: special_inputs)
    specials = special_inputs()
    for name, d in specials.items():
        f = z.compress(d)
        row = {"name": name, "size": len(d), "len": len(f), "sha256": hashlib.sha256(f).hexdigest(),
               "input_sha256": hashlib.sha256(d).hexdigest()}
        if len(f) <= 700:
            row["frame"] = base64.b64encode(f).decode()
        out["special"].append(row)
    # config[0]: 128 KiB uniform random (class R generator, index 15 of the mix is 'R'; use explicit class)
    buf = corpus.make(0, 1, 131072, mix=ord("R"))
    f = z.compress(buf.tobytes())
    out["config0"] = {"size": 131072, "len": len(f), "sha256": hashlib.sha256(f).hexdigest(), "head": f[:12].hex()}
    # the reference's own known-answer vectors (base64 strings quoted from its tests)
    out["reference_kats"] = {
        "zstd_sampleHello_frame_b64": "KLUv/QRYuQAAaGVsbG8gY29tcHJlc3Npb24gd29ybGR8Qm9f",   # ZstdTest.kt:88
        "zstd_sampleHello_plain": "hello compression world",
    }
    # decoder inputs from other levels / multi-block frames (decode-side coverage)
    dec = []
    for (idx, S, lvl) in [(200, 65536, 1), (201, 65536, 6), (202, 65536, 19), (203, 4096, 19), (300, 300000, 3), (301, 300000, 19)]:
        d = corpus.make(idx, 1, S).tobytes()
        f = z.compress(d, lvl)
        dec.append({"index": idx, "size": S, "level": lvl, "frame": base64.b64encode(f).decode(),
                    "plain_sha256": hashlib.sha256(d).hexdigest()})
    out["decode_only"] = dec
    with open(os.path.join(HERE, "zstd_l3_golden.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "zstd_l3_golden.json"), os.path.getsize(os.path.join(HERE, "zstd_l3_golden.json")), "bytes")


if __name__ == "__main__":
    main()
