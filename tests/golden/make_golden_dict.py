#!/usr/bin/env python3
"""Generates tests/golden/zstd_dict_golden.json with a binary libzstd 1.5.7: frames compressed at level 3 WITH a
raw-content dictionary, the way Kompressor's ZstdCompressor(level, dictionary) drives the library
(ZSTD_CCtx_loadDictionary, Wrapper.cpp:41-56).  They are decoder inputs: the GPU path decodes them with the same
dictionary (ZstdDecompressor(dictionary), Wrapper.cpp:58-73; reference test ZstdTest.kt:49-65).  Inputs are rebuilt by
tests/helpers.py: dict_cases().  Run in the build container only:

    python tests/golden/make_golden_dict.py
"""
import base64
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from libzstd_ref import LibZstd          # noqa: E402
import helpers                           # noqa: E402


def main():
    z = LibZstd()
    assert z.lib.ZSTD_versionNumber() == 10507
    rows = []
    for name, d, plain in helpers.dict_cases():
        f = z.compress_with_dict(plain, d, 3)
        assert len(f) < len(z.compress(plain, 3)), name          # the dictionary helps, as in the reference's test
        rows.append({"name": name, "dict_sha256": hashlib.sha256(d).hexdigest(), "plain_sha256": hashlib.sha256(plain).hexdigest(),
                     "plain_size": len(plain), "frame": base64.b64encode(f).decode()})
    # compress side: length + sha256 of the frame libzstd makes of each seeded (dictionary, plain) pair
    comp = []
    for d, plain in helpers.dict_compress_cases():
        f = z.compress_with_dict(plain, d, 3)
        comp.append([len(d), len(plain), hashlib.sha256(d + plain).hexdigest()[:16], len(f), hashlib.sha256(f).hexdigest()])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_dict_golden.json")
    with open(path, "w") as fh:
        json.dump({"libzstd": "1.5.7", "level": 3, "rows": rows, "compress": comp}, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes", len(rows), "frames")


if __name__ == "__main__":
    main()
