#!/usr/bin/env python3
"""Generates tests/golden/foreign_frames.{json,bin} (TEST INFRASTRUCTURE; needs a libzstd 1.5.7 on the machine, as
oracle/libzstd_ref.py finds it).  Frames of OTHER compression settings than the product's own -- levels 1..22, a content
checksum, a large window, a tiny block size via many small inputs -- of seeded corpus slices: what the decoder has to
take from any zstd encoder (ZstdDecompressor decodes what it is given; reference tests ZstdTest.kt:28-47 round-trip
the library's own frames).  The .bin holds the frames back to back, the .json their offsets, the recipe of each plain
text (corpus seed, class, size) and its sha256.

    python tests/golden/make_foreign_frames.py
"""
import ctypes
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from kompressor_amd import corpus            # noqa: E402
from libzstd_ref import LibZstd              # noqa: E402

ZSTD_c_compressionLevel, ZSTD_c_windowLog, ZSTD_c_checksumFlag, ZSTD_c_strategy = 100, 101, 201, 107


def plain_of(row):
    if row["kind"] == "corpus":
        return corpus.make(row["seed"], 1, row["size"], mix=ord(row["cls"])).tobytes()
    if row["kind"] == "zeros":
        return bytes(row["size"])
    if row["kind"] == "period":
        p = bytes(range(1, row["period"] + 1))
        return (p * (row["size"] // len(p) + 1))[: row["size"]]
    raise ValueError(row["kind"])


def main():
    z = LibZstd()
    lib = z.lib
    rows, blob = [], bytearray()

    def add(row, level, extra=()):
        data = plain_of(row)
        cctx = lib.ZSTD_createCCtx()
        lib.ZSTD_CCtx_setParameter(cctx, ZSTD_c_compressionLevel, level)
        for k, v in extra:
            lib.ZSTD_CCtx_setParameter(cctx, k, v)
        cap = lib.ZSTD_compressBound(len(data))
        out = ctypes.create_string_buffer(cap)
        n = lib.ZSTD_compress2(cctx, out, cap, data, len(data))
        assert not lib.ZSTD_isError(n)
        lib.ZSTD_freeCCtx(cctx)
        r = dict(row)
        r.update(level=level, extra=[list(e) for e in extra], off=len(blob), len=int(n), plain_sha256=hashlib.sha256(data).hexdigest())
        rows.append(r)
        blob.extend(out.raw[:n])

    seed = 900000
    sizes = [12000, 8000, 16384, 12345, 20000, 30011, 5000, 32768]
    for li, level in enumerate([1, 2, 4, 5, 7, 9, 12, 15, 17, 19, 22]):
        for ci, cls in enumerate("TXSBDIZR"):
            seed += 1
            add({"kind": "corpus", "seed": seed, "cls": cls, "size": sizes[(ci + li) % len(sizes)]}, level)
    # a content checksum; an explicit window larger than the content; long literal runs and long matches
    for cls in "TB":
        seed += 1
        add({"kind": "corpus", "seed": seed, "cls": cls, "size": 20000}, 3, [(ZSTD_c_checksumFlag, 1)])
    seed += 1
    add({"kind": "corpus", "seed": seed, "cls": "T", "size": 200000}, 6, [(ZSTD_c_windowLog, 23)])
    seed += 1
    add({"kind": "corpus", "seed": seed, "cls": "B", "size": 180000}, 19, [])
    add({"kind": "zeros", "size": 300000}, 3)
    add({"kind": "zeros", "size": 70000}, 19)
    for period in (1, 2, 3, 5, 7, 9, 63, 64, 65, 200):
        add({"kind": "period", "period": period, "size": 66000}, 5)
    g = os.path.join(ROOT, "tests", "golden")
    open(os.path.join(g, "foreign_frames.bin"), "wb").write(bytes(blob))
    json.dump({"source": "tests/golden/make_foreign_frames.py with " + os.path.basename(z.path), "blob_sha256": hashlib.sha256(bytes(blob)).hexdigest(), "rows": rows},
              open(os.path.join(g, "foreign_frames.json"), "w"), indent=0)
    print(len(rows), "frames,", len(blob), "bytes")


if __name__ == "__main__":
    main()
