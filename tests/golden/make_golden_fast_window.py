#!/usr/bin/env python3
"""Generates tests/golden/zstd_fast_window_golden.json with a binary libzstd 1.5.7: levels 1, 2 and two negative ones on inputs
LONGER than the level's window (helpers.fast_window_inputs(): 512 KiB + 1 .. 3.4 MiB), in the three ways the reference's callers
make libzstd frame them -- "oneshot": ZSTD_compressStream2 with finish = true from the first call and output slices of
max(8192, n / 10) bytes (SliceTransform.kt:33-56); "stream": finish = false pieces closed by finish = true, 8 KiB output slices
(SliceTransformRawSource.kt:32-55, the Ktor encoder's pattern at level 1: ZstdContentEncoder.kt:11); "compress2": ZSTD_compress2
into a bound-sized buffer.  Beyond window + 128 KiB libzstd's staging buffer wraps and the blocks go through
ZSTD_compressBlock_fast_extDict.  Per row frame length and sha256.  Run in the build container only:

    python tests/golden/make_golden_fast_window.py
"""
import hashlib
import json
import os
import random
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
    rng = random.Random(11)
    rows = []
    for name, level, d in helpers.fast_window_inputs():
        n = len(d)
        one = z.compress_streaming(d, [0, n], out_chunk=max(8192, n // 10), level=level)
        cuts = sorted({0, n, rng.randrange(1, n), rng.randrange(1, n)})
        stream = z.compress_streaming(d, cuts, out_chunk=8192, level=level)
        empty = z.compress_streaming(d, [0, n, n], out_chunk=8192, level=level)
        c2 = z.compress(d, level)
        rows.append({"name": name, "level": level, "size": n, "input_sha256": hashlib.sha256(d).hexdigest(),
                     "oneshot_len": len(one), "oneshot_sha256": hashlib.sha256(one).hexdigest(),
                     "stream_len": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest(),
                     "stream_empty_end_len": len(empty), "stream_empty_end_sha256": hashlib.sha256(empty).hexdigest(),
                     "compress2_len": len(c2), "compress2_sha256": hashlib.sha256(c2).hexdigest()})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_fast_window_golden.json")
    with open(path, "w") as fh:
        json.dump({"libzstd": "1.5.7", "rows": rows}, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes", len(rows), "rows")


if __name__ == "__main__":
    main()
