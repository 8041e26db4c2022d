#!/usr/bin/env python3
"""Generates tests/golden/zstd_l3_buffered_golden.json with a binary libzstd 1.5.7 driven EXACTLY as the reference
drives it above 128 KiB: ZSTD_compressStream2 with output slices of max(8192, n / 10) bytes (SliceTransform.kt:33-56),
finish = true from the first call ("oneshot"), and the streaming callers' pattern, finish = false pieces closed by
finish = true, 8 KiB output slices (SliceTransformRawSource.kt:32-55) ("stream").  With output slices below
ZSTD_compressBound libzstd stages the input in chunks of 128 KiB: the frames differ from ZSTD_compress2's whenever the
block pre-splitter cuts, and beyond 2 MiB + 128 KiB the staging buffer wraps.

Inputs: helpers.multiblock_inputs() (<= 2 MiB) and helpers.beyond_window_inputs() (2 MiB .. 6.7 MiB); per row frame
length and sha256.  Run in the build container only:

    python tests/golden/make_golden_buffered.py
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

LAP = 17 * 131072


def main():
    z = LibZstd()
    assert z.lib.ZSTD_versionNumber() == 10507
    rng = random.Random(7)
    rows = []
    for name, d in helpers.multiblock_inputs() + helpers.beyond_window_inputs():
        n = len(d)
        one = z.compress_streaming(d, [0, n], out_chunk=max(8192, n // 10))
        cuts = sorted({0, n, rng.randrange(1, n), rng.randrange(1, n)})
        stream = z.compress_streaming(d, cuts, out_chunk=8192)
        row = {"name": name, "size": n, "input_sha256": hashlib.sha256(d).hexdigest(),
               "oneshot_len": len(one), "oneshot_sha256": hashlib.sha256(one).hexdigest(),
               "stream_len": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest()}
        if n > (2 << 20):                # ZSTD_compress2 into a bound-sized buffer beyond the window (below it: zstd_l3_multiblock_golden.json)
            c2 = z.compress(d)
            row["compress2_len"], row["compress2_sha256"] = len(c2), hashlib.sha256(c2).hexdigest()
        if n <= (512 << 10):             # level 1 (the Ktor encoder's level) through the same one-shot driver, up to its window
            l1 = z.compress_streaming(d, [0, n], out_chunk=max(8192, n // 10), level=1)
            row["l1_oneshot_len"], row["l1_oneshot_sha256"] = len(l1), hashlib.sha256(l1).hexdigest()
        rows.append(row)
    # streams whose closing call brings a few bytes right after the staging buffer wrapped (they are compressed in place)
    tails = []
    for k, t in ((1, 1), (1, 4000), (2, 777)):
        name, d = f"tail_{k}x_lap_plus_{t}", helpers.beyond_window_inputs()[-1][1][: k * LAP + t]
        f = z.compress_streaming(d, [0, k * LAP, k * LAP + t], out_chunk=8192)
        tails.append({"name": name, "laps": k, "tail": t, "len": len(f), "sha256": hashlib.sha256(f).hexdigest()})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_l3_buffered_golden.json")
    with open(path, "w") as fh:
        json.dump({"libzstd": "1.5.7", "level": 3, "rows": rows, "tails": tails}, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes", len(rows), "rows")


if __name__ == "__main__":
    main()
