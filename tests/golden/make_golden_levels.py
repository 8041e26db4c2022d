#!/usr/bin/env python3
"""Generates tests/golden/zstd_levels_golden.json with a binary libzstd 1.5.7: frame length + sha256 at levels 1 and 2
(strategy "fast"; level 1 is what the reference's Ktor ZstdContentEncoder uses: ZstdContentEncoder.kt:11) for the size
ladder of make_golden.py (8 slices per size, indices 1000..1007) and the first 256 slices of the 64 KiB mix; streaming
frames at level 3; level-1 frames of several blocks (128 KiB < size <= 512 KiB) one-shot and streamed.
Run in the build container only:

    python tests/golden/make_golden_levels.py
"""
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

LADDER = [0, 1, 6, 7, 8, 9, 23, 40, 63, 64, 65, 200, 255, 256, 257, 300, 600, 1023, 1024, 1025, 2000, 2048, 4095, 4096,
          5000, 9000, 16383, 16384, 16385, 20000, 32768, 40959, 40960, 40961, 65535, 65536, 65537, 90000, 131071, 131072]


def main():
    z = LibZstd()
    assert z.lib.ZSTD_versionNumber() == 10507
    out = {"libzstd": "1.5.7", "ladder": [], "config1": []}
    for S in LADDER:
        buf = corpus.make(1000, 8, S) if S else np.zeros(0, dtype=np.uint8)
        for k in range(8):
            d = buf[k * S:(k + 1) * S].tobytes()
            row = [S, k]
            for lvl in (1, 2):
                f = z.compress(d, lvl)
                row += [len(f), hashlib.sha256(f).hexdigest()]
            out["ladder"].append(row)
    S = 65536
    buf = corpus.make(0, 256, S)
    for i in range(256):
        d = buf[i * S:(i + 1) * S].tobytes()
        row = [i]
        for lvl in (1, 2):
            f = z.compress(d, lvl)
            row += [len(f), hashlib.sha256(f).hexdigest()]
        out["config1"].append(row)
    # streaming frames (unknown size while compressing): the reference's finish = false ... finish = true call pattern
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    out["stream"] = []
    for d, cuts in helpers.stream_cases():
        f = z.compress_streaming(d, cuts, 8192)
        out["stream"].append([len(d), cuts[-2], len(f), hashlib.sha256(f).hexdigest()])
    # level 1 above 128 KiB (frames of several blocks, up to its 512 KiB window), one-shot and as streams
    out["l1_multiblock"] = []
    for name, d in helpers.multiblock_inputs():
        if len(d) <= 512 * 1024:
            f = z.compress(d, 1)
            out["l1_multiblock"].append([name, len(d), len(f), hashlib.sha256(f).hexdigest()])
    out["l1_stream"] = []
    for d, cuts in helpers.stream_cases():
        if len(d) <= 512 * 1024:
            f = z.compress_streaming(d, cuts, 8192, 1)
            out["l1_stream"].append([len(d), cuts[-2], len(f), hashlib.sha256(f).hexdigest()])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_levels_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
