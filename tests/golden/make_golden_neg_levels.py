#!/usr/bin/env python3
"""Generates tests/golden/zstd_neg_levels_golden.json with a binary libzstd 1.5.7: frame length + sha256 at the negative levels
-1, -3, -20 and -1000 (strategy "fast" on row 0 of libzstd's parameter tables, a step of 1 - level, literals left uncompressed:
ZSTD_getCParams(level < 0) / ZSTD_literalsCompressionIsDisabled) for the size ladder of make_golden.py (8 slices per size,
indices 1000..1007) and the first 64 slices of the 64 KiB mix.  Run in the build container only:

    python tests/golden/make_golden_neg_levels.py
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
LEVELS = [-1, -3, -20, -1000]


def main():
    z = LibZstd()
    assert z.lib.ZSTD_versionNumber() == 10507
    out = {"libzstd": "1.5.7", "levels": LEVELS, "ladder": [], "config1": []}
    for S in LADDER:
        buf = corpus.make(1000, 8, S) if S else np.zeros(0, dtype=np.uint8)
        for k in range(8):
            d = buf[k * S:(k + 1) * S].tobytes()
            row = [S, k]
            for lvl in LEVELS:
                f = z.compress(d, lvl)
                row += [len(f), hashlib.sha256(f).hexdigest()[:32]]
            out["ladder"].append(row)
    S = 65536
    buf = corpus.make(0, 64, S)
    for i in range(64):
        row = [i]
        for lvl in LEVELS:
            f = z.compress(buf[i * S:(i + 1) * S].tobytes(), lvl)
            row += [len(f), hashlib.sha256(f).hexdigest()[:32]]
        out["config1"].append(row)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_neg_levels_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
