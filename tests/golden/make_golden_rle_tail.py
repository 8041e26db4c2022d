#!/usr/bin/env python3
"""Generates tests/golden/zstd_rle_tail_golden.json with a binary libzstd 1.5.7: length + sha256 of the frames ZSTD_compress2 writes at
levels -5, -1, 1, 2, 3, 4 (4: above 256 KiB), 5, 7 and 10 for tests/helpers.py rle_tail_cases() -- slices whose last block is a short run
of one byte (see there).  Run in the build container only:

    python tests/golden/make_golden_rle_tail.py
"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import helpers
from libzstd_ref import LibZstd

LEVELS = (-5, -1, 1, 2, 3, 4, 5, 7, 10)


def main():
    z = LibZstd()
    rows = {}
    for name, d in helpers.rle_tail_cases():
        rows[name] = {str(lvl): [len(f), hashlib.sha256(f).hexdigest()] for lvl in LEVELS for f in (z.compress(d, lvl),)}
    path = os.path.join(HERE, "zstd_rle_tail_golden.json")
    json.dump({"libzstd": "1.5.7", "generator": "tests/golden/make_golden_rle_tail.py", "levels": list(LEVELS), "rows": rows}, open(path, "w"), indent=0)
    print(path, len(rows), "inputs x", len(LEVELS), "levels")


if __name__ == "__main__":
    main()
