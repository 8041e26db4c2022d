#!/usr/bin/env python3
"""Generates tests/golden/zstd_level2_big_golden.json with a binary libzstd 1.5.7: level-2 frames of several blocks
(128 KiB < size <= 1 MiB, level 2's window) -- ZSTD_compress2's frame, the frame the reference's one-shot driver gets
(output slices of max(8192, n / 10) bytes: SliceTransform.kt:47-56), and streamed frames (finish = false ... finish = true).
Level 2 is the odd one: sizes above 128 KiB up to 256 KiB use a double-fast row, everything else a fast one.
Run in the build container only:

    python tests/golden/make_golden_level2_big.py
"""
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
    out = {"libzstd": "1.5.7", "multiblock": [], "stream": []}
    for name, d in helpers.multiblock_inputs():
        if len(d) <= 1024 * 1024:
            f0 = z.compress(d, 2)
            f3 = z.compress_streaming(d, [0, len(d)], max(8192, len(d) // 10), 2)
            out["multiblock"].append([name, len(d), len(f0), hashlib.sha256(f0).hexdigest(), len(f3), hashlib.sha256(f3).hexdigest()])
    for d, cuts in helpers.stream_cases():
        if len(d) <= 1024 * 1024:
            f = z.compress_streaming(d, cuts, 8192, 2)
            out["stream"].append([len(d), cuts[-2], len(f), hashlib.sha256(f).hexdigest()])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_level2_big_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes", len(out["multiblock"]), len(out["stream"]))


if __name__ == "__main__":
    main()
