#!/usr/bin/env python3
"""Generates tests/golden/zstd_l3_multiblock_golden.json with a binary libzstd 1.5.7: frames of inputs above
128 KiB (several blocks: pre-splitter decisions, repcodes and Huffman tables carried between blocks, raw / RLE /
treeless-literals blocks).  Inputs are rebuilt on any machine by tests/helpers.py: multiblock_inputs(); outputs are
frame length, sha256 and the block list.  Run in the build container only:

    python tests/golden/make_golden_multiblock.py
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
    rows = []
    for name, d in helpers.multiblock_inputs():
        f = z.compress(d, 3)
        rows.append({"name": name, "size": len(d), "input_sha256": hashlib.sha256(d).hexdigest(), "len": len(f),
                     "sha256": hashlib.sha256(f).hexdigest(), "blocks": [list(b) for b in helpers.parse_frame_blocks(f)]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zstd_l3_multiblock_golden.json")
    with open(path, "w") as fh:
        json.dump({"libzstd": "1.5.7", "level": 3, "rows": rows}, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes", len(rows), "frames")


if __name__ == "__main__":
    main()
