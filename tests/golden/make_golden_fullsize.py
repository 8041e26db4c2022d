#!/usr/bin/env python3
"""Generates tests/golden/fullsize_golden.json: BASELINE.json's configs at their FULL size, as one sha256 per group of
4 096 slices over the concatenated frames (plus the group's total frame bytes), so that the driver-run GPU suite can
compare all 65 536 frames of configs[1] (zstd level 3, libzstd 1.5.7), all 65 536 streams of configs[4] (raw DEFLATE
level 6, this machine's zlib) and all 131 072 slices of rank 0's configs[3] block (text / binary alternating: 1 Mi slices over
8 ranks) byte for byte without carrying the frames.

Run in the build container only (needs the binary libzstd 1.5.7 that oracle/libzstd_ref.py finds):

    python tests/golden/make_golden_fullsize.py
"""
import hashlib
import json
import os
import sys
import zlib
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
HERE = os.path.dirname(os.path.abspath(__file__))
S = 65536
GROUP = 4096


def _group(args):
    codec, mix, g = args
    from kompressor_amd import corpus
    buf = corpus.make(g * GROUP, GROUP, S, mix, threads=1)
    h = hashlib.sha256()
    total = 0
    if codec == "zstd3":
        from libzstd_ref import LibZstd
        z = LibZstd()
        assert z.lib.ZSTD_versionNumber() == 10507
        for i in range(GROUP):
            f = z.compress(buf[i * S:(i + 1) * S].tobytes())
            h.update(f)
            total += len(f)
    else:
        for i in range(GROUP):
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0)
            f = c.compress(buf[i * S:(i + 1) * S].tobytes()) + c.flush()
            h.update(f)
            total += len(f)
    return [g, total, h.hexdigest()]


def main():
    from kompressor_amd import corpus
    jobs = [("zstd3", corpus.MIX_CONFIG1, g) for g in range(16)] + [("deflate6", corpus.MIX_CONFIG1, g) for g in range(16)] + \
           [("zstd3", corpus.MIX_TEXT_BINARY, g) for g in range(32)]
    with ProcessPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(_group, jobs))
    out = {"slice_bytes": S, "group": GROUP, "libzstd": "1.5.7", "zlib": zlib.ZLIB_RUNTIME_VERSION,
           "what": "per group g of 4096 slices (indices g*4096 ..): [g, total frame bytes, sha256 of the frames back to back]",
           "config1_zstd3": res[:16], "config4_deflate6": res[16:32], "config3_zstd3_first_16384": res[32:36], "config3_zstd3_rank0": res[32:]}
    path = os.path.join(HERE, "fullsize_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
