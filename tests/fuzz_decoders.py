#!/usr/bin/env python3
"""Mutation fuzzer for the two decoders (TEST INFRASTRUCTURE): the zstd frame decoder and inflate, run as kernel bodies on the
CPU wave emulator with the input and the output placed against PROT_NONE guard pages, so that a read or write outside
the buffers the C ABI documents kills the process instead of passing silently (GPU AddressSanitizer is not available on
the pool; this is the CPU-side stand-in).  Every mutated frame is also decoded by the binary reference library (libzstd
1.5.7 / zlib through Python): a frame this decoder accepts must decode to the same bytes there; a frame it rejects may
be valid only if the status says "unsupported" (or, for zstd, if libzstd turns the damaged frame into other bytes than
the original: its fast Huffman loop skips the end-of-stream check this decoder makes).

    python tests/fuzz_decoders.py --which zstd --iters 2000 --seed 1
    python tests/fuzz_decoders.py --which inflate --iters 2000 --seed 1

Prints one line per 100 cases and "FUZZ OK <cases> ..." at the end; on a mismatch prints the case (seed, mutation, hex of
the frame when short) and exits 1.  A crash (guard page hit) or a hang shows as the missing "FUZZ OK" line: rerun with the
same seed and --verbose to see the last case started.
"""
import argparse
import ctypes
import mmap
import os
import random
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers                           # noqa: E402
from kompressor_amd import corpus        # noqa: E402

PAGE = 4096
IN_SLACK = 0         # the decoders may not read a single byte past the end of an entry
OUT_SLACK = 0        # nothing may be written past d_out_cap


class Guarded:
    """size usable bytes that END exactly `slack` bytes before a PROT_NONE page, and start right after one."""
    _libc = ctypes.CDLL(None, use_errno=True)

    def __init__(self, size, slack):
        body = (size + slack + PAGE - 1) // PAGE * PAGE
        self.map = mmap.mmap(-1, body + 2 * PAGE)
        self.base = ctypes.addressof(ctypes.c_char.from_buffer(self.map))
        assert self.base % PAGE == 0
        for a in (self.base, self.base + PAGE + body):
            if self._libc.mprotect(ctypes.c_void_p(a), PAGE, 0) != 0:
                raise OSError(ctypes.get_errno(), "mprotect")
        self.size = size
        self.addr = self.base + PAGE + body - slack - size          # first usable byte
        self.off = self.addr - self.base

    def write(self, data):
        self.map[self.off:self.off + len(data)] = data

    def read(self, n):
        return bytes(self.map[self.off:self.off + n])

    def close(self):
        for a in (self.base, self.base + len(self.map) - PAGE):
            self._libc.mprotect(ctypes.c_void_p(a), PAGE, 3)
        # the ctypes view keeps the mmap exported; dropping both is enough (no explicit close with live exports)
        self.map = None


def mutate(rng, f, others):
    f = bytearray(f)
    kind = rng.randrange(8)
    if not f:
        return bytes(f), "empty"
    if kind == 0:
        for _ in range(rng.randrange(1, 4)):
            i = rng.randrange(len(f)); f[i] ^= 1 << rng.randrange(8)
        what = "bitflip"
    elif kind == 1:
        for _ in range(rng.randrange(1, 3)):
            f[rng.randrange(len(f))] = rng.randrange(256)
        what = "byteset"
    elif kind == 2:
        f = f[:rng.randrange(len(f))]; what = "truncate"
    elif kind == 3:
        i = rng.randrange(len(f)); j = min(len(f), i + rng.randrange(1, 9)); del f[i:j]; what = "delete"
    elif kind == 4:
        i = rng.randrange(len(f)); f[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9))); what = "insert"
    elif kind == 5:
        o = rng.choice(others); i = rng.randrange(len(f)); j = rng.randrange(len(o)) if o else 0
        f = f[:i] + bytearray(o[j:]); what = "splice"
    elif kind == 6:
        # header region: the first 16 bytes decide sizes, windows and table modes
        i = rng.randrange(min(len(f), 16)); f[i] = rng.randrange(256); what = "header"
    else:
        i = rng.randrange(len(f)); f[i] = (f[i] + rng.choice((1, 255, 128))) & 255; what = "nudge"
    return bytes(f), what


def sources(rng, count):
    out = []
    sizes = [0, 1, 5, 40, 200, 700, 3000, 9000, 20000, 70000]
    for i in range(count):
        S = rng.choice(sizes) + rng.randrange(0, 64)
        mix = rng.choice("TXSBDIZR")
        d = corpus.make(90000 + i, 1, S, mix=ord(mix)).tobytes() if S else b""
        out.append(d)
    out.append(corpus.make(77, 1, 140000, mix=ord("T")).tobytes())      # two blocks
    out.append(bytes(300000))                                             # RLE blocks
    return out


def run_zstd(args):
    z = helpers.live_libzstd()
    if z is None:
        print("FUZZ SKIP no libzstd 1.5.7 here"); return 0
    rng = random.Random(args.seed)
    srcs = sources(rng, 24)
    frames = []
    dicts = []
    if args.dict:
        # frames made with dictionaries in zstd's own format (trained ones from the fixture, built ones with incomplete tables): the
        # decoder starts from the dictionary's Huffman weights, FSE tables and repeat offsets, and matches reach into its content
        cases = helpers.formatted_dict_cases()
        for ci in rng.sample(range(len(cases)), min(6, len(cases))):
            name, dd, inputs, _ = cases[ci]
            dicts.append(dd)
            for d in rng.sample(inputs, 8) + [b""]:
                for lvl in (1, 3, 7, 19):
                    frames.append((z.compress_with_dict(d, dd, lvl), d, len(dicts) - 1))
        # ... and frames that name no dictionary, or another one, decoded with a dictionary loaded
        for d in srcs[:6]:
            frames.append((z.compress(d, 3), d, 0))
    else:
        for k, d in enumerate(srcs):
            for lvl in ((1, 3, 6, 19) if len(d) < 100000 else (3,)):
                frames.append((z.compress(d, lvl), d, -1))
    emu = helpers.emu()
    emu.emu_zstd_decompress_dict.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint32, ctypes.c_uint32] + [ctypes.c_void_p] * 5 + [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32]
    stats = {"cases": 0, "accepted": 0, "rejected": 0, "both_valid_same": 0}
    only = [f for f, _, _ in frames]
    for it in range(args.iters):
        f0, d, di = rng.choice(frames)
        if args.dict and rng.random() < 0.05:
            di = rng.randrange(len(dicts))                  # the wrong dictionary: "Dictionary mismatch", or other content with ID 0
        dd = dicts[di] if di >= 0 else None
        f, what = (f0, "intact") if it % 50 == 0 else mutate(rng, f0, only)
        cap = len(d) + rng.choice((0, 0, 0, 1, 64, 5000)) if rng.random() < 0.85 else rng.randrange(0, len(d) + 1)
        if args.verbose:
            print(f"case {it} {what} len {len(f)} cap {cap}", flush=True)
        gin = Guarded(max(len(f), 1), IN_SLACK); gin.write(f)
        gout = Guarded(max(cap, 1), OUT_SLACK)
        in_off = np.array([gin.off], dtype=np.uint64); in_len = np.array([len(f)], dtype=np.uint32)
        out_off = np.array([gout.off], dtype=np.uint64); out_cap = np.array([cap], dtype=np.uint32)
        olen = np.zeros(1, dtype=np.uint32); st = np.zeros(1, dtype=np.uint32)
        r = emu.emu_zstd_decompress_dict(gin.base, helpers._vp(in_off), helpers._vp(in_len), 1, 1, gout.base, helpers._vp(out_off),
                                         helpers._vp(out_cap), helpers._vp(olen), helpers._vp(st), 128 * 1024 + 64, dd, len(dd) if dd else 0)
        assert r == 0, f"emulator reported {r} (case {it}, {what})"
        mine = gout.read(int(olen[0])) if st[0] == 0 else None
        gin.close(); gout.close()
        try:
            ref = z.decompress_with_dict(f, cap, dd) if dd is not None else z.decompress(f, cap)
            err = None
        except RuntimeError as e:
            ref, err = None, str(e)
        stats["cases"] += 1
        bad = None
        if st[0] == 0:
            stats["accepted"] += 1
            if ref is None:
                bad = f"accepted a frame libzstd rejects ({err})"
            elif ref != mine:
                bad = "decoded bytes differ from libzstd's"
            else:
                stats["both_valid_same"] += 1
        else:
            stats["rejected"] += 1
            # 14 = unsupported (reserved header bit ...).  libzstd accepts some damaged Huffman streams that RFC 8878 calls
            # faulty ("not entirely and exactly consumed"): its fast loop (1.5.x, table log 11, x86-64) does not test the
            # end of the four literal streams, and its double-symbol decoder (every version) clamps the bit count of a
            # stream's last symbol, swallowing up to a symbol's worth of left-over bits (checked case: 5 bits left after
            # the 2047th symbol of a stream; libzstd 1.4.8 and 1.5.7 both decode it, to other bytes than the original).
            # This decoder checks the exact end and says 20.  A damaged frame is only a finding when libzstd still
            # restores the ORIGINAL content from it.
            if ref is not None and int(st[0]) == 20 and ref != d:
                stats["stricter"] = stats.get("stricter", 0) + 1
            elif ref is not None and int(st[0]) not in (14,):
                bad = f"rejected (status {int(st[0])}) a frame libzstd decodes to {len(ref)} bytes" + (" (the original content)" if ref == d else "")
        if bad:
            print(f"FUZZ MISMATCH case {it} seed {args.seed} mutation {what} frame_len {len(f)} cap {cap}: {bad}")
            if len(f) <= 400:
                print("frame hex:", f.hex())
            return 1
        if (it + 1) % 100 == 0:
            print(f"{it + 1} cases: {stats}", flush=True)
    print(f"FUZZ OK {stats['cases']} zstd cases: {stats}")
    return 0


def run_inflate(args):
    rng = random.Random(args.seed)
    srcs = [d for d in sources(rng, 24) if len(d) <= 65536]
    streams = []
    for d in srcs:
        for lvl in (1, 6, 9):
            for fmt, wbits in ((0, -15), (1, 15), (2, 31)):
                c = zlib.compressobj(lvl, zlib.DEFLATED, wbits, 8)
                streams.append((c.compress(d) + c.flush(), d, fmt))
        c = zlib.compressobj(0, zlib.DEFLATED, -15)                  # stored blocks
        streams.append((c.compress(d) + c.flush(), d, 0))
    emu = helpers.emu()
    fn = emu.emu_inflate
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint32] + [ctypes.c_void_p] * 5 + [ctypes.c_uint32]
    if args.pre:
        # the two-kernel path: k_inflate_predecode (a lane per stream) + k_inflate_exec, staging made for 64 KiB slices
        pre = emu.emu_inflate_pre
        pre.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint32] + [ctypes.c_void_p] * 5 + [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        cov = np.zeros(1, dtype=np.uint32)
        fn = lambda *a: (pre(*a, 65536, helpers._vp(cov)), stats.__setitem__("covered", stats.get("covered", 0) + int(cov[0])))[0]    # noqa: E731
    stats = {"cases": 0, "accepted": 0, "rejected": 0}
    only = [s for s, _, _ in streams]
    for it in range(args.iters):
        s0, d, fmt = rng.choice(streams)
        s, what = (s0, "intact") if it % 50 == 0 else mutate(rng, s0, only)
        use_fmt = fmt if rng.random() < 0.7 or fmt == 0 else 3        # 3 = auto-detect zlib / gzip
        cap = len(d) + rng.choice((0, 0, 0, 1, 64)) if rng.random() < 0.85 else rng.randrange(0, len(d) + 1)
        if args.verbose:
            print(f"case {it} {what} len {len(s)} cap {cap} fmt {use_fmt}", flush=True)
        gin = Guarded(max(len(s), 1), IN_SLACK); gin.write(s)
        gout = Guarded(max(cap, 1), OUT_SLACK)
        in_off = np.array([gin.off], dtype=np.uint64); in_len = np.array([len(s)], dtype=np.uint32)
        out_off = np.array([gout.off], dtype=np.uint64); out_cap = np.array([cap], dtype=np.uint32)
        olen = np.zeros(1, dtype=np.uint32); st = np.zeros(1, dtype=np.uint32)
        r = fn(gin.base, helpers._vp(in_off), helpers._vp(in_len), 1, gout.base, helpers._vp(out_off), helpers._vp(out_cap),
               helpers._vp(olen), helpers._vp(st), use_fmt)
        assert r == 0, f"emulator reported {r} (case {it}, {what})"
        mine = gout.read(int(olen[0])) if st[0] == 0 else None
        gin.close(); gout.close()
        wb = {0: -15, 1: 15, 2: 31, 3: 47}[use_fmt]
        try:
            o = zlib.decompressobj(wb)
            ref = o.decompress(s, cap + 1)
            if not o.eof or len(ref) > cap or o.unused_data:
                raise zlib.error("incomplete, too large or followed by other bytes")
            err = None
        except zlib.error as e:
            ref, err = None, str(e)
        stats["cases"] += 1
        bad = None
        if st[0] == 0:
            stats["accepted"] += 1
            if ref is None:
                bad = f"accepted a stream zlib rejects ({err})"
            elif ref != mine:
                bad = "decoded bytes differ from zlib's"
        else:
            stats["rejected"] += 1
            if ref is not None:
                bad = f"rejected (status {int(st[0])}) a stream zlib decodes to {len(ref)} bytes"
        if bad:
            print(f"FUZZ MISMATCH case {it} seed {args.seed} mutation {what} stream_len {len(s)} cap {cap} fmt {use_fmt}: {bad}")
            if len(s) <= 400:
                print("stream hex:", s.hex())
            return 1
        if (it + 1) % 100 == 0:
            print(f"{it + 1} cases: {stats}", flush=True)
    print(f"FUZZ OK {stats['cases']} inflate cases: {stats}")
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", choices=("zstd", "inflate"), default="zstd")
    ap.add_argument("--pre", action="store_true", help="inflate: through the pre-decoder + executor kernels instead of k_inflate alone")
    ap.add_argument("--dict", action="store_true", help="zstd: frames made with dictionaries in zstd's own format, decoded with the dictionary")
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    sys.exit(run_zstd(args) if args.which == "zstd" else run_inflate(args))


if __name__ == "__main__":
    main()
