"""Shared test plumbing: seeded/special inputs, the oracle (CPU restatement,
oracle/zstd_l3_ref.c), the optional live libzstd 1.5.7, and the CPU wave
emulator build of the kernel bodies (tests/emu)."""
import ctypes
import hashlib
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_PATH = os.path.join(ROOT, "tests", "golden", "zstd_l3_golden.json")


def special_inputs():
    """Hand-made edge inputs (same dict tests/golden/make_golden.py used)."""
    return {
        "empty": b"",
        "one_byte": b"a",
        "hello": b"hello compression world",
        "zeros_64k": bytes(65536),
        "zeros_128k": bytes(131072),
        "ab_64k": (b"ab" * 32768),
        "abc_100": (b"abc" * 34)[:100],
        "ramp_64k": bytes(range(256)) * 256,
        "long_literals_then_match": bytes((i * 7 + (i >> 8) * 13) & 0xFF for i in range(70000)) + bytes(1000),
        "all_same_but_last": bytes(65535) + b"\x01",
        "two_symbols": bytes((0x41 if (i * 2654435761) & 0x10000 else 0x42) for i in range(65536)),
    }


def multiblock_inputs():
    """Seeded inputs above 128 KiB (multi-block frames): (name, bytes).  Same list the generator
    tests/golden/make_golden_multiblock.py used.  Segments of different corpus classes, byte runs and short
    periods, so that libzstd's block pre-splitter cuts blocks and emits raw / RLE / treeless-literals blocks."""
    import random
    from kompressor_amd import corpus
    out = [("zeros_1m", bytes(1 << 20)), ("zeros_300k", bytes(300000)), ("run_128k_plus_5", b"\x07" * (131072 + 5)),
           ("period256_1m", bytes(range(256)) * 4096),
           ("random_1m", corpus.make(9001, 1, 1 << 20, mix=ord("R")).tobytes()),
           ("random_zero_alternating", corpus.make(9002, 1, 131072, mix=ord("R")).tobytes() + bytes(131072)
            + corpus.make(9003, 1, 131072, mix=ord("R")).tobytes() + bytes(70000))]
    for t in range(1, 9):
        out.append((f"text_256k_plus_{t}", corpus.make(500 + t, 1, 262144 + t, mix=ord("T")).tobytes()))
    for cls in "TXSBDIZ":
        out.append((f"class_{cls}_1m", corpus.make(9100 + ord(cls), 1, 1 << 20, mix=ord(cls)).tobytes()))
    rng = random.Random(20260517)
    for k in range(24):
        total = rng.choice([131073, 140000, 262144, 262145, 300000, 524288, 1 << 20, 777777, 1500000, 2 << 20])
        parts, have = [], 0
        while have < total:
            cls = rng.choice("TXSBDIZR")
            n = rng.choice([100, 700, 1000, 3000, 9000, 20000, 60000, 140000])
            r = rng.random()
            if r < 0.1:
                seg = bytes([rng.randrange(256)]) * n
            elif r < 0.2:
                unit = corpus.make(rng.randrange(1 << 30), 1, rng.choice([10, 300]), mix=ord("R")).tobytes()
                seg = unit * (n // len(unit) + 1)
            else:
                seg = corpus.make(rng.randrange(1 << 30), 1, n, mix=ord(cls)).tobytes()
            parts.append(seg)
            have += len(seg)
        out.append((f"mixed_{k}_{total}", b"".join(parts)[:total]))
    return out


def beyond_window_inputs():
    """Seeded inputs longer than the level-3 window (2 MiB) and than libzstd's staging buffer (2 MiB + 128 KiB): sizes
    around the window, the buffer's wrap points (multiples of 17 x 128 KiB) and chunk boundaries; pieces of every class,
    byte runs and copies of earlier pieces at any distance, so that matches and repcodes straddle the window limit and
    the segment boundaries.  (name, bytes); same list tests/golden/make_golden_buffered.py used."""
    import random
    from kompressor_amd import corpus
    lap = 17 * 131072
    rng = random.Random(20261004)

    def build(n):
        out = bytearray()
        while len(out) < n:
            r = rng.random()
            if r < 0.3 and len(out) > 1000:
                a = rng.randrange(0, len(out))
                out += out[a:a + rng.randrange(10, 300000)]
            elif r < 0.35:
                out += bytes([rng.randrange(256)]) * rng.randrange(1, 300000)
            else:
                out += corpus.make(rng.randrange(1 << 30), 1, rng.randrange(1000, 400000), mix=ord(rng.choice("TXSBDIZR"))).tobytes()
        return bytes(out[:n])

    sizes = [(2 << 20) + 1, (2 << 20) + 65536, lap - 1, lap, lap + 1, lap + 131072 + 7, 20 * 131072 + 2, 3000000, 2 * lap - 3, 2 * lap + 50000, 5000000, 3 * lap + 4097]
    return [(f"beyond_{n}", build(n)) for n in sizes]


def fast_window_inputs():
    """Seeded inputs longer than the windows of the "fast" levels (512 KiB at level 1 and the negative levels, 1 MiB at level 2)
    and than libzstd's staging buffers for them (window + 128 KiB: 5 and 9 chunks): sizes around the windows, the wrap points and
    chunk boundaries; pieces of every class, byte runs and copies of earlier pieces at any distance.  (name, level, bytes); the
    list tests/golden/make_golden_fast_window.py used."""
    import random
    from kompressor_amd import corpus
    rng = random.Random(20261006)

    def build(n):
        out = bytearray()
        while len(out) < n:
            r = rng.random()
            if r < 0.3 and len(out) > 1000:
                a = rng.randrange(0, len(out))
                out += out[a:a + rng.randrange(10, 300000)]
            elif r < 0.35:
                out += bytes([rng.randrange(256)]) * rng.randrange(1, 200000)
            else:
                out += corpus.make(rng.randrange(1 << 30), 1, rng.randrange(1000, 300000), mix=ord(rng.choice("TXSBDIZR"))).tobytes()
        return bytes(out[:n])

    rows = []
    for level in (1, -1, -5, 2):
        win = (1 << 20) if level == 2 else (1 << 19)
        lap = win + 131072
        sizes = [win + 1, lap - 1, lap, lap + 1, lap + 131072 + 7, 2 * lap - 3, 2 * lap + 50000, 3 * lap + 4097] if level in (1, 2) else [win + 9, lap, lap + 70001, 2 * lap + 50000]
        rows += [(f"l{level}_{n}", level, build(n)) for n in sizes]
    return rows


def fast_window_golden():
    with open(os.path.join(ROOT, "tests", "golden", "zstd_fast_window_golden.json")) as f:
        return json.load(f)


def deflate_long_inputs():
    """Seeded DEFLATE inputs above 64 KiB (zlib's window slides many times): sizes around the slide points
    (k x 32 KiB + 32 506), the reference's own round-trip size (1 MiB + 3 random bytes, ZlibTest.kt:16,28-33), runs, far
    copies.  (name, bytes); same list tests/golden/make_golden_deflate.py used."""
    import random
    from kompressor_amd import corpus
    rng = random.Random(20261005)

    def build(n):
        out = bytearray()
        while len(out) < n:
            r = rng.random()
            if r < 0.25 and len(out) > 1000:
                a = rng.randrange(0, len(out))
                out += out[a:a + rng.randrange(10, 100000)]
            elif r < 0.32:
                out += bytes([rng.randrange(256)]) * rng.randrange(1, 100000)
            else:
                out += corpus.make(rng.randrange(1 << 30), 1, rng.randrange(100, 200000), mix=ord(rng.choice("TXSBDIZR"))).tobytes()
        return bytes(out[:n])

    sizes = [65537, 65274 + 262, 98042, 98043, 131072, 200000, 300001, 32768 * 9 + 32506, 777777, 1 << 20]
    out = [(f"long_{n}", build(n)) for n in sizes]
    out.append(("random_1m_plus_3", np.random.default_rng(1).integers(0, 256, (1 << 20) + 3, dtype=np.uint8).tobytes()))
    out.append(("zeros_1m", bytes(1 << 20)))
    return out


def buffered_golden():
    with open(os.path.join(os.path.dirname(GOLDEN_PATH), "zstd_l3_buffered_golden.json")) as fh:
        return json.load(fh)


def parse_frame_blocks(f):
    """[(block type, header size field, literals type or -1)] of a zstd frame (RFC 8878 section 3.1.1)."""
    fhd = f[4]
    pos = 5
    ss = (fhd >> 5) & 1
    if not ss:
        pos += 1
    pos += [1 if ss else 0, 2, 4, 8][fhd >> 6]
    out = []
    while True:
        h = f[pos] | (f[pos + 1] << 8) | (f[pos + 2] << 16)
        pos += 3
        t, sz = (h >> 1) & 3, h >> 3
        out.append((t, sz, (f[pos] & 3) if t == 2 else -1))
        pos += 1 if t == 1 else sz
        if h & 1:
            break
    return out


def stream_cases():
    """(data, cut points) for streaming frames: the pieces data[cuts[i]:cuts[i+1]] are fed with finish = false, the last one
    with finish = true.  Uses the multi-block inputs (sizes 128 KiB+1 .. 2 MiB) and a few smaller ones; cut points seeded."""
    import random
    rng = random.Random(424242)
    out = []
    datas = [d for _, d in multiblock_inputs()][:30] + [special_inputs()[k] for k in ("hello", "ramp_64k", "zeros_128k", "two_symbols")]
    for d in datas:
        n = len(d)
        r = rng.random()
        if r < 0.3:
            cuts = [0, n, n]
        elif r < 0.6:
            cuts = [0, rng.randrange(1, n), n]
        else:
            cuts = sorted(set([0, n] + [rng.randrange(1, n) for _ in range(rng.randrange(2, 6))]))
        out.append((d, cuts))
    return out


def level4_golden():
    """libzstd 1.5.7 at level 4 where it is the double-fast parse with one block (tests/golden/make_golden_level4.py)."""
    with open(os.path.join(os.path.dirname(GOLDEN_PATH), "zstd_level4_golden.json")) as fh:
        return json.load(fh)


def neg_levels_golden():
    """libzstd 1.5.7 at negative levels (tests/golden/make_golden_neg_levels.py); sha256 cut to 32 hex digits."""
    with open(os.path.join(os.path.dirname(GOLDEN_PATH), "zstd_neg_levels_golden.json")) as fh:
        return json.load(fh)


def levels_golden():
    with open(os.path.join(os.path.dirname(GOLDEN_PATH), "zstd_levels_golden.json")) as fh:
        return json.load(fh)


def level2_big_golden():
    with open(os.path.join(os.path.dirname(GOLDEN_PATH), "zstd_level2_big_golden.json")) as fh:
        return json.load(fh)


def multiblock_golden():
    with open(os.path.join(os.path.dirname(GOLDEN_PATH), "zstd_l3_multiblock_golden.json")) as fh:
        return json.load(fh)


def fullsize_golden():
    with open(os.path.join(os.path.dirname(GOLDEN_PATH), "fullsize_golden.json")) as fh:
        return json.load(fh)


def golden():
    with open(GOLDEN_PATH) as fh:
        return json.load(fh)


def sha256(b):
    return hashlib.sha256(b).hexdigest()


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


# ---------------------------------------------------------------- oracle ----
_ORACLE = None


def build_oracle():
    src = os.path.join(ROOT, "oracle", "zstd_l3_ref.c")
    out_dir = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libkref.so")
    if _newer(lib, [src]):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-fvisibility=hidden", "-o", lib, src], check=True)
    return lib


class Oracle:
    """ctypes view of the C restatement (test infrastructure only)."""

    def __init__(self):
        k = self.lib = ctypes.CDLL(build_oracle())
        k.kref_zstd_l3_compress.restype = ctypes.c_size_t
        k.kref_zstd_l3_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        k.kref_compress_bound.restype = ctypes.c_size_t
        k.kref_compress_bound.argtypes = [ctypes.c_size_t]
        k.kref_zstd_l3_seqstore.restype = ctypes.c_size_t
        k.kref_params_l3.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]

    def compress(self, d: bytes) -> bytes:
        cap = self.lib.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap)
        n = self.lib.kref_zstd_l3_compress(o, cap, d, len(d))
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle: input outside the restatement's scope")
        return o.raw[:n]

    def compress_dict(self, d: bytes, dictionary: bytes):
        """Frame of ZstdCompressor(3, dictionary) (raw-content dictionary); returns (frame, attached)."""
        k = self.lib
        k.kref_zstd_l3_compress_dict.restype = ctypes.c_size_t
        k.kref_zstd_l3_compress_dict.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                                 ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
        cap = k.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap)
        mode = ctypes.c_int(-1)
        n = k.kref_zstd_l3_compress_dict(o, cap, d, len(d), dictionary, len(dictionary), ctypes.byref(mode))
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle: input outside the restatement's scope")
        return o.raw[:n], bool(mode.value)

    def compress_level(self, d: bytes, level: int) -> bytes:
        """Frame at level 1 or 2 or at a negative level (strategy "fast"), or 4 where it is the double-fast parse (16 KiB < size <=
        128 KiB, and above 256 KiB ZSTD_compress2's frame)."""
        k = self.lib
        if level == 4:
            k.kref_zstd_l4_compress.restype = ctypes.c_size_t
            k.kref_zstd_l4_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            cap = k.kref_compress_bound(len(d)) + 64
            o = ctypes.create_string_buffer(cap)
            n = k.kref_zstd_l4_compress(o, cap, d, len(d))
            if n == 2 ** 64 - 1:
                raise RuntimeError("oracle: level 4 has no double-fast row for this size")
            return o.raw[:n]
        k.kref_zstd_fast_compress.restype = ctypes.c_size_t
        k.kref_zstd_fast_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        cap = k.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap)
        n = k.kref_zstd_fast_compress(o, cap, d, len(d), level)
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle: input outside the restatement's scope")
        return o.raw[:n]

    def compress_stream(self, d: bytes, empty_end: bool) -> bytes:
        """Frame of a stream fed with finish = false calls and closed with finish = true (unknown size while compressing);
        empty_end: the closing call brought no data."""
        k = self.lib
        k.kref_zstd_l3_compress_stream.restype = ctypes.c_size_t
        k.kref_zstd_l3_compress_stream.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        cap = k.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap)
        n = k.kref_zstd_l3_compress_stream(o, cap, d, len(d), 1 if empty_end else 0)
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle: input outside the restatement's scope")
        return o.raw[:n]

    def compress_buffered(self, d: bytes, known_size: bool = True, empty_end: bool = False, out_chunk=None, tail_direct: int = 0, level: int = 3) -> bytes:
        """The frame ZstdCompressor(3) really produces above 128 KiB (libzstd stages the input in chunks of 128 KiB because
        the reference's output slices are smaller than ZSTD_compressBound): known_size = finish = true from the first
        call, out_chunk = the driver's output slice size (known_size = 2: no staging, ZSTD_compress2 into a bound-sized buffer,
        any length); False = a stream fed with finish = false first (tail_direct: the
        bytes its closing call brought, when they arrived on an empty staging buffer with room for their bound).
        Any length (the window slides)."""
        k = self.lib
        fn = k.kref_zstd_l4_compress_buffered if level == 4 else k.kref_zstd_l3_compress_buffered      # (level 4: its double-fast rows)
        fn.restype = ctypes.c_size_t
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t]
        cap = k.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap)
        if out_chunk is None:
            out_chunk = max(8192, len(d) // 10)              # SliceTransform.kt:47-56 getOutput
        n = fn(o, cap, d, len(d), int(known_size), 1 if empty_end else 0, out_chunk, tail_direct)
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle: input outside the restatement's scope")
        return o.raw[:n]

    def compress_level_big(self, d: bytes, level: int, stream: bool = False, empty_end: bool = False) -> bytes:
        """Level 1 or 2 frame of several blocks (any size the window holds): stream False / 0 = ZSTD_compress2's frame,
        True / 1 = a stream's, 3 = the one-shot frame the reference's driver gets (input staged in 128 KiB chunks)."""
        k = self.lib
        k.kref_zstd_fast_compress_big.restype = ctypes.c_size_t
        k.kref_zstd_fast_compress_big.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        cap = k.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap)
        n = k.kref_zstd_fast_compress_big(o, cap, d, len(d), level, int(stream), 1 if empty_end else 0)
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle: input outside the restatement's scope")
        return o.raw[:n]

    def compress_fast_buffered(self, d: bytes, level: int, stream=0, empty_end: bool = False, out_chunk: int = 0, tail_direct: int = 0) -> bytes:
        """Levels 1, 2 and the negative ones at ANY length (round 4: beyond the level's window libzstd's staging buffer wraps and
        the blocks are parsed by ZSTD_compressBlock_fast_extDict): stream 0 = ZSTD_compress2's frame, 1 / 2 = a stream's (closed with /
        without data), 3 = the one-shot frame the reference's driver gets (out_chunk 0 = max(8192, n / 10))."""
        k = self.lib
        k.kref_zstd_fast_compress_buffered.restype = ctypes.c_size_t
        k.kref_zstd_fast_compress_buffered.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t]
        cap = k.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap)
        n = k.kref_zstd_fast_compress_buffered(o, cap, d, len(d), level, int(stream), 1 if empty_end else 0, out_chunk, tail_direct)
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle: input outside the restatement's scope")
        return o.raw[:n]

    def build_dictionary(self, dict_id: int, lit_sample: bytes, ll, of, ml, rep, content: bytes) -> bytes:
        """A dictionary in zstd's own format put together from chosen statistics (kref_build_dictionary: test infrastructure): the
        literal sample's Huffman table, three code histograms taken as they are (zero counts stay zero), repeat offsets, content."""
        k = self.lib
        k.kref_build_dictionary.restype = ctypes.c_size_t
        k.kref_build_dictionary.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        cap = len(content) + 4096
        out = ctypes.create_string_buffer(cap)
        arr = lambda v, n: (ctypes.c_uint32 * n)(*v)
        n = k.kref_build_dictionary(out, cap, dict_id, lit_sample, len(lit_sample), arr(ll, 36), arr(of, 32), arr(ml, 53), arr(rep, 3), content, len(content))
        if n == 2 ** 64 - 1:
            raise RuntimeError("oracle: the statistics do not fit the dictionary format")
        return out.raw[:n]

    def compress_lazy(self, d: bytes, level: int):
        """Frame of ZstdCompressor(level) at levels 5 .. 10 (4 .. 8 up to 16 KiB): strategies greedy / lazy / lazy2; None where the level is
        another strategy at this size."""
        k = self.lib
        k.kref_zstd_lazy_compress.restype = ctypes.c_size_t
        k.kref_zstd_lazy_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        cap = k.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap)
        n = k.kref_zstd_lazy_compress(o, cap, d, len(d), level)
        return None if n == 2 ** 64 - 1 else o.raw[:n]

    def compress_lazy_big(self, d: bytes, level: int):
        """Frame of ZstdCompressor(level) at levels 4 .. 10 for 128 KiB < len(d) <= 2 MiB (frames of several blocks; the lazy parsers' state,
        the previous block's tables and the strategies' pre-splitter carried from block to block) -> (frame, block sizes); None where the
        level is double-fast at this size (level 4 above 256 KiB).  No product path yet: the oracle of the next row of SURVEY 8f."""
        k = self.lib
        k.kref_zstd_lazy_compress_big.restype = ctypes.c_size_t
        k.kref_zstd_lazy_compress_big.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        cap = k.kref_compress_bound(len(d)) + 64
        o = ctypes.create_string_buffer(cap); nb = ctypes.c_uint32(0); bs = (ctypes.c_uint32 * (len(d) // 8192 + 4))()
        n = k.kref_zstd_lazy_compress_big(o, cap, d, len(d), level, bs, ctypes.byref(nb))
        return None if n == 2 ** 64 - 1 else (o.raw[:n], list(bs)[:nb.value])

    def params(self, n):
        a = (ctypes.c_uint32 * 4)()
        self.lib.kref_params_l3(n, a)
        return tuple(a)


def oracle():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE


class DeflateOracle:
    """ctypes view of oracle/deflate_l6_ref.c (test infrastructure only)."""

    def __init__(self):
        src = os.path.join(ROOT, "oracle", "deflate_l6_ref.c")
        out_dir = os.path.join(ROOT, "oracle", "_build")
        os.makedirs(out_dir, exist_ok=True)
        lib = os.path.join(out_dir, "libdref.so")
        if _newer(lib, [src]):
            subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-fvisibility=hidden", "-o", lib, src], check=True)
        d = self.lib = ctypes.CDLL(lib)
        d.dref_deflate_l6_raw.restype = ctypes.c_size_t
        d.dref_deflate_l6_raw.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        d.dref_deflate_raw_level.restype = ctypes.c_size_t
        d.dref_deflate_raw_level.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        d.dref_deflate_raw_params.restype = ctypes.c_size_t
        d.dref_deflate_raw_params.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int]

    def compress(self, d: bytes, level: int = 6, window_bits: int = 15, mem_level: int = 8) -> bytes:
        """raw DEFLATE at level 1 .. 9 with deflateInit2's windowBits 9 .. 15 and memLevel 1 .. 9"""
        cap = len(d) + len(d) // 7 + 256
        o = ctypes.create_string_buffer(cap)
        n = self.lib.dref_deflate_raw_params(o, cap, d, len(d), level, window_bits, mem_level)
        if n == 2 ** 64 - 1:
            raise RuntimeError("deflate oracle: output did not fit")
        return o.raw[:n]


_DORACLE = None


def deflate_oracle():
    global _DORACLE
    if _DORACLE is None:
        _DORACLE = DeflateOracle()
    return _DORACLE


def deflate_golden():
    with open(os.path.join(ROOT, "tests", "golden", "deflate_l6_golden.json")) as fh:
        return json.load(fh)


def foreign_frames():
    """Frames of other zstd settings than the product's own (tests/golden/make_foreign_frames.py): [(row, frame, plain)]."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_foreign_frames import plain_of
    g = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(g, "foreign_frames.json")))
    blob = open(os.path.join(g, "foreign_frames.bin"), "rb").read()
    assert sha256(blob) == meta["blob_sha256"]
    out = []
    for r in meta["rows"]:
        plain = plain_of(r)
        assert sha256(plain) == r["plain_sha256"], r          # the corpus generator still makes the bytes the frame was made of
        out.append((r, blob[r["off"]: r["off"] + r["len"]], plain))
    return out


def live_libzstd():
    """A libzstd 1.5.7 found on this machine, or None (never required)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        from libzstd_ref import LibZstd
        return LibZstd()
    except Exception:
        return None


def require_live_libzstd(report=True):
    """For the -m gpu tests that check against the binary library of the machine they run on: the image of the GPU boxes
    carries libzstd 1.5.7 (the Pillow wheel's), so its absence there is a broken box or a changed image, not a reason to
    skip -- the test FAILS with a sentence, and the library's path goes into the test's output (pytest -rA / the junit
    record show that the differential leg ran, and against what)."""
    import pytest
    z = live_libzstd()
    if z is None:
        pytest.fail("no binary libzstd 1.5.7 on this GPU box: the differential tests against the live library cannot run "
                    "(oracle/libzstd_ref.py looks in the Pillow wheel's pillow.libs/); they are not skipped silently")
    if report:
        print(f"[live library] libzstd 1.5.7 at {z.path}")
    return z


# -------------------------------------------------------------- emulator ----
_EMU = None


def build_emu():
    emu = os.path.join(ROOT, "tests", "emu")
    csrc = os.path.join(ROOT, "kompressor_amd", "csrc")
    lib = os.path.join(emu, "libkxemu.so")
    srcs = [os.path.join(emu, f) for f in os.listdir(emu) if f.endswith((".cpp", ".h"))]
    srcs += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if _newer(lib, srcs):
        subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I" + emu, "-I" + csrc, "-o", lib,
                        os.path.join(emu, "emu_core.cpp"), os.path.join(emu, "emu_zstd.cpp")], check=True)
    return lib


def emu():
    global _EMU
    if _EMU is None:
        # KXEMU_LIB: another build of the emulator, e.g. one compiled with -fsanitize=undefined (tests/emu/README)
        _EMU = ctypes.CDLL(os.environ.get("KXEMU_LIB") or build_emu())
    return _EMU


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def compress_bound(n):
    return n + (n >> 8) + ((((128 << 10) - n) >> 11) if n < (128 << 10) else 0)


def emu_compress(datas, G=8, nblocks=2):
    """Run the match + entropy kernel bodies on the CPU wave emulator."""
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    pos = 0
    for i, d in enumerate(datas):
        offs[i] = pos
        pos += len(d)
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, d in enumerate(datas):
        buf[int(offs[i]):int(offs[i]) + len(d)] = np.frombuffer(d, dtype=np.uint8)
    cap = max([len(d) for d in datas] + [64])
    stride = (compress_bound(cap) + 64 + 15) & ~15
    out = np.zeros(n * stride, dtype=np.uint8)
    ooff = np.arange(n, dtype=np.uint64) * stride
    olen = np.zeros(n, dtype=np.uint32)
    r = emu().emu_zstd_compress(_vp(buf), _vp(offs), _vp(lens), n, G, nblocks, _vp(out), _vp(ooff), _vp(olen), cap)
    assert r == 0, f"emulator reported {r}"
    return [out[i * stride:i * stride + int(olen[i])].tobytes() for i in range(n)]


def emu_compress_level(datas, level, G=4, nblocks=2):
    """Levels 1 / 2 (strategy fast): fast match kernel + entropy kernel on the emulator."""
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    pos = 0
    for i, d in enumerate(datas):
        offs[i] = pos
        pos += len(d)
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, d in enumerate(datas):
        buf[int(offs[i]):int(offs[i]) + len(d)] = np.frombuffer(d, dtype=np.uint8)
    cap = max([len(d) for d in datas] + [64])
    stride = (compress_bound(cap) + 64 + 15) & ~15
    out = np.zeros(n * stride, dtype=np.uint8)
    ooff = np.arange(n, dtype=np.uint64) * stride
    olen = np.zeros(n, dtype=np.uint32)
    r = emu().emu_zstd_compress_level(_vp(buf), _vp(offs), _vp(lens), n, G, nblocks, _vp(out), _vp(ooff), _vp(olen), cap, level)
    assert r == 0, f"emulator reported {r}"
    return [out[i * stride:i * stride + int(olen[i])].tobytes() for i in range(n)]


def emu_compress_lazy(datas, level, nblocks=2):
    """zstd levels 5 .. 10 (and 4 .. 8 up to 16 KiB) on the emulator: k_zstd_lazy_sort + k_zstd_lazy + k_zstd_entropy bodies; a slice the
    level does not serve at its size comes back as b''."""
    n = len(datas)
    cap = max(max((len(d) for d in datas), default=1), 64)
    lens = np.array([len(d) for d in datas], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64); stride = (cap + 63) & ~63
    buf = np.zeros(n * stride + 64, dtype=np.uint8)
    for i, d in enumerate(datas):
        offs[i] = i * stride; buf[i * stride:i * stride + len(d)] = np.frombuffer(d, dtype=np.uint8)
    ostride = cap + (cap >> 7) + 1024
    out = np.zeros(n * ostride, dtype=np.uint8); ooff = (np.arange(n, dtype=np.uint64) * ostride); olen = np.zeros(n, dtype=np.uint32)
    r = emu().emu_zstd_compress_lazy(_vp(buf), _vp(offs), _vp(lens), n, nblocks, _vp(out), _vp(ooff), _vp(olen), cap, level)
    assert r == 0, f"emulated lazy kernels failed: {r}"
    return [out[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes() for i in range(n)]


def emu_compress_dict(datas, dictionary, G=4, nblocks=2):
    """Dictionary match kernel + entropy kernel on the emulator: frames of ZstdCompressor(3, dictionary)."""
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    pos = 0
    for i, d in enumerate(datas):
        offs[i] = pos
        pos += len(d)
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, d in enumerate(datas):
        buf[int(offs[i]):int(offs[i]) + len(d)] = np.frombuffer(d, dtype=np.uint8)
    cap = max([len(d) for d in datas] + [64])
    stride = (compress_bound(cap) + 64 + 15) & ~15
    out = np.zeros(n * stride, dtype=np.uint8)
    ooff = np.arange(n, dtype=np.uint64) * stride
    olen = np.zeros(n, dtype=np.uint32)
    dbuf = np.frombuffer(dictionary, dtype=np.uint8).copy()
    r = emu().emu_zstd_compress_dict(_vp(buf), _vp(offs), _vp(lens), n, G, nblocks, _vp(out), _vp(ooff), _vp(olen), cap, _vp(dbuf), len(dictionary))
    assert r == 0, f"emulator reported {r}"
    return [out[i * stride:i * stride + int(olen[i])].tobytes() for i in range(n)]


def emu_compress_big(datas, G=16, nblocks=2, by_rounds=False, stream=0, level=3, tail_or_chunk=0, wide=False):
    """Frames of several blocks (slices above 128 KiB) on the emulator: the product's one-wave-per-slice kernel body,
    or (by_rounds) the same steps as separate launches per round of blocks.  stream: 0 ZSTD_compress2's frames, 1 / 2
    streaming frames, 3 the reference's one-shot driver (tail_or_chunk: a stream's tail_direct / the driver's output
    slice size, 0 = max(8192, n / 10)); wide: table entries without check bits.  Returns (frames, rounds)."""
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    pos = 0
    for i, d in enumerate(datas):
        offs[i] = pos
        pos += (len(d) + 63) & ~63
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, d in enumerate(datas):
        buf[int(offs[i]):int(offs[i]) + len(d)] = np.frombuffer(d, dtype=np.uint8)
    cap = max([len(d) for d in datas] + [64])
    stride = (compress_bound(cap) + 1024 + 63) & ~63
    out = np.zeros(n * stride, dtype=np.uint8)
    ooff = np.arange(n, dtype=np.uint64) * stride
    olen = np.zeros(n, dtype=np.uint32)
    rounds = ctypes.c_uint32(0)
    r = emu().emu_zstd_compress_big_ex2(_vp(buf), _vp(offs), _vp(lens), n, G, nblocks, _vp(out), _vp(ooff), _vp(olen),
                                        ctypes.byref(rounds) if by_rounds else None, stream | ((level if level in (1, 2, 4) else 1 if level < 0 else 0) << 8) | (((1 - level) << 16) if level < 0 else 0),
                                        tail_or_chunk, 1 if wide else 0)
    assert r == 0, f"emulator reported {r}"
    return [out[i * stride:i * stride + int(olen[i])].tobytes() for i in range(n)], rounds.value


def emu_deflate(datas, zlib_wrapper=False, fmt=None, level=6, window_bits=15, mem_level=8, old_kernels=False):
    """chains -> best -> parse -> encode kernel bodies on the CPU wave emulator."""
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    pos = 0
    for i, d in enumerate(datas):
        offs[i] = pos
        pos += len(d)
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, d in enumerate(datas):
        buf[int(offs[i]):int(offs[i]) + len(d)] = np.frombuffer(d, dtype=np.uint8)
    big = max([len(d) for d in datas] + [65536])
    stride = (big + (big >> 12) + (big >> 14) + 64 + 63) & ~63
    if (window_bits, mem_level) != (15, 8):
        stride = (big + (big >> 3) + (big >> 6) + 64 + 63) & ~63
    out = np.zeros(n * stride, dtype=np.uint8)
    ooff = np.arange(n, dtype=np.uint64) * stride
    olen = np.zeros(n, dtype=np.uint32)
    r = emu().emu_deflate_params(_vp(buf), _vp(offs), _vp(lens), n, _vp(out), _vp(ooff), _vp(olen), None, None, fmt if fmt is not None else (1 if zlib_wrapper else 0), level,
                                 window_bits, mem_level, int(old_kernels))
    assert r == 0, f"emulator reported {r}"
    return [out[i * stride:i * stride + int(olen[i])].tobytes() for i in range(n)]


def emu_inflate(streams, caps, zlib_wrapper=False, fmt=None, pre=True, stage_bytes=65536):
    n = len(streams)
    lens = np.array([len(f) for f in streams], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    pos = 16
    for i, f in enumerate(streams):
        offs[i] = pos
        pos += (len(f) + 31) & ~15
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, f in enumerate(streams):
        buf[int(offs[i]):int(offs[i]) + len(f)] = np.frombuffer(f, dtype=np.uint8)
    caps = np.array(caps, dtype=np.uint32)
    ooff = np.zeros(n, dtype=np.uint64)
    t = 0
    for i in range(n):
        ooff[i] = t
        t += int(caps[i]) + 16
    out = np.zeros(t + 64, dtype=np.uint8)
    olen = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    f_ = fmt if fmt is not None else (1 if zlib_wrapper else 0)
    r = emu().emu_inflate(_vp(buf), _vp(offs), _vp(lens), n, _vp(out), _vp(ooff), _vp(caps), _vp(olen), _vp(st), f_)
    assert r == 0, f"emulator reported {r}"
    res = [out[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes() for i in range(n)], [int(x) for x in st]
    if pre:
        # the same streams through the two-kernel path (lane-per-stream pre-decoder + executor): same bytes, same status
        out2 = np.zeros(t + 64, dtype=np.uint8)
        olen2 = np.zeros(n, dtype=np.uint32)
        st2 = np.zeros(n, dtype=np.int32)
        cov = np.zeros(n, dtype=np.uint32)
        r = emu().emu_inflate_pre(_vp(buf), _vp(offs), _vp(lens), n, _vp(out2), _vp(ooff), _vp(caps), _vp(olen2), _vp(st2), f_, int(stage_bytes), _vp(cov))
        assert r == 0, f"emulator reported {r} (two-kernel inflate)"
        res2 = [out2[int(ooff[i]):int(ooff[i]) + int(olen2[i])].tobytes() for i in range(n)], [int(x) for x in st2]
        assert res2 == res, "two-kernel inflate differs from k_inflate"
        emu_inflate.last_covered = [int(x) for x in cov]
    return res


def dict_cases():
    """(name, dictionary, plain) triples for the decoder's raw-content dictionary support: the plain text shares
    material with the dictionary, so libzstd's frames hold matches that start inside it.  Seeded; the frames libzstd
    1.5.7 made from them are committed in tests/golden/zstd_dict_golden.json (make_golden_dict.py)."""
    from kompressor_amd import corpus
    out = []
    for k, (cls, dsz, psz) in enumerate([("T", 4096, 3000), ("T", 16384, 12000), ("X", 8192, 20000), ("S", 32768, 8000),
                                        ("B", 2048, 5000), ("T", 65536, 40000), ("D", 1000, 700), ("T", 300, 64)]):
        base = corpus.make(31000 + k, 1, dsz + psz, mix=ord(cls)).tobytes()
        d = base[:dsz]
        # plain = fresh material interleaved with pieces lifted from the dictionary (also from its very start and end)
        fresh = corpus.make(32000 + k, 1, psz, mix=ord(cls)).tobytes()
        third = max(1, psz // 3)
        plain = (d[-min(dsz, 200):] + fresh[:third] + d[:min(dsz, 300)] + fresh[third:2 * third] + d[dsz // 2:dsz // 2 + min(dsz // 2, 500)] + fresh[2 * third:])[:psz]
        out.append((f"{cls}_{dsz}_{psz}", d, plain))
    return out


def dict_compress_cases():
    """Seeded (dictionary, plain) pairs for the compress side with a dictionary: dictionary sizes 8 B .. 128 KiB, inputs
    1 B .. 128 KiB on both sides of the 16 KiB attach / copy cut-off; plain text unrelated to the dictionary, equal to
    it, runs, and mixtures of dictionary pieces and fresh material.  tests/golden/make_golden_dict.py stores what
    libzstd 1.5.7 makes of them."""
    import random
    from kompressor_amd import corpus
    rng = random.Random(60317)
    out = []
    for t in range(160):
        cls = rng.choice("TXSBDIZR")
        dsz = rng.choice([8, 12, 33, 255, 777, 3000, 8192, 16384, 16385, 40000, 100000, 131072])
        psz = rng.randrange(1, 131073) if rng.random() < 0.5 else rng.choice([1, 6, 7, 8, 9, 63, 64, 65, 1024, 16383, 16384, 16385, 16386, 65535, 131071, 131072])
        r = rng.random()
        d = corpus.make(rng.randrange(1 << 30), 1, dsz, mix=ord(cls)).tobytes()
        if r < 0.3:
            plain = corpus.make(rng.randrange(1 << 30), 1, psz, mix=ord(rng.choice("TXSBDIZR"))).tobytes()
        elif r < 0.4:
            plain = (d * (psz // dsz + 1))[:psz]
        elif r < 0.5:
            plain = bytes(psz) if rng.random() < 0.5 else (d[-3:] * psz)[:psz]
        else:
            fresh = corpus.make(rng.randrange(1 << 30), 1, psz, mix=ord(cls)).tobytes()
            parts, have = [], 0
            while have < psz:
                if rng.random() < 0.5:
                    a0 = rng.randrange(dsz)
                    seg = d[a0:a0 + rng.choice([4, 9, 40, 300, 5000])]
                else:
                    a0 = rng.randrange(psz)
                    seg = fresh[a0:a0 + rng.choice([1, 3, 20, 200, 3000])]
                parts.append(seg)
                have += len(seg)
            plain = b"".join(parts)[:psz]
        out.append((d, plain))
    return out


def emu_decompress(frames, caps, nblocks=2, dictionary=None):
    n = len(frames)
    lens = np.array([len(f) for f in frames], dtype=np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    pos = 16
    for i, f in enumerate(frames):
        offs[i] = pos
        pos += (len(f) + 31) & ~15
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for i, f in enumerate(frames):
        buf[int(offs[i]):int(offs[i]) + len(f)] = np.frombuffer(f, dtype=np.uint8)
    caps = np.array(caps, dtype=np.uint32)
    ooff = np.zeros(n, dtype=np.uint64)
    t = 0
    for i in range(n):
        ooff[i] = t
        t += int(caps[i]) + 16
    out = np.zeros(t + 64, dtype=np.uint8)
    olen = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.uint32)
    if dictionary is not None:
        dbuf = np.frombuffer(dictionary, dtype=np.uint8).copy()
        r = emu().emu_zstd_decompress_dict(_vp(buf), _vp(offs), _vp(lens), n, nblocks, _vp(out), _vp(ooff), _vp(caps), _vp(olen), _vp(st),
                                           128 * 1024 + 64, _vp(dbuf), len(dictionary))
    else:
        r = emu().emu_zstd_decompress(_vp(buf), _vp(offs), _vp(lens), n, nblocks, _vp(out), _vp(ooff), _vp(caps), _vp(olen), _vp(st),
                                      128 * 1024 + 64)
    assert r == 0, f"emulator reported {r}"
    return [out[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes() for i in range(n)], [int(x) for x in st]


def formatted_dict_built():
    """Seeded dictionaries in zstd's own format (magic EC30A437), put together by the oracle's builder so that the paths ZDICT's own
    dictionaries never take are covered: Huffman tables that lack byte values (HUF_repeat_check), sequence tables with missing codes
    (FSE_repeat_check), IDs of 0, 1, 2 and 4 bytes, repeat offsets anywhere in the content.  -> [(name, dictionary, class)]"""
    import random
    from kompressor_amd import corpus
    rng = random.Random(40417)
    out = []
    for case in range(8):
        cls = "TXSD"[case % 4]
        content = corpus.make(7000 + case, 1, rng.randrange(600, 40000), mix=ord(cls)).tobytes()
        lit = corpus.make(7100 + case, 1, 3000, mix=ord("TXSDB"[(case + 1) % 5])).tobytes()
        if case % 3 == 0:
            lit = bytes(b for b in lit if 97 <= b <= 122) or b"abcabcabd"        # letters only: most byte values get no code
        ll = [rng.randrange(50) for _ in range(36)]; of = [rng.randrange(50) for _ in range(32)]; ml = [rng.randrange(50) for _ in range(53)]
        if case % 2 == 0:       # complete tables (offsets: codes up to 19)
            ll = [x + 1 for x in ll]; ml = [x + 1 for x in ml]; of = [x + 1 for x in of[:20]] + [0] * 12
        if case % 4 == 1:
            ll[5] = 0; ml[7] = 0
        of[0] = max(of[0], 1); ll[0] = max(ll[0], 1); ml[0] = max(ml[0], 1)
        rep = [rng.randrange(1, len(content) + 1) for _ in range(3)]
        did = [0, 5, 300, 70000, 2 ** 31 + 5][case % 5]
        out.append((f"built{case}_{cls}_{len(content)}_id{did}", oracle().build_dictionary(did, lit, ll, of, ml, rep, content), cls))
    return out


def formatted_dict_inputs(cls, salt=0):
    """Seeded inputs for one dictionary: sizes on both sides of everything the dictionary path branches on (6 / 8 literals, 64, 256, 1024,
    the 16 KiB attach / copy cut-off, 1 000 sequences), of the dictionary's class."""
    from kompressor_amd import corpus
    sizes = (1, 5, 6, 7, 8, 30, 64, 200, 256, 700, 1023, 1024, 1025, 3000, 9000, 16384, 16385, 30000, 65536, 131072)
    return [corpus.make(99000 + n + salt, 1, n, mix=ord(cls)).tobytes() for n in sizes]


def formatted_dict_golden():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zstd_dict_formatted_golden.json")))


def formatted_dict_cases():
    """[(name, dictionary, inputs, golden row)] of tests/golden/zstd_dict_formatted_golden.json: the trained dictionaries from the fixture,
    the built ones rebuilt here and checked against the fixture's hash."""
    import base64, hashlib
    built = {name: d for name, d, _ in formatted_dict_built()}
    out = []
    for row in formatted_dict_golden()["rows"]:
        d = base64.b64decode(row["dict_b64"]) if "dict_b64" in row else built[row["name"]]
        assert hashlib.sha256(d).hexdigest() == row["dict_sha256"], row["name"]
        out.append((row["name"], d, formatted_dict_inputs(row["class"], salt=row["salt"]), row))
    return out


def lazy_level_inputs():
    """Seeded inputs for zstd levels 4 .. 10 (strategies greedy / lazy / lazy2): every corpus class at sizes on both sides of the 16 KiB
    change of match finder, up to one block; inputs with long runs (the row finder leaves the middle of a gap above 384 out of its
    tables; buckets of thousands) and random bytes (lazy skipping)."""
    import random
    from kompressor_amd import corpus
    rng = random.Random(51020)
    out = []
    for cls in "TXSBDIZR":
        for n in (1, 7, 8, 9, 100, 1000, 5000, 16384, 16385, 30000, 65536, 131072):
            out.append(corpus.make(52000 + n, 1, n, mix=ord(cls)).tobytes())
    for t in range(16):
        n = rng.choice([rng.randrange(2000, 16385), rng.randrange(16385, 131073), 131072])
        buf = bytearray()
        while len(buf) < n:
            buf += bytes([rng.choice(b"AB\x00")]) * rng.choice([5, 40, 300, 385, 386, 700, 5000, 20000, 45000])
            if rng.random() < 0.3:
                buf += corpus.make(rng.randrange(1 << 20), 1, rng.randrange(1, 400), mix=ord("T")).tobytes()
        out.append(bytes(buf[:n]))
    return out


def lazy_levels_golden():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zstd_lazy_levels_golden.json")))


def deflate_params_cases():
    """Seeded cases for deflateInit2's windowBits / memLevel (ZlibCompressor(format, compressionLevel, windowBits, memLevel)):
    [(level, window_bits, mem_level, fmt, seed, size, class)] -- every (windowBits, memLevel) pair at a deflate_fast and a deflate_slow
    level on 64 KiB slices, random settings over a ladder of sizes (empty, below MIN_MATCH, around MIN_LOOKAHEAD, around the smallest
    windows, the 64 KiB edge), and slices above 64 KiB (windows that slide hundreds of times, memLevel 9's 16-bit hash there)."""
    import random
    rng = random.Random(81531)
    out = []
    for wb in range(9, 16):
        for ml in range(1, 10):
            out.append((rng.choice((1, 2, 3)), wb, ml, 0, rng.randrange(1 << 20), 65536, rng.choice("TXSB")))
            out.append((rng.choice((4, 5, 6, 7, 8, 9)), wb, ml, 0, rng.randrange(1 << 20), rng.choice((65536, 40000, 9000)), rng.choice("TXSBDIZR")))
    ladder = (0, 1, 2, 3, 4, 100, 250, 251, 262, 263, 511, 512, 513, 762, 763, 1024, 4096, 16383, 16385, 32768, 65535, 65536)
    for _ in range(320):
        out.append((rng.randrange(1, 10), rng.randrange(9, 16), rng.randrange(1, 10), rng.randrange(3), rng.randrange(1 << 20), rng.choice(ladder), rng.choice("TXSBDIZR")))
    for size in (65537, 70000, 150000, 300000, 1 << 20):
        for _ in range(6):
            out.append((rng.randrange(1, 10), rng.choice((9, 10, 12, 14, 15)), rng.choice((1, 5, 8, 9, 9)), rng.randrange(3), rng.randrange(1 << 20), size, rng.choice("TXSBZ")))
    return out


def deflate_params_input(case):
    from kompressor_amd import corpus
    level, wb, ml, fmt, seed, size, cls = case
    return corpus.make(seed, 1, size, mix=ord(cls)).tobytes() if size else b""


def deflate_params_golden():
    with open(os.path.join(ROOT, "tests", "golden", "deflate_params_golden.json")) as fh:
        return json.load(fh)


def deflate_nil_corner_input(window_bits, seed=0, mem_level=8, k=1, extra=138):
    """An input that puts zlib's NIL where a candidate would be: at the end of the input fill_window runs at every step, so the window can
    slide when strstart is exactly w_size + MAX_DIST above the base -- after which the string MAX_DIST back sits at the new base, position 0
    of the window, which deflate takes for "no entry".  Here the eight bytes at that string are repeated at strstart and no other string
    between them shares their hash: zlib emits literals, a parser that forgets the base a match of distance MAX_DIST.  Compressible
    (a 16-letter alphabet), so that the blocks are not stored.  (Found by the differential fuzz, seed 61, with windowBits 9.)"""
    W = 1 << window_bits
    MD = W - 262
    hb = mem_level + 7
    sh = (hb + 2) // 3
    mask = (1 << hb) - 1
    basep = W * k
    n = basep + 2 * W - 262 + extra
    p = basep + 2 * W - 262
    c = p - MD
    for s in range(seed, seed + 1000):
        rng = np.random.default_rng(s)
        d = rng.integers(0, 16, n, dtype=np.uint8)
        d[c:c + 8] = rng.integers(200, 256, 8, dtype=np.uint8)
        d[p - 3:p] = rng.integers(100, 120, 3, dtype=np.uint8)          # no match runs into p
        d[p:p + 8] = d[c:c + 8]
        a = d.astype(np.int64)
        h = ((a[:-2] << (2 * sh)) ^ (a[1:-1] << sh) ^ a[2:]) & mask
        if not (h[c + 1:p] == h[p]).any():
            return d.tobytes()
    raise AssertionError("no seed")


def lazy_big_inputs():
    """Seeded inputs for zstd levels 4 .. 10 above 128 KiB (frames of several blocks): sizes on both sides of libzstd's parameter classes
    (256 KiB) and of the block size, up to 2 MiB; inputs whose statistics change inside a later block (the strategies' pre-splitter cuts
    there), constant and periodic ones (RLE blocks, repeat offsets across blocks), incompressible ones (raw blocks: savings stay below 3)."""
    import random
    from kompressor_amd import corpus
    rng = random.Random(61001)
    out = []
    for size in (131073, 131080, 140000, 200000, 262143, 262144, 262145, 300000, 524288, 700000, (1 << 20) + 5, 2 << 20):
        out.append(corpus.make(62000 + size, 1, size, mix=ord("TXSBDIZR"[len(out) % 8])).tobytes())
    for t in range(8):
        cut = 131072 + rng.randrange(8192, 120000)
        a = corpus.make(63000 + t, 1, cut, mix=ord("TXSB"[t % 4])).tobytes()
        out.append(a + corpus.make(63100 + t, 1, rng.randrange(60000, 400000), mix=ord("BZTR"[t % 4])).tobytes())
    out.append(bytes(300000))
    out.append((corpus.make(5, 1, 1000).tobytes() * 400)[:380000])
    out.append(corpus.make(64000, 1, 280000, mix=ord("R")).tobytes())
    return out


def lazy_big_golden():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zstd_lazy_big_golden.json")))


def rle_tail_cases():
    """Inputs whose last block is a short run of one byte (or nearly one): ZSTD_compressBlock_internal turns a block into an RLE block when
    the entropy stage's result is below 25 bytes -- 0 when the block would go out raw -- and the block is one repeated byte, whatever the
    parser found in it (the "fast" parser finds nothing in a block of 10 bytes).  Found by the differential fuzz late in round 4 (levels 1, 2
    and the negative ones wrote such a tail raw).  -> [(name, bytes)]"""
    import random
    from kompressor_amd import corpus
    rng = random.Random(72001)
    out = []
    for t, tail in enumerate((6, 7, 8, 9, 10, 11, 15, 24, 25, 26, 40, 63, 64, 65, 200, 5000)):
        base = (131072, 262144, 131072 + 40000)[t % 3]
        body = corpus.make(72100 + t, 1, base, mix=ord("TXSBZ"[t % 5])).tobytes()
        b = (0, 0x41, body[-1])[t % 3]
        out.append((f"run_{base}_{tail}_{b}", body + bytes([b]) * tail))
        if t % 4 == 1:
            out.append((f"almost_{base}_{tail}_{b}", body + bytes([b]) * (tail - 1) + bytes([b ^ 1])))
    return out


def rle_tail_golden():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zstd_rle_tail_golden.json")))
