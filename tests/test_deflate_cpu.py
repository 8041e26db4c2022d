"""Raw DEFLATE level 6 (BASELINE configs[4]) without a GPU: the oracle
(oracle/deflate_l6_ref.c) against the committed golden vectors made by this
machine's zlib and against Python's zlib live, and the kernel bodies
(kompressor_amd/csrc/deflate_*.h) on the CPU wave emulator."""
import base64
import zlib

import numpy as np

import helpers
from kompressor_amd import corpus


def raw6(d):
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0)
    return c.compress(d) + c.flush()


def test_oracle_matches_golden_and_reference_kat():
    G = helpers.deflate_golden()
    o = helpers.deflate_oracle()
    S = 65536
    rows = G["config4"][:256]
    buf = corpus.make(0, len(rows), S)
    for i, cls, flen, sha in rows:
        f = o.compress(buf[i * S:(i + 1) * S].tobytes())
        assert len(f) == flen and helpers.sha256(f) == sha, f"slice {i} class {cls}"
    for r in G["ladder"]:
        S, k = r["size"], r["index"] - 1000
        d = corpus.make(1000, 8, S)[k * S:(k + 1) * S].tobytes() if S else b""
        f = o.compress(d)
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r
    sp = helpers.special_inputs()
    for r in G["special"]:
        f = o.compress(sp[r["name"]])
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r["name"]
    # reference ZlibTest.kt:66-84 (zlib format, default level): the raw body is level-6 raw deflate
    kat = G["reference_kat"]
    assert o.compress(kat["plain"].encode()) == base64.b64decode(kat["raw_body_b64"])
    assert base64.b64decode(kat["zlib_b64"])[2:-4] == base64.b64decode(kat["raw_body_b64"])


def test_oracle_against_live_zlib_on_larger_inputs():
    o = helpers.deflate_oracle()
    rng = np.random.default_rng(3)
    for n in [70000, 98304, 131072, 200000]:           # several window slides
        d = corpus.make(4000, 1, n).tobytes()
        assert o.compress(d) == raw6(d), n
        d = rng.integers(0, 3, n, dtype=np.uint8).tobytes()
        assert o.compress(d) == raw6(d), n


def test_emulated_deflate_kernels_match_golden():
    G = helpers.deflate_golden()
    S = 65536
    rows = G["config4"][:8]
    buf = corpus.make(0, len(rows), S)
    outs = helpers.emu_deflate([buf[i * S:(i + 1) * S].tobytes() for i in range(len(rows))])
    for (i, cls, flen, sha), f in zip(rows, outs):
        assert len(f) == flen and helpers.sha256(f) == sha, f"slice {i} class {cls}"
    rows = [r for r in G["ladder"] if r["index"] in (1000, 1005)]
    datas = []
    for r in rows:
        S2, k = r["size"], r["index"] - 1000
        datas.append(corpus.make(1000, 8, S2)[k * S2:(k + 1) * S2].tobytes() if S2 else b"")
    outs = helpers.emu_deflate(datas)                   # ragged batch incl. empty input and the window-slide sizes
    for r, f in zip(rows, outs):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r
    sp = helpers.special_inputs()
    rows = G["special"]
    outs = helpers.emu_deflate([sp[r["name"]] for r in rows])
    for r, f in zip(rows, outs):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r["name"]
        assert zlib.decompress(f, -15) == sp[r["name"]]


def test_deflate_levels_4_to_9_oracle_and_emulator_against_zlib():
    """zlib's other lazy-matching levels (deflate.c configuration_table: 4: 4 4 16 16 ... 9: 32 258 258 4096): the oracle and the
    kernel bodies against this Python's zlib, raw streams and the level-dependent wrapper bytes (78 5E / 78 DA, gzip XFL)."""
    import zlib
    o = helpers.deflate_oracle()
    for lvl in (4, 5, 7, 8, 9):
        datas = [corpus.make(6100 + n + lvl, 1, n, mix=ord(c)).tobytes() for n, c in ((1, "T"), (300, "X"), (5000, "B"), (40000, "T"), (65536, "S"), (100000, "D"))]
        outs = helpers.emu_deflate(datas, level=lvl)
        for d, f in zip(datas, outs):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, 0)
            ref = co.compress(d) + co.flush()
            assert f == ref and o.compress(d, lvl) == ref, (lvl, len(d))
        assert helpers.emu_deflate(datas[:2], zlib_wrapper=True, level=lvl) == [zlib.compress(d, lvl) for d in datas[:2]]
        co = zlib.compressobj(lvl, zlib.DEFLATED, 31, 8, 0)
        assert helpers.emu_deflate(datas[1:2], fmt=2, level=lvl)[0] == co.compress(datas[1]) + co.flush()


def test_deflate_levels_1_to_3_oracle_and_emulator_against_zlib():
    """zlib's deflate_fast levels (configuration_table 1: 4 4 8 4, 2: 4 5 16 8, 3: 4 6 32 32; the strings inside a match longer
    than max_insert_length stay out of the hash chains): the oracle and k_deflate_fast's body against this Python's zlib --
    every content class, the empty and tiny inputs, sizes around the window slide, and the wrapper bytes (78 01 / 78 5E, XFL 4)."""
    import zlib
    o = helpers.deflate_oracle()
    sizes = ((0, "T"), (1, "T"), (2, "X"), (3, "X"), (4, "B"), (300, "X"), (5000, "B"), (40000, "T"), (65536, "S"), (65536, "I"),
             (65536, "R"), (65536, "L"), (65274, "T"), (65275, "D"), (100000, "D"), (140000, "T"))
    for lvl in (1, 2, 3):
        datas = [corpus.make(7100 + n + lvl, 1, n, mix=ord(c)).tobytes() for n, c in sizes]
        outs = helpers.emu_deflate(datas, level=lvl)
        for d, f in zip(datas, outs):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, 0)
            ref = co.compress(d) + co.flush()
            assert o.compress(d, lvl) == ref, (lvl, len(d))
            assert f == ref, (lvl, len(d))
        assert helpers.emu_deflate(datas[5:7], zlib_wrapper=True, level=lvl) == [zlib.compress(d, lvl) for d in datas[5:7]]
        co = zlib.compressobj(lvl, zlib.DEFLATED, 31, 8, 0)
        assert helpers.emu_deflate(datas[5:6], fmt=2, level=lvl)[0] == co.compress(datas[5]) + co.flush()


def test_emulated_inflate_predecoder_covers_huffman_streams():
    """k_inflate_predecode (a lane per stream, staging in the zstd decoder's format) + k_inflate_exec: every helpers.emu_inflate
    call runs them beside k_inflate and compares bytes and status; here also WHICH streams the pre-decoder takes: dynamic and
    fixed Huffman blocks of any level and strategy in all three wrappers, not stored blocks, not streams whose output is larger
    than the staging, not damaged ones -- those fall to inflate_stream inside k_inflate_exec."""
    import zlib
    datas = [corpus.make(8800 + k, 1, s, mix=ord(c)).tobytes() for k, (s, c) in enumerate(
        [(65536, "T"), (65536, "X"), (65536, "B"), (65536, "S"), (65536, "D"), (65536, "I"), (65536, "Z"), (40000, "T"), (300, "X"), (1, "T"), (0, "T")])]
    for lvl, strat, wb, fmt in [(6, 0, -15, 0), (1, 0, -15, 0), (9, 0, 15, 1), (6, zlib.Z_FIXED, 31, 2), (6, zlib.Z_HUFFMAN_ONLY, 15, 3), (4, zlib.Z_RLE, -15, 0)]:
        streams = []
        for x in datas:
            c = zlib.compressobj(lvl, zlib.DEFLATED, wb, 8, strat)
            streams.append(c.compress(x) + c.flush())
        outs, st = helpers.emu_inflate(streams, [max(len(x), 1) for x in datas], fmt=fmt)
        assert st == [0] * len(datas) and outs == datas, (lvl, strat)
        assert helpers.emu_inflate.last_covered == [1] * len(datas), (lvl, strat, helpers.emu_inflate.last_covered)
    rnd = corpus.make(8900, 1, 65536, mix=ord("R")).tobytes()
    big = corpus.make(8901, 1, 200000, mix=ord("T")).tobytes()
    good = zlib.compress(datas[0], 6)
    bad = good[:1000] + bytes([good[1000] ^ 0x40]) + good[1001:]
    streams = [zlib.compress(rnd, 6), zlib.compress(big, 6), bad, zlib.compress(datas[1], 0), good]
    outs, st = helpers.emu_inflate(streams, [65536, 200000, 65536, 65536, 65536], fmt=1)
    assert st[0] == 0 and st[1] == 0 and st[2] != 0 and st[3] == 0 and st[4] == 0
    assert outs[0] == rnd and outs[1] == big and outs[3] == datas[1] and outs[4] == datas[0]
    assert helpers.emu_inflate.last_covered[:2] == [0, 0] and helpers.emu_inflate.last_covered[3:] == [0, 1]     # stored; above the 64 KiB staging; level 0; and a good one


def test_emulated_zlib_wrapper_and_inflate():
    G = helpers.deflate_golden()
    kat = G["reference_kat"]
    # the reference's compress-side known-answer vector, byte for byte (ZlibTest.kt:66-84)
    out = helpers.emu_deflate([kat["plain"].encode()], zlib_wrapper=True)[0]
    assert out == base64.b64decode(kat["zlib_b64"])
    d = corpus.make(123, 1, 30000).tobytes()
    assert helpers.emu_deflate([d], zlib_wrapper=True)[0] == zlib.compress(d, 6)
    # inflate: every block type and level, raw and wrapped, plus error codes
    datas = [corpus.make(3000 + k, 1, s).tobytes() for k, s in enumerate([0, 1, 100, 5000, 40000, 65536])]
    for lvl, strat in [(6, 0), (1, 0), (9, 0), (6, zlib.Z_FIXED), (0, 0)]:
        streams = []
        for x in datas:
            c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
            streams.append(c.compress(x) + c.flush())
        outs, st = helpers.emu_inflate(streams, [max(len(x), 1) for x in datas])
        assert st == [0] * len(datas) and outs == datas, (lvl, strat)
    outs, st = helpers.emu_inflate([zlib.compress(x, 6) for x in datas], [max(len(x), 1) for x in datas], zlib_wrapper=True)
    assert st == [0] * len(datas) and outs == datas
    good = zlib.compress(datas[3], 6)
    bad = bytearray(good)
    bad[-1] ^= 1
    outs, st = helpers.emu_inflate([bytes(bad), good, good[:-9]], [5000, 100, 5000], zlib_wrapper=True)
    assert st[0] == -3 and st[1] == -5 and st[2] != 0


def _gzip6(x):
    c = zlib.compressobj(6, zlib.DEFLATED, 31, 8, 0)       # deflateInit2(6, Z_DEFLATED, 15 + 16, 8, 0): ZlibFormat.Gzip
    return c.compress(x) + c.flush()


def test_emulated_gzip_wrapper_and_autodetect():
    """ZlibFormat.Gzip / AutoDetectZlibGzip (ZlibFormat.kt:39-55): header, CRC-32 and ISIZE as zlib writes them."""
    import gzip
    datas = [corpus.make(4100 + k, 1, s).tobytes() for k, s in enumerate([0, 1, 63, 64, 100, 5000, 40001, 65536])]
    outs = helpers.emu_deflate(datas, fmt=2)
    for x, f in zip(datas, outs):
        assert f == _gzip6(x), len(x)
    # decode: our own members, members from the gzip module (other XFL / OS / MTIME), optional header fields
    dec, st = helpers.emu_inflate(outs, [max(len(x), 1) for x in datas], fmt=2)
    assert st == [0] * len(datas) and dec == datas
    foreign = [gzip.compress(x, 9, mtime=1234567) for x in datas]
    dec, st = helpers.emu_inflate(foreign, [max(len(x), 1) for x in datas], fmt=2)
    assert st == [0] * len(datas) and dec == datas
    body = _gzip6(datas[5])[10:]
    fancy = bytes([0x1F, 0x8B, 8, 4 | 8 | 16 | 2, 1, 2, 3, 4, 0, 3]) + bytes([5, 0]) + b"extra" + b"name.txt\0" + b"a comment\0" + b"\x12\x34" + body
    dec, st = helpers.emu_inflate([fancy], [5000], fmt=2)
    assert st == [0] and dec[0] == datas[5]
    # the reference's gzip decode vector (ZlibTest.kt:86-98), also through auto-detection
    kat = base64.b64decode("H4sIAIUNSGkAA8tIzcnJV0jOzy0oSi0uzszPUyjPL8pJAQDFwzyrFwAAAA==")
    zl = base64.b64decode(helpers.deflate_golden()["reference_kat"]["zlib_b64"])
    dec, st = helpers.emu_inflate([kat, kat, zl], [100, 100, 100], fmt=3)
    assert st == [0, 0, 0] and dec == [b"hello compression world"] * 3
    dec, st = helpers.emu_inflate([kat], [100], fmt=2)
    assert st == [0] and dec[0] == b"hello compression world"
    # errors: wrong CRC, wrong ISIZE, not gzip, truncated, capacity
    good = outs[5]
    bad_crc = bytearray(good); bad_crc[-5] ^= 0x40
    bad_len = bytearray(good); bad_len[-1] ^= 1
    not_gz = bytearray(good); not_gz[1] = 0x8C
    dec, st = helpers.emu_inflate([bytes(bad_crc), bytes(bad_len), bytes(not_gz), good[:-12], good], [5000, 5000, 5000, 5000, 100], fmt=2)
    assert st[0] == -3 and st[1] == -3 and st[2] == -3 and st[3] != 0 and st[4] == -5


def test_long_slices_oracle_and_emulated_kernels_match_zlib():
    """Slices above 64 KiB: zlib's window slides every 32 KiB (candidates end at MAX_DIST, a block whose start has left the
    buffer cannot be stored).  The oracle and the four kernel bodies (links and match records as distances, 32-bit
    positions, the parse's model of fill_window) against the committed zlib streams of 12 inputs up to 1 MiB + 3 -- the
    reference's own round-trip size (ZlibTest.kt:16,28-33) -- and against Python's zlib live."""
    G = helpers.deflate_golden()
    o = helpers.deflate_oracle()
    inputs = helpers.deflate_long_inputs()
    rows = {r["name"]: r for r in G["long"]}
    for name, d in inputs:
        r = rows[name]
        assert len(d) == r["size"] and helpers.sha256(d) == r["input_sha256"], name
        f = o.compress(d)
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], ("oracle", name)
        if zlib.ZLIB_RUNTIME_VERSION == G["zlib"]:
            assert f == raw6(d), name
    # what the product runs above 64 KiB: the sort + wave-wide parse kernels segment by segment (deflate_lazy.h: a 64 KiB span per launch,
    # the parse's state carried from one segment to the next)
    emu_in = [(name, d) for name, d in inputs if name not in ("long_300001", "long_777777", "long_1048576", "zeros_1m")]      # (the CPU suite's time; the GPU suite runs all twelve)
    outs = helpers.emu_deflate([d for _, d in emu_in])
    for (name, d), f in zip(emu_in, outs):
        assert len(f) == rows[name]["len"] and helpers.sha256(f) == rows[name]["sha256"], ("kernels", name)
    # lengths on both sides of the segments' ends (a span ends every 32 KiB; the last one may hold a few bytes, or none that enter a chain)
    edge = corpus.make(515, 1, 140000, mix=ord("X")).tobytes()
    cuts = [edge[:k] for k in (65537, 65539, 98303, 98304, 98305, 98307, 131071, 131072, 131074)]
    for d, f in zip(cuts, helpers.emu_deflate(cuts, level=5, window_bits=14, mem_level=9)):
        c = zlib.compressobj(5, zlib.DEFLATED, -14, 9, 0)
        assert f == c.compress(d) + c.flush(), len(d)
    # the older kernels (chain links, all-positions search, the parse over its records as a wave and as a lane per slice): ablation
    # build only since the segments, kept honest here on two of the inputs
    small_long = sorted(inputs, key=lambda x: len(x[1]))[:2]
    for old in (1, 2):
        for (name, d), f in zip(small_long, helpers.emu_deflate([d for _, d in small_long], old_kernels=old)):
            assert len(f) == rows[name]["len"] and helpers.sha256(f) == rows[name]["sha256"], ("older kernels", old, name)
    # the wrappers' checksums run over the whole slice
    d = inputs[5][1]
    assert helpers.emu_deflate([d], fmt=1)[0] == zlib.compress(d, 6)
    c = zlib.compressobj(6, zlib.DEFLATED, 31, 8, 0)
    assert helpers.emu_deflate([d], fmt=2)[0] == c.compress(d) + c.flush()


def _wrap(raw, d, fmt, level, wb):
    """the zlib / gzip wrapper zlib puts around a raw stream (deflate.c deflate(): INIT_STATE header, trailer)"""
    import struct
    if fmt == 0:
        return raw
    if fmt == 1:
        lf = 0 if level < 2 else 1 if level < 6 else 2 if level == 6 else 3
        hdr = ((8 + ((wb - 8) << 4)) << 8) | (lf << 6)
        hdr += 31 - hdr % 31
        return struct.pack(">H", hdr) + raw + struct.pack(">I", zlib.adler32(d))
    xfl = 2 if level == 9 else 4 if level == 1 else 0
    return bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, xfl, 3]) + raw + struct.pack("<II", zlib.crc32(d), len(d) & 0xFFFFFFFF)


def test_window_bits_and_mem_level_oracle_and_emulator_match_golden():
    """deflateInit2's windowBits 9 .. 15 and memLevel 1 .. 9 (the reference's ZlibCompressor(format, compressionLevel, windowBits,
    memLevel), ZlibCompressor.jvm.kt:7-17): the oracle on all 476 committed cases (tests/golden/deflate_params_golden.json, zlib
    1.2.11), the emulated kernels -- both the sort + wave-wide parse and the older chain / search / parse kernels, memLevel 9's
    two-pass chains among them -- on the smaller ones and a few above 64 KiB."""
    G = helpers.deflate_params_golden()
    cases = helpers.deflate_params_cases()
    assert len(cases) == len(G["rows"])
    o = helpers.deflate_oracle()
    emu_rows = []
    for k, (case, (glen, gsha)) in enumerate(zip(cases, G["rows"])):
        level, wb, ml, fmt, seed, size, cls = case
        d = helpers.deflate_params_input(case)
        f = _wrap(o.compress(d, level, wb, ml), d, fmt, level, wb)
        assert len(f) == glen and helpers.sha256(f) == gsha, case
        if size <= 4096 or k % 29 == 0:
            emu_rows.append((k, case, d, f))
    assert len(emu_rows) >= 120 and any(c[5] > 65536 for _, c, _, _ in emu_rows)
    for j, (k, case, d, f) in enumerate(emu_rows):
        level, wb, ml, fmt, seed, size, cls = case
        got = helpers.emu_deflate([d], fmt=fmt, level=level, window_bits=wb, mem_level=ml, old_kernels=(j % 3 == 0))[0]
        assert got == f, case
    # memLevel 9 above 64 KiB at a lazy level: the hash is wider than the chain kernel's table (two passes)
    d = corpus.make(4242, 1, 90000, mix=ord("T")).tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0)
    assert helpers.emu_deflate([d], level=6, window_bits=15, mem_level=9)[0] == c.compress(d) + c.flush()
    # (the live zlib of this machine over random settings, so that the fixture is not the only witness)
    rng = np.random.default_rng(5)
    for t in range(60):
        level, wb, ml = int(rng.integers(1, 10)), int(rng.integers(9, 16)), int(rng.integers(1, 10))
        d = corpus.make(7000 + t, 1, int(rng.integers(0, 200000)), mix=ord("TXSBDIZR"[t % 8])).tobytes()
        c = zlib.compressobj(level, zlib.DEFLATED, -wb, ml, 0)
        assert o.compress(d, level, wb, ml) == c.compress(d) + c.flush(), (level, wb, ml, len(d))


def test_emulated_inflate_honours_the_declared_window():
    """inflateInit2(windowBits) below what a zlib header names: "invalid window size" (zlib inflate.c HEAD state), on both inflate
    paths; raw and gzip streams carry no such field.  (kmp_zlib_create_decompressor passes the declared window on.)"""
    d = corpus.make(31, 1, 5000, mix=ord("T")).tobytes()
    for made_with in (9, 12, 15):
        c = zlib.compressobj(6, zlib.DEFLATED, made_with, 8, 0)
        s = c.compress(d) + c.flush()
        for declared in (9, 11, 12, 15):
            try:
                ref = zlib.decompressobj(declared).decompress(s)
            except zlib.error:
                ref = None
            for pre in (True, False):
                outs, st = helpers.emu_inflate([s], [len(d)], fmt=1 | (declared << 8), pre=pre)
                assert (outs[0] if st[0] == 0 else None) == ref, (made_with, declared, pre, st)
                assert ref is not None or st[0] == -3


def test_the_window_base_is_nil_when_the_input_ends():
    """helpers.deflate_nil_corner_input: the string exactly MAX_DIST back lies at the base of zlib's window after the slide that
    fill_window makes at the end of the input -- no candidate for zlib; the oracle and both emulated parsers (the wave-wide one up to
    64 KiB, the older one above and on request) must say so too, at the default window (a 98 180-byte slice) as at small ones."""
    o = helpers.deflate_oracle()
    for wb, seeds in ((9, (0, 1, 2)), (12, (0, 1)), (15, (0,))):
        for seed in seeds:
            d = helpers.deflate_nil_corner_input(wb, seed)
            for lvl in ((6, 4, 9, 2) if wb < 15 else (6,)):
                c = zlib.compressobj(lvl, zlib.DEFLATED, -wb, 8, 0)
                ref = c.compress(d) + c.flush()
                assert o.compress(d, lvl, wb, 8) == ref
                for old in ((False, True) if len(d) <= 65536 else (True,)):
                    assert helpers.emu_deflate([d], level=lvl, window_bits=wb, mem_level=8, old_kernels=old)[0] == ref, (wb, seed, lvl, old)
