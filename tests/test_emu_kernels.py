"""The kernel bodies (kompressor_amd/csrc/zstd_match.h, zstd_entropy.h,
zstd_decode.h) executed lane for lane on the CPU wave emulator (tests/emu)
against the golden vectors.  This is host-side coverage of the device
algorithm; the -m gpu tests are the parity tests proper."""
import base64

import numpy as np

import pytest

import helpers
from kompressor_amd import corpus


@pytest.fixture(scope="module")
def G():
    return helpers.golden()


@pytest.mark.parametrize("team", [2, 4, 8, 16, 64])
def test_emulated_compress_matches_golden_64k(G, team):
    rows = G["config1"][:16] if team == 8 else G["config1"][16:32] if team == 4 else G["config1"][32:40]
    S = 65536
    first = rows[0][0]
    buf = corpus.make(first, len(rows), S)
    frames = helpers.emu_compress([buf[k * S:(k + 1) * S].tobytes() for k in range(len(rows))], G=team)
    for (i, cls, flen, sha), f in zip(rows, frames):
        assert len(f) == flen and helpers.sha256(f) == sha, f"slice {i} class {cls} team {team}"


@pytest.mark.parametrize("team,ring", [(4, 256), (2, 256), (8, 512), (4, 512)])
def test_emulated_split_phase_parser_matches_golden(G, monkeypatch, team, ring):
    """zstd_match2.h (the level-3 parse as a split-phase stage machine with an LDS source window): the same frames as libzstd
    1.5.7 on 64 KiB slices of every class, on the ragged ladder (empty, tiny, maximum), and on the hand-made edge inputs."""
    monkeypatch.setenv("KXEMU_MATCH_V2", "1")
    monkeypatch.setenv("KXEMU_RING", str(ring))
    rows = G["config1"][40:56] if team == 4 and ring == 256 else G["config1"][56:64]
    S = 65536
    buf = corpus.make(rows[0][0], len(rows), S)
    frames = helpers.emu_compress([buf[k * S:(k + 1) * S].tobytes() for k in range(len(rows))], G=team, nblocks=max(1, len(rows) * team // 128))
    for (i, cls, flen, sha), f in zip(rows, frames):
        assert len(f) == flen and helpers.sha256(f) == sha, f"slice {i} class {cls} team {team}"
    lad = [r for r in G["ladder"] if r["index"] in (1001, 1005)]
    datas = []
    for r in lad:
        S2, k = r["size"], r["index"] - 1000
        datas.append(corpus.make(1000, 8, S2)[k * S2:(k + 1) * S2].tobytes() if S2 else b"")
    frames = helpers.emu_compress(datas, G=team, nblocks=3)
    for r, f in zip(lad, frames):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r
    sp = helpers.special_inputs()
    frames = helpers.emu_compress([sp[r["name"]] for r in G["special"]], G=team)
    for r, f in zip(G["special"], frames):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r["name"]


@pytest.mark.parametrize("team", [4, 8])
def test_emulated_fused_kernel_matches_golden(G, monkeypatch, team):
    """k_zstd_l3_fused's body (zstd_entropy.h: a wave entropy-codes each slice the moment one of its teams has parsed it, the parse
    state set aside in private memory around the call): the frames of libzstd 1.5.7 on 64 KiB slices of every class (several
    slices per team, so that teams fetch new slices after such a call), the ragged ladder and the hand-made edge inputs."""
    monkeypatch.setenv("KXEMU_FUSE", "1")
    rows = G["config1"][64:80] if team == 4 else G["config1"][80:88]
    S = 65536
    buf = corpus.make(rows[0][0], len(rows), S)
    frames = helpers.emu_compress([buf[k * S:(k + 1) * S].tobytes() for k in range(len(rows))], G=team, nblocks=1 if team == 8 else 2)
    for (i, cls, flen, sha), f in zip(rows, frames):
        assert len(f) == flen and helpers.sha256(f) == sha, f"slice {i} class {cls} team {team}"
    lad = G["ladder"]
    datas = []
    for r in lad:
        S2, k = r["size"], r["index"] - 1000
        datas.append(corpus.make(1000, 8, S2)[k * S2:(k + 1) * S2].tobytes() if S2 else b"")
    frames = helpers.emu_compress(datas, G=team, nblocks=1)
    for r, f in zip(lad, frames):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r
    sp = helpers.special_inputs()
    frames = helpers.emu_compress([sp[r["name"]] for r in G["special"]], G=team, nblocks=1)
    for r, f in zip(G["special"], frames):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r["name"]


@pytest.mark.parametrize("team", [4, 8])
def test_emulated_level_4_matches_golden(monkeypatch, team):
    """zstd_match.h with level 4's double-fast row (hash 17, chain up to 17, minimum match 4; a team's tables 1 MiB): the
    frames of libzstd 1.5.7 at level 4 on 64 KiB slices of every class and on the ragged sizes above 16 KiB."""
    monkeypatch.setenv("KXEMU_LEVEL", "4")
    G = helpers.level4_golden()
    S = 65536
    lo = 0 if team == 4 else 16
    buf = corpus.make(lo, 16, S)
    frames = helpers.emu_compress([buf[k * S:(k + 1) * S].tobytes() for k in range(16)], G=team, nblocks=1)
    for (i, flen, sha), f in zip(G["config1"][lo:lo + 16], frames):
        assert len(f) == flen and helpers.sha256(f) == sha, (i, team)
    lad = [r for r in G["ladder"] if r[1] in ((0, 2, 5) if team == 4 else (1, 6))]
    frames = helpers.emu_compress([corpus.make(1000, 8, S2)[k * S2:(k + 1) * S2].tobytes() for S2, k, _l, _s in lad], G=team, nblocks=1)
    for (S2, k, flen, sha), f in zip(lad, frames):
        assert len(f) == flen and helpers.sha256(f) == sha, (S2, k, team)


def test_emulated_level_4_on_the_block_chain_path():
    """Level 4's double-fast rows for frames of several blocks (above 256 KiB: window 21, chain 18, hash 18; per-slice tables of 2 MiB)
    and for streams of any size: ZSTD_compress2's frames, the reference driver's, streamed ones -- one input beyond the 2 MiB window
    (sliding window, extDict blocks) -- against the oracle (pinned on libzstd 1.5.7: 99 frames while developing, four fuzz legs on the GPU
    box); slices of the size classes level 4 runs as "greedy" (up to 16 KiB, 128 - 256 KiB) come back refused."""
    o = helpers.oracle()
    datas = [corpus.make(80, 1, S).tobytes() for S in (20000, 262145, 400000)]
    far = corpus.make(81, 1, 2300000, mix=ord("S")).tobytes()                  # beyond the window + one chunk: the staging buffer wraps
    for mode in (0, 3, 1, 2):
        ins = (datas[:2] if mode in (0, 3) else datas[1:2]) + ([far] if mode == 3 else [])        # (the streamed forms on one input, the 400 000-byte one on the GPU only: the CPU suite's time)
        frames, _ = helpers.emu_compress_big(ins, G=8, nblocks=2, stream=mode, level=4)
        for d, f in zip(ins, frames):
            want = o.compress_buffered(d, 2 if mode == 0 else mode == 3, mode == 2, level=4)
            assert f == want, (len(d), mode)
    frames, _ = helpers.emu_compress_big([datas[0][:9000], datas[1], datas[2][:200000]], G=8, nblocks=2, stream=0, level=4)
    assert [len(f) for f in frames][0] == 0 and len(frames[2]) == 0 and frames[1] == o.compress_buffered(datas[1], 2, level=4)


def test_emulated_negative_levels_above_128_kib():
    """Negative levels on the block-chain path (the level-1 machinery with row 0 of libzstd's tables, a step of 1 - level, literals
    left raw): ZSTD_compress2's frames, the frames the reference's driver gets and streamed ones, against the oracle (pinned on
    libzstd 1.5.7 for these: 324 frames of three levels while developing, and the fuzz legs on the GPU box)."""
    o = helpers.oracle()
    datas = []
    for S in (131073, 262145, 400000):
        buf = corpus.make(60, 2, S)
        datas += [buf[k * S:(k + 1) * S].tobytes() for k in range(2 if S < 400000 else 1)]          # (the CPU suite's time)
    for lvl, mode in ((-1, 0), (-6, 3), (-2, 1)):
        frames, _ = helpers.emu_compress_big(datas, G=8, nblocks=2, stream=mode, level=lvl)
        for d, f in zip(datas, frames):
            assert f == o.compress_level_big(d, lvl, stream=mode), (lvl, len(d), mode)


@pytest.mark.parametrize("level,team", [(-1, 4), (-3, 8), (-20, 4), (-1000, 16)])
def test_emulated_negative_levels_match_golden(level, team):
    """zstd_match_fast.h with a step of 1 - level on row 0 of libzstd's tables, the entropy stage with literals left raw: the frames
    of libzstd 1.5.7 at negative levels on a part of the size ladder and on 64 KiB slices of every class."""
    G = helpers.neg_levels_golden()
    j = G["levels"].index(level)
    rows = [r for r in G["ladder"] if r[1] in (0, 3, 6) and r[0] in (0, 1, 7, 8, 9, 64, 300, 1024, 4096, 16384, 16385, 40960, 65537, 131072)]
    datas = [corpus.make(1000, 8, r[0])[r[1] * r[0]:(r[1] + 1) * r[0]].tobytes() if r[0] else b"" for r in rows]
    frames = helpers.emu_compress_level(datas, level, G=team, nblocks=1)
    for r, f in zip(rows, frames):
        assert len(f) == r[2 + 2 * j] and helpers.sha256(f)[:32] == r[3 + 2 * j], (r[0], r[1], level)
    S = 65536
    buf = corpus.make(0, 16, S)
    frames = helpers.emu_compress_level([buf[i * S:(i + 1) * S].tobytes() for i in range(16)], level, G=team, nblocks=1)
    for r, f in zip(G["config1"][:16], frames):
        assert len(f) == r[1 + 2 * j] and helpers.sha256(f)[:32] == r[2 + 2 * j], (r[0], level)


def test_emulated_split_phase_parser_at_the_end_of_a_slice(monkeypatch):
    """Matches that run into the last bytes of a slice (the window's 16-byte looks must not count bytes they do not hold): every
    distance of a repeat's start from the end, several periods, against the oracle."""
    monkeypatch.setenv("KXEMU_MATCH_V2", "1")
    o = helpers.oracle()
    rng = np.random.default_rng(11)
    ds = []
    for per in (1, 2, 3, 5, 8, 15, 16, 17, 33):
        pat = rng.integers(0, 256, per, dtype=np.uint8).tobytes()
        for tail in range(0, 40, 3):
            pre = rng.integers(0, 256, 64 + per, dtype=np.uint8).tobytes()
            body = (pat * 80)[:per + 9 + tail]
            ds += [pre + body, pre + body + b"\x07", pre + body[:-1] + bytes([body[-1] ^ 1])]
    frames = helpers.emu_compress(ds, G=4, nblocks=4)
    for d, f in zip(ds, frames):
        assert f == o.compress(d), len(d)


def test_emulated_compress_ladder_ragged_batch(G):
    # one batch with every ladder size at once: ragged lengths, empty input, 128 KiB maximum
    rows = [r for r in G["ladder"] if r["index"] in (1000, 1003)]
    datas = []
    for r in rows:
        S, k = r["size"], r["index"] - 1000
        datas.append(corpus.make(1000, 8, S)[k * S:(k + 1) * S].tobytes() if S else b"")
    frames = helpers.emu_compress(datas, G=8, nblocks=3)
    for r, f in zip(rows, frames):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r


def test_emulated_compress_specials(G):
    sp = helpers.special_inputs()
    rows = G["special"]
    frames = helpers.emu_compress([sp[r["name"]] for r in rows], G=16)
    for r, f in zip(rows, frames):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r["name"]


def test_emulated_decoder(G):
    kat = base64.b64decode(G["reference_kats"]["zstd_sampleHello_frame_b64"])        # ZstdTest.kt:84-91
    outs, st = helpers.emu_decompress([kat], [64])
    assert st == [0] and outs[0].decode() == G["reference_kats"]["zstd_sampleHello_plain"]
    # frames with a payload kept in the fixture
    frames, plains = [], []
    sp = helpers.special_inputs()
    for r in G["special"]:
        if "frame" in r:
            frames.append(base64.b64decode(r["frame"]))
            plains.append(sp[r["name"]])
    outs, st = helpers.emu_decompress(frames, [max(len(p), 1) for p in plains])
    assert st == [0] * len(frames)
    assert outs == plains
    # other levels and multi-block frames (treeless literals, repeat tables, window descriptor)
    d = G["decode_only"]
    outs, st = helpers.emu_decompress([base64.b64decode(r["frame"]) for r in d], [r["size"] for r in d])
    for r, o, s in zip(d, outs, st):
        assert s == 0 and helpers.sha256(o) == r["plain_sha256"], r["index"]


def test_emulated_decoder_without_the_predecoders(G, monkeypatch):
    """k_zstd_decode's own sequence and literal decoding (what runs for blocks the pre-decode kernels did not stage): the
    emulator's switch KXEMU_NO_PRE leaves everything to it."""
    monkeypatch.setenv("KXEMU_NO_PRE", "1")
    d = G["decode_only"]
    outs, st = helpers.emu_decompress([base64.b64decode(r["frame"]) for r in d], [r["size"] for r in d])
    for r, o, s in zip(d, outs, st):
        assert s == 0 and helpers.sha256(o) == r["plain_sha256"], r["index"]
    S = 40000
    buf = corpus.make(7100, 6, S)
    datas = [buf[k * S:(k + 1) * S].tobytes() for k in range(6)]
    frames = helpers.emu_compress(datas, G=8)
    outs, st = helpers.emu_decompress(frames, [S] * 6)
    assert st == [0] * 6 and outs == datas


def test_emulated_decoder_takes_frames_of_other_settings(monkeypatch):
    """A third of tests/golden/foreign_frames.* (levels 1 .. 22, checksum, large window, runs and periods) through the three
    decode kernels on the emulator, and a few of them through k_zstd_decode alone."""
    rows = helpers.foreign_frames()
    pick = [x for i, x in enumerate(rows) if i % 3 == 0 or x[0]["kind"] != "corpus"]
    outs, st = helpers.emu_decompress([f for _, f, _ in pick], [len(p) for _, _, p in pick])
    for (r, _, plain), o, s_ in zip(pick, outs, st):
        assert s_ == 0 and o == plain, (r["level"], r.get("cls"), r["size"])
    monkeypatch.setenv("KXEMU_NO_PRE", "1")
    few = pick[::6]
    outs, st = helpers.emu_decompress([f for _, f, _ in few], [len(p) for _, _, p in few])
    for (r, _, plain), o, s_ in zip(few, outs, st):
        assert s_ == 0 and o == plain, (r["level"], r.get("cls"), r["size"])


def test_emulated_sequence_count_sort():
    """The three small kernels that order the pre-decoders' lane slots (count, rank, perm): the slot -> entry map is a
    permutation, most sequences first, and the key is the first compressed block's sequence count / 64 (0 for raw frames,
    empty entries and garbage)."""
    import ctypes
    o = helpers.oracle()
    datas = [corpus.make(8800 + i, 1, n, mix=ord(c)).tobytes() for i, (n, c) in enumerate(
        [(65536, "T"), (65536, "R"), (3000, "X"), (40000, "B"), (65536, "Z"), (100, "T"), (20000, "S"), (65536, "D"), (50000, "I")] * 35)]
    frames = [o.compress(d) for d in datas[:9]] * 35 + [b"", b"garbage!" * 4]
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
    key = np.zeros(n, dtype=np.uint32)
    perm = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)                      # noqa: E731
    assert helpers.emu().emu_seq_sort(vp(buf), vp(offs), vp(lens), n, vp(key), vp(perm)) == 0
    assert sorted(perm.tolist()) == list(range(n))
    ks = key[perm]
    assert all(int(ks[i]) >= int(ks[i + 1]) for i in range(n - 1))
    assert key[1] == 0 and key[n - 1] == 0 and key[n - 2] == 0 and key[0] > 40         # random slice: raw block; empty; garbage; text


def test_emulated_decoder_rejects_bad_frames(G):
    good = base64.b64decode(next(r["frame"] for r in G["special"] if r["name"] == "ramp_64k"))
    bad_magic = b"\x00" + good[1:]
    truncated = good[:-3]
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x40
    small_cap = good
    outs, st = helpers.emu_decompress([bad_magic, truncated, bytes(flipped), small_cap], [65536, 65536, 65536, 100])
    assert st[0] == 10                      # Unknown frame descriptor
    assert st[1] in (20, 72)                # corruption / src size
    assert st[2] != 0 or outs[2] != bytes(range(256)) * 256
    assert st[3] == 70                      # Destination buffer is too small


def test_emulated_roundtrip_property():
    S = 40000
    buf = corpus.make(7000, 8, S)
    datas = [buf[k * S:(k + 1) * S].tobytes() for k in range(8)]
    frames = helpers.emu_compress(datas, G=8)
    outs, st = helpers.emu_decompress(frames, [S] * 8)
    assert st == [0] * 8 and outs == datas


def test_emulated_multiblock_frames():
    """Frames of several blocks (slices above 128 KiB) against the libzstd 1.5.7 goldens: block-mode match kernel
    + frame kernel; a few of the smaller vectors keep the CPU suite short (the GPU suite runs all of them)."""
    rows = helpers.multiblock_golden()["rows"]
    inputs = helpers.multiblock_inputs()
    pick = [i for i, (name, d) in enumerate(inputs) if name in ("zeros_300k", "run_128k_plus_5", "text_256k_plus_3",
                                                                  "mixed_1_300000", "mixed_17_262145", "mixed_7_140000")]
    assert len(pick) == 6
    frames, _ = helpers.emu_compress_big([inputs[i][1] for i in pick], G=16)
    for i, f in zip(pick, frames):
        assert len(f) == rows[i]["len"] and helpers.sha256(f) == rows[i]["sha256"], inputs[i][0]
    # the same steps as separate launches per round of blocks (the experiment switch of the library)
    frames, rounds = helpers.emu_compress_big([inputs[i][1] for i in pick[:3]], G=8, by_rounds=True)
    for i, f in zip(pick, frames):
        assert helpers.sha256(f) == rows[i]["sha256"], inputs[i][0]
    assert rounds >= 3
    # another team width, and small slices through the block path give the single-block frames
    o = helpers.oracle()
    small = [corpus.make(88000 + k, 1, s).tobytes() for k, s in enumerate([0, 5, 7, 8, 300, 20000])]
    frames, _ = helpers.emu_compress_big(small + [inputs[pick[-1]][1]], G=4)
    for d, f in zip(small, frames):
        assert f == o.compress(d), len(d)
    assert helpers.sha256(frames[-1]) == rows[pick[-1]]["sha256"]


def test_emulated_decoder_with_raw_dictionary():
    """ZstdDecompressor(dictionary) (Wrapper.cpp:58-73, ZstdTest.kt:49-65): frames libzstd 1.5.7 made with a raw-content
    dictionary decode with it, and fail (or decode to something else) without it."""
    import base64
    import json
    import os
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "zstd_dict_golden.json")))["rows"]
    for (name, d, plain), row in zip(helpers.dict_cases(), rows):
        assert name == row["name"] and helpers.sha256(d) == row["dict_sha256"] and helpers.sha256(plain) == row["plain_sha256"]
        frame = base64.b64decode(row["frame"])
        outs, st = helpers.emu_decompress([frame], [len(plain)], dictionary=d)
        assert st == [0] and outs[0] == plain, name
        outs, st = helpers.emu_decompress([frame], [len(plain)])
        assert st[0] != 0 or outs[0] != plain, name            # the reference's test expects a failure without the dictionary


def test_emulated_compress_with_raw_dictionary():
    """ZstdCompressor(3, dictionary) (Wrapper.cpp:41-56): the dictionary match kernel (attached CDict up to 16 KiB of
    input, copied-table variant above) + entropy kernel against the frames of libzstd 1.5.7."""
    import json
    import os
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "zstd_dict_golden.json")))["compress"]
    cases = helpers.dict_compress_cases()
    picked = 0
    modes = set()
    for (d, plain), (dsz, psz, tag, flen, sha) in zip(cases, G):
        if dsz > 130560 or psz > 40000 or picked >= 28:
            continue
        f = helpers.emu_compress_dict([plain], d, G=(4, 2, 8, 16)[picked % 4])[0]
        assert len(f) == flen and helpers.sha256(f) == sha, (dsz, psz)
        modes.add(psz <= 16384)
        picked += 1
    assert picked == 28 and modes == {True, False}


def test_emulated_levels_1_and_2():
    """Strategy "fast" (levels 1 and 2): fast match kernel + entropy kernel against the frames of libzstd 1.5.7 (every
    fifth row of the size ladder, both levels, rotating team widths)."""
    G = helpers.levels_golden()
    n = 0
    for S, k, l1, s1, l2, s2 in G["ladder"][::5]:
        d = corpus.make(1000, 8, S)[k * S:(k + 1) * S].tobytes() if S else b""
        for lvl, flen, sha in ((1, l1, s1), (2, l2, s2)):
            f = helpers.emu_compress_level([d], lvl, G=(4, 2, 8, 16)[n % 4])[0]
            assert len(f) == flen and helpers.sha256(f) == sha, (S, k, lvl)
            n += 1
    assert n == 128


def test_emulated_streaming_frames():
    """Frames of slices that arrive through finish = false calls (size unknown while compressing): block-chain kernel in
    stream mode against libzstd 1.5.7's streaming API (the smaller vectors; the GPU suite runs all)."""
    rows = helpers.levels_golden()["stream"]
    n = 0
    for (d, cuts), (size, fed, flen, sha) in zip(helpers.stream_cases(), rows):
        if len(d) > 300000 or n >= 7:
            continue
        empty = cuts[-1] == cuts[-2]
        f = helpers.emu_compress_big([d], G=8, stream=2 if empty else 1)[0][0]
        assert len(f) == flen and helpers.sha256(f) == sha, (size, cuts)
        n += 1
    assert n >= 7


def test_emulated_level1_multiblock_and_streams():
    """Level 1 above 128 KiB on the block-chain kernel (fast parse in block mode, fromBorders pre-splitter, literals gathered
    by the frame step), one-shot and as streaming frames, against libzstd 1.5.7 (the smaller vectors; the GPU suite runs all)."""
    G = helpers.levels_golden()
    ins = dict(helpers.multiblock_inputs())
    rows = [r for r in G["l1_multiblock"] if r[1] <= 300000][:5]
    frames = helpers.emu_compress_big([ins[r[0]] for r in rows], G=8, nblocks=2, level=1)[0]
    for (name, n, flen, sha), f in zip(rows, frames):
        assert len(f) == flen and helpers.sha256(f) == sha, name
    cases = [(d, cuts) for d, cuts in helpers.stream_cases() if len(d) <= 512 * 1024]
    n = 0
    for (d, cuts), (size, fed, flen, sha) in zip(cases, G["l1_stream"]):
        if len(d) > 300000 or n >= 5:
            continue
        empty = cuts[-1] == cuts[-2]
        f = helpers.emu_compress_big([d], G=(4, 16)[n % 2], stream=2 if empty else 1, level=1)[0][0]
        assert len(f) == flen and helpers.sha256(f) == sha, (size, cuts)
        n += 1
    assert n >= 5


def test_emulated_level2_multiblock_and_streams():
    """Level 2 above 128 KiB: a batch goes through both block-chain kernels (the double-fast one takes the slices of 128 KiB <
    size <= 256 KiB, the fast one the others); both framings and streams against libzstd 1.5.7 (the smaller vectors; the GPU
    suite runs all)."""
    G = helpers.level2_big_golden()
    ins = dict(helpers.multiblock_inputs())
    rows = [r for r in G["multiblock"] if r[1] <= 300000][:6] + [r for r in G["multiblock"] if 300000 < r[1] <= 600000][:1]
    assert any(131072 < r[1] <= 262144 for r in rows) and any(r[1] > 262144 for r in rows)
    datas = [ins[r[0]] for r in rows]
    f0 = helpers.emu_compress_big(datas, G=8, nblocks=2, level=2)[0]
    f3 = helpers.emu_compress_big(datas, G=16, nblocks=2, level=2, stream=3)[0]
    for (name, n, l0, s0, l3, s3), a, b in zip(rows, f0, f3):
        assert (len(a), helpers.sha256(a)) == (l0, s0) and (len(b), helpers.sha256(b)) == (l3, s3), name
    cases = [(d, cuts) for d, cuts in helpers.stream_cases() if len(d) <= 1024 * 1024]
    n = 0
    for (d, cuts), (size, fed, flen, sha) in zip(cases, G["stream"]):
        if len(d) > 300000 or n >= 5:
            continue
        empty = cuts[-1] == cuts[-2]
        f = helpers.emu_compress_big([d], G=(4, 16)[n % 2], stream=2 if empty else 1, level=2)[0][0]
        assert len(f) == flen and helpers.sha256(f) == sha, (size, cuts)
        n += 1
    assert n >= 5


def test_decoder_concatenated_and_skippable_frames():
    """One entry with several frames back to back, skippable frames between them, an empty entry and garbage after a frame:
    the semantics of ZSTD_decompress / ZSTD_decompressStream (ZSTD_decompressMultiFrame), which the reference's
    ZstdDecompressor inherits (Wrapper.cpp:130-147)."""
    import struct
    o = helpers.oracle()
    a = corpus.make(31, 1, 5000, mix=ord("T")).tobytes()
    b = corpus.make(32, 1, 70000, mix=ord("X")).tobytes()
    c = b"hello compression world"
    fa, fb, fc = o.compress(a), o.compress(b), o.compress(c)
    skip = struct.pack("<II", 0x184D2A53, 7) + b"ignored"
    skip0 = struct.pack("<II", 0x184D2A50, 0)
    cases = [
        (fa + fb, a + b, 0), (fa + skip + fc + fb, a + c + b, 0), (skip0 + fc, c, 0), (fc + skip, c, 0), (skip, b"", 0),
        (b"", b"", 0),                                   # nothing in, nothing out
        (fa + b"\x01\x02\x03", None, 72),                # fewer than a header's bytes left over: "Src size is incorrect"
        (fa + b"garbage!!", None, 72),                    # a frame was decoded, then no magic: same code
        (b"garbage!!", None, 10),                         # no frame at all: "Unknown frame descriptor"
        (fa + fb[:-5], None, 72),                         # second frame truncated
        (fa + struct.pack("<II", 0x184D2A51, 100) + b"short", None, 72),
    ]
    outs, st = helpers.emu_decompress([f for f, _, _ in cases], [len(a) + len(b) + len(c) + 8] * len(cases))
    z = helpers.live_libzstd()
    for (f, want, code), out, s in zip(cases, outs, st):
        assert s == code, (len(f), s, code)
        if want is not None:
            assert out == want
        if z is not None:
            try:
                ref = z.decompress(f, len(a) + len(b) + len(c) + 8)
            except RuntimeError:
                ref = None
            assert ref == want
    # the second frame may not reach back into the first one's output, and capacity is shared
    outs, st = helpers.emu_decompress([fa + fb], [len(a) + len(b) - 1])
    assert st == [70]


@pytest.mark.parametrize("which", ["zstd", "inflate"])
def test_decoders_survive_mutated_input(which):
    """tests/fuzz_decoders.py, a short run: mutated frames / streams decoded by the kernel bodies with the input and output
    buffers placed against guard pages (no byte read or written outside them), differentially against libzstd / zlib."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_decoders.py"),
                        "--which", which, "--iters", "200", "--seed", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ("FUZZ OK" in r.stdout or "FUZZ SKIP" in r.stdout), r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("which", ["zstd", "zstd-split-phase", "zstd-fused", "l1", "neg", "l4", "deflate", "deflatep"])
def test_compressors_stay_inside_their_buffers(which):
    """tests/guard_pages_compress.py --quick: slices that end exactly at an unmapped page, outputs bounded by
    kmp_zstd_compress_bound + 1024: the compressor kernel bodies read and write nothing outside (the split-phase parser
    with its 16-byte looks at candidates and its window refills included; the fused kernel; level -3; level 4; DEFLATE at random
    windowBits / memLevel with the output ending at kmp_deflate_bound_params' room, a slice above 64 KiB in spans)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    if which == "zstd-split-phase":
        which = "zstd"; env["KXEMU_MATCH_V2"] = "1"; env["KXEMU_RING"] = "512"
    if which == "zstd-fused":
        which = "zstd"; env["KXEMU_FUSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard_pages_compress.py"), which, "--quick"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "GUARD OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_epoch_wrap_clears_the_team_tables():
    """A team's epoch has 9 bits (the rest of a table entry: 5 check bits, 18 index bits): four teams take 2 300 small slices
    in one launch, so every team passes the wrap (tables cleared, epoch restarts at 1) -- frames stay the oracle's, at
    level 3 and at level 1."""
    o = helpers.oracle()
    datas = [corpus.make(9000 + i, 1, 200 + (i % 7) * 37, mix=ord("TXSB"[i % 4])).tobytes() for i in range(2300)]
    for f, d in zip(helpers.emu_compress(datas, G=16, nblocks=1), datas):
        assert f == o.compress(d)
    for f, d in zip(helpers.emu_compress_level(datas, 1, G=16, nblocks=1), datas):
        assert f == o.compress_level(d, 1)


def test_reference_driver_frames_and_a_wrapped_staging_buffer_on_the_emulator():
    """The block-chain kernel body in the modes the reference really exercises above 128 KiB (libzstd stages the input in
    128 KiB chunks; tests/golden/zstd_l3_buffered_golden.json): a 300 000-byte slice one-shot and streamed, and a slice one
    byte longer than libzstd's staging buffer (2 MiB + 128 KiB + 1: the buffer wraps, the last block is parsed by the
    extDict variant, the window's low limit has moved)."""
    rows = {r["name"]: r for r in helpers.buffered_golden()["rows"]}
    inputs = dict(helpers.multiblock_inputs() + helpers.beyond_window_inputs())
    for name, modes in (("mixed_1_300000", ((3, "oneshot"), (1, "stream"))), ("beyond_2228225", ((3, "oneshot"),))):
        d = inputs[name]
        for mode, key in modes:
            f, _ = helpers.emu_compress_big([d], G=16, nblocks=1, stream=mode)
            assert len(f[0]) == rows[name][key + "_len"] and helpers.sha256(f[0]) == rows[name][key + "_sha256"], (name, key)


def test_fast_levels_beyond_their_window_on_the_emulator():
    """Levels 1, -1 and 2 on slices longer than the level's window and than libzstd's staging buffer for it (window + 128 KiB): the
    block-chain kernel body with the window-aware "fast" parser and its extDict variant (zstd_match_fast.h:
    zstd_match_fast_ext_body), against tests/golden/zstd_fast_window_golden.json (libzstd 1.5.7) -- the reference's one-shot
    driver, a stream closed with and without data, ZSTD_compress2 in place; two slices side by side in one wave, narrow and wide
    (no check bits) table entries."""
    G = {r["name"]: r for r in helpers.fast_window_golden()["rows"]}
    ins = {name: (level, d) for name, level, d in helpers.fast_window_inputs()}
    lap1 = (1 << 19) + 131072
    for names in (("l1_%d" % (lap1 + 1), "l1_%d" % (lap1 + 131072 + 7)), ("l-1_%d" % (lap1 + 70001),), ("l2_%d" % ((1 << 20) + 131072 + 1),)):
        level = ins[names[0]][0]
        datas = [ins[nm][1] for nm in names]
        # (two call patterns at level 1, one each at the others: the CPU suite's time; the GPU suite runs them all, ZSTD_compress2 in place too)
        for mode, key in ((3, "oneshot"), (2, "stream_empty_end")) if level == 1 else ((1, "stream"),) if level < 0 else ((3, "oneshot"),):
            frames, _ = helpers.emu_compress_big(datas, G=(8, 4)[mode & 1], nblocks=1, stream=mode, level=level, wide=(mode == 2))
            for nm, f in zip(names, frames):
                assert (len(f), helpers.sha256(f)) == (G[nm][key + "_len"], G[nm][key + "_sha256"]), (nm, key)


def test_xcd_aware_slice_mapping_is_a_bijection_with_contiguous_chunks():
    """kx_xcd_chunk: workgroup b (on XCD b % 8) takes a slice out of a contiguous eighth of the batch; every slice is taken
    exactly once for any batch size, and the workgroups of one XCD walk their chunk in order."""
    import ctypes
    f = helpers.emu().emu_xcd_chunk
    f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    for n in (1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 100, 1000, 4099, 65536):
        got = [f(i, n) for i in range(n)]
        assert sorted(got) == list(range(n)), n
        for x in range(min(8, n)):
            mine = got[x::8]
            assert mine == list(range(mine[0], mine[0] + len(mine))), (n, x)       # contiguous, ascending
        starts = [got[x] for x in range(min(8, n))]
        assert starts == sorted(starts)



def test_formatted_dictionaries_on_the_emulator():
    """The kernel bodies with a dictionary in zstd's own format: the host parser (zstd_cdict_host.h), the dictionary match body with the
    dictionary's repeat offsets, the entropy body coding against the dictionary's tables (k_zstd_entropy_prior's body) -- the oracle's
    frames, i.e. libzstd's; and the decoder bodies with the dictionary's tables behind the first block: libzstd's level-3 and level-19
    frames back to the input, a frame that names another dictionary's ID refused with libzstd's code."""
    import base64
    o = helpers.oracle()
    cases = helpers.formatted_dict_cases()
    for name, d, inputs, row in cases[:2] + cases[2::3]:
        want = [o.compress_dict(p, d)[0] for p in inputs]
        assert helpers.emu_compress_dict(inputs, d) == want, name
        frames = want + [base64.b64decode(x) for x in row["level19"]]
        plain = inputs + [inputs[i] for i in row["level19_inputs"]]
        outs, sts = helpers.emu_decompress(frames, [max(len(p), 1) for p in plain], dictionary=d)
        assert list(sts) == [0] * len(frames) and outs == plain, name
    (n0, d0, in0, r0), (n1, d1, in1, r1) = cases[0], cases[1]
    f = o.compress_dict(in0[12], d0)[0]
    assert list(helpers.emu_decompress([f], [len(in0[12])], dictionary=d1)[1]) == [32]                      # dictionary_wrong
    assert list(helpers.emu_decompress([f], [len(in0[12])], dictionary=b"raw content, no magic " * 8)[1]) == [32]
    assert list(helpers.emu_decompress([f], [len(in0[12])])[1]) == [32]


def test_lazy_levels_on_the_emulator():
    """zstd_lazy.h on the emulator -- the sort body (workgroups of four waves), the wave-per-slice parse with its bitmap of positions that
    never enter the finder's tables, the entropy body choosing sequence tables by price -- against the oracle: levels 4 .. 10, both match
    finders, the long-run inputs (gaps above 384, buckets of thousands), random bytes (lazy skipping); a slice whose level is another
    strategy at its size comes back refused."""
    o = helpers.oracle()
    inputs = helpers.lazy_level_inputs()
    pick = [p for i, p in enumerate(inputs) if len(p) <= 65536 and (i % 3 == 0 or len(p) < 200)] + inputs[-16:][:5]
    for lvl in (4, 5, 6, 8, 10):
        want = [o.compress_lazy(p, lvl) or b"" for p in pick]
        assert helpers.emu_compress_lazy(pick, lvl) == want, lvl
    assert helpers.emu_compress_lazy([inputs[4], inputs[9]], 9) == [b"", o.compress_lazy(inputs[9], 9)]       # 100 bytes at level 9: "btlazy2"


def test_a_short_tail_of_one_byte_is_an_rle_block_on_the_emulator():
    """zstd_frame_block's RLE rule (the entropy stage's result below 25 bytes and one repeated byte: ZSTD_compressBlock_internal) on slices
    whose last block is a short run the "fast" parser finds nothing in -- tests/golden/zstd_rle_tail_golden.json (libzstd 1.5.7); the
    differential fuzz found levels 1 / 2 / the negative ones writing such a tail raw."""
    G = helpers.rle_tail_golden()["rows"]
    cases = dict(helpers.rle_tail_cases())
    for name, levels in (("run_262144_10_65", (1, 3)), ("run_131072_9_0", (2, -1)), ("almost_171072_11_101", (1,)), ("run_262144_24_65", (1,))):
        d = cases[name]
        for lvl in levels:
            f, _ = helpers.emu_compress_big([d], G=8, nblocks=1, stream=0, level=lvl)
            assert [len(f[0]), helpers.sha256(f[0])] == G[name][str(lvl)], (name, lvl)
