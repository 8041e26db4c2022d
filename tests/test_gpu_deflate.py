"""Raw DEFLATE level 6 on the GPU through the C ABI: golden vectors, oracle on the same
seeded inputs, inflate round trip (Python zlib) at the full BASELINE configs[4] size,
and the zlib-compatible streaming entry points."""
import base64
import zlib

import numpy as np
import pytest

import helpers
from kompressor_amd import corpus

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def gpu_inflate(batch, streams, caps):
    n = len(streams)
    slen = np.array([len(x) for x in streams], dtype=np.int32)
    soff = np.zeros(n, dtype=np.int64)
    pos = 0
    for i, x in enumerate(streams):
        soff[i] = pos
        pos += (len(x) + 15) & ~15
    host = np.zeros(pos + 64, dtype=np.uint8)
    for i, x in enumerate(streams):
        host[soff[i]:soff[i] + len(x)] = np.frombuffer(x, dtype=np.uint8)
    cap = torch.tensor(caps, dtype=torch.int32).cuda()
    out, o2, l2, st = batch.inflate(torch.from_numpy(host).cuda(), torch.from_numpy(soff).cuda(), torch.from_numpy(slen).cuda(), cap)
    torch.cuda.synchronize()
    out, o2, l2, st = out.cpu().numpy(), o2.cpu().numpy(), l2.cpu().numpy(), st.cpu().numpy()
    return [out[o2[i]:o2[i] + l2[i]].tobytes() for i in range(n)], [int(x) for x in st]


def gpu_deflate(batch, datas, level=6, fmt=None):
    n = len(datas)
    lens = np.array([len(d) for d in datas], dtype=np.int32)
    offs = np.zeros(n, dtype=np.int64)
    pos = 0
    for i, d in enumerate(datas):
        offs[i] = pos
        pos += len(d)
    host = np.zeros(pos + 64, dtype=np.uint8)
    for i, d in enumerate(datas):
        host[offs[i]:offs[i] + len(d)] = np.frombuffer(d, dtype=np.uint8)
    dst, ooff, olen = batch.deflate(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), level=level, format=fmt)
    torch.cuda.synchronize()
    dst, ooff, olen = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    return [dst[ooff[i]:ooff[i] + olen[i]].tobytes() for i in range(n)]


@pytest.fixture(scope="module")
def batch():
    from kompressor_amd.batch import ZstdBatch
    b = ZstdBatch(max_slices=2048, max_slice_bytes=65536)
    yield b
    b.close()


def test_deflate_golden(batch):
    G = helpers.deflate_golden()
    S = 65536
    rows = G["config4"]
    buf = corpus.make(0, len(rows), S)
    outs = gpu_deflate(batch, [buf[i * S:(i + 1) * S].tobytes() for i in range(len(rows))])
    bad = [(i, cls) for (i, cls, flen, sha), f in zip(rows, outs) if len(f) != flen or helpers.sha256(f) != sha]
    assert not bad, f"{len(bad)} of {len(rows)} streams differ from zlib, first: {bad[:5]}"
    rows = G["ladder"]
    datas = []
    for r in rows:
        S2, k = r["size"], r["index"] - 1000
        datas.append(corpus.make(1000, 8, S2)[k * S2:(k + 1) * S2].tobytes() if S2 else b"")
    for r, f in zip(rows, gpu_deflate(batch, datas)):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r
    sp = helpers.special_inputs()
    rows = G["special"]
    for r, f in zip(rows, gpu_deflate(batch, [sp[r["name"]] for r in rows])):
        assert len(f) == r["len"] and helpers.sha256(f) == r["sha256"], r["name"]


def test_deflate_against_oracle_and_inflate(batch):
    o = helpers.deflate_oracle()
    S = 65536
    buf = corpus.make(40000, 256, S)
    datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(256)]
    datas += [buf[i * S:i * S + 1 + (i * 977) % 65000].tobytes() for i in range(128)]
    for i, (d, f) in enumerate(zip(datas, gpu_deflate(batch, datas))):
        assert f == o.compress(d), i
        assert zlib.decompress(f, -15) == d


def test_deflate_levels_4_to_9_against_zlib(batch):
    """ZlibCompressor(level = 4 .. 9): zlib's deflate_slow with the level's good_length / max_lazy / nice_length / max_chain
    (deflate.c configuration_table); raw streams of a ragged batch against the oracle (pinned on zlib, test_deflate_cpu.py)
    and against the zlib of this Python, then inflated on the GPU."""
    o = helpers.deflate_oracle()
    S = 65536
    buf = corpus.make(41000, 48, S)
    datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(48)]
    datas += [buf[i * S:i * S + 1 + (i * 1777) % 65000].tobytes() for i in range(48)] + [b"", b"a", bytes(65536)]
    for lvl in (4, 5, 7, 8, 9):
        outs = gpu_deflate(batch, datas, level=lvl)
        for i, (d, f) in enumerate(zip(datas, outs)):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, 0)
            assert f == co.compress(d) + co.flush(), (lvl, i)
            if i % 16 == 0:
                assert f == o.compress(d, lvl), (lvl, i)
        back, st = gpu_inflate(batch, outs, [max(len(d), 1) for d in datas])
        assert st == [0] * len(datas) and back == datas, lvl


def test_deflate_levels_1_to_3_against_zlib(batch):
    """ZlibCompressor(level = 1 .. 3): zlib's deflate_fast (k_deflate_fast keeps the hash chains itself: strings inside a match
    longer than max_insert_length never enter them); a ragged batch of every content class against the zlib of this Python and
    the oracle, the wrapper bytes of the Zlib / Gzip formats, then inflated on the GPU."""
    o = helpers.deflate_oracle()
    S = 65536
    buf = corpus.make(43000, 64, S)
    datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(64)]
    datas += [buf[i * S:i * S + 1 + (i * 1777) % 65000].tobytes() for i in range(64)] + [b"", b"a", b"ab", b"abc", bytes(65536)]
    datas += [corpus.make(43100 + k, 1, 65536, mix=ord(c)).tobytes() for k, c in enumerate("TXBSDRLI")]
    for lvl in (1, 2, 3):
        outs = gpu_deflate(batch, datas, level=lvl)
        for i, (d, f) in enumerate(zip(datas, outs)):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, 0)
            assert f == co.compress(d) + co.flush(), (lvl, i)
            if i % 16 == 0:
                assert f == o.compress(d, lvl), (lvl, i)
        back, st = gpu_inflate(batch, outs, [max(len(d), 1) for d in datas])
        assert st == [0] * len(datas) and back == datas, lvl
        assert gpu_deflate(batch, datas[64:68], level=lvl, fmt="zlib") == [zlib.compress(d, lvl) for d in datas[64:68]]
        co = zlib.compressobj(lvl, zlib.DEFLATED, 31, 8, 0)
        assert gpu_deflate(batch, datas[70:71], level=lvl, fmt="gzip")[0] == co.compress(datas[70]) + co.flush()


def test_zlib_streaming_abi_like_the_reference():
    from kompressor_amd.zlib import ZlibCompressor, ZlibFormat
    G = helpers.deflate_golden()
    kat = G["reference_kat"]
    out = ZlibCompressor(ZlibFormat.Raw, compression_level=6).transform_bytes(kat["plain"].encode())
    assert out == base64.b64decode(kat["raw_body_b64"])
    d = corpus.make(77, 1, 65536).tobytes()
    out = ZlibCompressor(ZlibFormat.Raw, 6).transform_bytes(d)
    assert zlib.decompress(out, -15) == d and helpers.sha256(out) == helpers.sha256(helpers.deflate_oracle().compress(d))
    assert ZlibCompressor(ZlibFormat.Raw, -1).transform_bytes(b"") == b"\x03\x00"
    # the reference's compress-side known-answer vector, byte for byte (ZlibTest.kt:66-84: zlib format, default level)
    from kompressor_amd.zlib import ZlibDecompressor
    assert ZlibCompressor(ZlibFormat.Zlib).transform_bytes(kat["plain"].encode()) == base64.b64decode(kat["zlib_b64"])
    assert ZlibDecompressor(ZlibFormat.Zlib).transform_bytes(base64.b64decode(kat["zlib_b64"])) == kat["plain"].encode()
    zd = ZlibCompressor(ZlibFormat.Zlib, 6).transform_bytes(d)
    assert zd == zlib.compress(d, 6)
    assert ZlibDecompressor(ZlibFormat.Zlib).transform_bytes(zd) == d
    assert ZlibDecompressor(ZlibFormat.Raw).transform_bytes(out) == d
    big = corpus.make(900, 1, 1 << 20).tobytes()           # decoder is not limited to 64 KiB (capacity grows)
    assert ZlibDecompressor(ZlibFormat.Zlib).transform_bytes(zlib.compress(big, 9)) == big
    with pytest.raises(RuntimeError, match="Bad zlib result code -3: Z_DATA_ERROR"):
        ZlibDecompressor(ZlibFormat.Zlib).transform_bytes(zd[:-1] + bytes([zd[-1] ^ 1]))
    with pytest.raises(RuntimeError, match="Failed allocating zlib stream"):
        ZlibCompressor(ZlibFormat.Zlib, 0)                 # stored blocks only (level 0) stay on the CPU library
    # the other levels, each wrapper saying its level as zlib's does (78 01 / 78 5E / 78 DA, gzip XFL 4 at level 1 and 2 at level 9)
    for lvl in (1, 2, 3, 4, 5, 7, 8, 9):
        assert ZlibCompressor(ZlibFormat.Zlib, lvl).transform_bytes(d) == zlib.compress(d, lvl), lvl
        co = zlib.compressobj(lvl, zlib.DEFLATED, 31, 8, 0)
        assert ZlibCompressor(ZlibFormat.Gzip, lvl).transform_bytes(d) == co.compress(d) + co.flush(), lvl
    # above 64 KiB zlib's window slides; the reference's own round trip is 1 MiB + 3 random bytes (ZlibTest.kt:16,28-33)
    rnd = np.random.default_rng(1).integers(0, 256, (1 << 20) + 3, dtype=np.uint8).tobytes()
    zr = ZlibCompressor(ZlibFormat.Zlib, 6).transform_bytes(rnd)
    assert zr == zlib.compress(rnd, 6) and ZlibDecompressor(ZlibFormat.Zlib).transform_bytes(zr) == rnd


def test_inflate_batch_roundtrip_and_foreign_streams(batch):
    S = 65536
    buf = corpus.make(50000, 512, S)
    datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(512)]
    # our own streams back through the GPU inflate
    lens = np.full(512, S, dtype=np.int32)
    src = torch.from_numpy(buf).cuda()
    in_off = torch.arange(512, dtype=torch.int64, device="cuda") * S
    dst, ooff, olen = batch.deflate(src, in_off, torch.from_numpy(lens).cuda(), zlib_wrapper=True)
    cap = torch.full((512,), S, dtype=torch.int32, device="cuda")
    out, o2, l2, st = batch.inflate(dst, ooff, olen, cap, zlib_wrapper=True, out_off=in_off)
    torch.cuda.synchronize()
    assert int(st.abs().sum().item()) == 0 and int((l2 != S).sum().item()) == 0
    assert torch.equal(out[: 512 * S], src)
    # streams written by the host zlib at other levels / strategies (stored, fixed, dynamic blocks)
    streams, plains = [], []
    for k, (lvl, strat) in enumerate([(1, 0), (9, 0), (6, zlib.Z_FIXED), (0, 0), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)]):
        for d in datas[k * 8:(k + 1) * 8]:
            c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
            streams.append(c.compress(d) + c.flush())
            plains.append(d)
    n = len(streams)
    slen = np.array([len(x) for x in streams], dtype=np.int32)
    soff = np.zeros(n, dtype=np.int64)
    pos = 0
    for i, x in enumerate(streams):
        soff[i] = pos
        pos += (len(x) + 15) & ~15
    host = np.zeros(pos + 64, dtype=np.uint8)
    for i, x in enumerate(streams):
        host[soff[i]:soff[i] + len(x)] = np.frombuffer(x, dtype=np.uint8)
    cap = torch.full((n,), S, dtype=torch.int32, device="cuda")
    out, o2, l2, st = batch.inflate(torch.from_numpy(host).cuda(), torch.from_numpy(soff).cuda(), torch.from_numpy(slen).cuda(), cap)
    torch.cuda.synchronize()
    out, o2, l2, st = out.cpu().numpy(), o2.cpu().numpy(), l2.cpu().numpy(), st.cpu().numpy()
    for i in range(n):
        assert st[i] == 0 and out[o2[i]:o2[i] + l2[i]].tobytes() == plains[i], i


def test_inflate_paths_agree_on_the_gpu(monkeypatch):
    """The two-kernel inflate (k_inflate_predecode + k_inflate_exec, the default) and k_inflate alone (KMP_INFLATE_PRE=0) give
    the same bytes and the same status words: streams of every level and strategy, stored blocks, damaged and truncated
    streams, an output above the staging's slice size (those the pre-decoder leaves to inflate_stream)."""
    from kompressor_amd.batch import ZstdBatch
    rng = np.random.default_rng(5)
    datas = [corpus.make(9400 + i, 1, int(rng.integers(1, 65537)), mix=ord("TXSBDIZR"[i % 8])).tobytes() for i in range(96)]
    streams = []
    for i, d in enumerate(datas):
        lvl, strat = [(6, 0), (1, 0), (9, 0), (0, 0), (6, zlib.Z_FIXED), (4, zlib.Z_RLE)][i % 6]
        c = zlib.compressobj(lvl, zlib.DEFLATED, 15, 8, strat)
        streams.append(c.compress(d) + c.flush())
    caps = [len(d) for d in datas]
    for k in range(0, 96, 7):                              # damage some
        s_ = bytearray(streams[k]); s_[len(s_) // 2] ^= 0x10; streams[k] = bytes(s_)
    streams[3] = streams[3][:-5]
    caps[5] = max(caps[5] - 1, 0)
    big = corpus.make(9500, 1, 300000, mix=ord("T")).tobytes()
    streams.append(zlib.compress(big, 6)); caps.append(len(big)); datas.append(big)
    res = []
    for pre in ("1", "0"):
        monkeypatch.setenv("KMP_INFLATE_PRE", pre)           # (a switch of the ablation build; the product always pre-decodes batches)
        b = ZstdBatch(max_slices=2048, max_slice_bytes=65536, ablations=(pre == "0"))
        try:
            n = len(streams)
            lens = np.array([len(f) for f in streams], dtype=np.int32)
            offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
            host = np.frombuffer(b"".join(streams) + bytes(64), dtype=np.uint8).copy()
            dst, ooff, olen, st = b.inflate(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(),
                                            torch.tensor(caps, dtype=torch.int32).cuda(), format="zlib")
            torch.cuda.synchronize()
            dd, oo, ol, ss = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy(), st.cpu().numpy()
            res.append(([dd[oo[i]:oo[i] + ol[i]].tobytes() for i in range(n)], [int(x) for x in ss]))
        finally:
            b.close()
    assert res[0] == res[1]
    outs, st = res[0]
    assert st[-1] == 0 and outs[-1] == big and st[3] != 0 and st[0] != 0
    ok = [i for i in range(96) if st[i] == 0]
    assert len(ok) > 60 and all(outs[i] == datas[i] for i in ok)


def test_gzip_format_and_autodetect(batch):
    """ZlibFormat.Gzip / AutoDetectZlibGzip (ZlibFormat.kt:39-55) through the batch and the streaming entry points."""
    import gzip
    from kompressor_amd.zlib import ZlibCompressor, ZlibDecompressor, ZlibFormat

    def gz6(x):
        c = zlib.compressobj(6, zlib.DEFLATED, 31, 8, 0)
        return c.compress(x) + c.flush()

    S = 65536
    n = 256
    buf = corpus.make(61000, n, S)
    datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(n)]
    src = torch.from_numpy(buf).cuda()
    in_off = torch.arange(n, dtype=torch.int64, device="cuda") * S
    in_len = torch.full((n,), S, dtype=torch.int32, device="cuda")
    dst, ooff, olen = batch.deflate(src, in_off, in_len, format="gzip")
    torch.cuda.synchronize()
    h_dst, h_off, h_len = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
    for i in range(n):
        f = h_dst[int(h_off[i]):int(h_off[i]) + int(h_len[i])].tobytes()
        assert f == gz6(datas[i]), i
    cap = torch.full((n,), S, dtype=torch.int32, device="cuda")
    for fmt in ("gzip", "auto"):
        out, o2, l2, st = batch.inflate(dst, ooff, olen, cap, format=fmt, out_off=in_off)
        torch.cuda.synchronize()
        assert int(st.abs().sum().item()) == 0 and torch.equal(out[: n * S], src), fmt
    # streaming entry points, the reference's gzip vector (ZlibTest.kt:86-98) and a foreign member
    kat = base64.b64decode("H4sIAIUNSGkAA8tIzcnJV0jOzy0oSi0uzszPUyjPL8pJAQDFwzyrFwAAAA==")
    assert ZlibDecompressor(ZlibFormat.Gzip).transform_bytes(kat) == b"hello compression world"
    assert ZlibDecompressor(ZlibFormat.AutoDetectZlibGzip).transform_bytes(kat) == b"hello compression world"
    assert ZlibDecompressor(ZlibFormat.AutoDetectZlibGzip).transform_bytes(zlib.compress(datas[0], 6)) == datas[0]
    g = ZlibCompressor(ZlibFormat.Gzip).transform_bytes(datas[1])
    assert g == gz6(datas[1]) and gzip.decompress(g) == datas[1]
    assert ZlibDecompressor(ZlibFormat.Gzip).transform_bytes(gzip.compress(datas[2], 9, mtime=99)) == datas[2]
    with pytest.raises(RuntimeError, match="Bad zlib result code -3: Z_DATA_ERROR"):
        ZlibDecompressor(ZlibFormat.Gzip).transform_bytes(g[:-6] + bytes([g[-6] ^ 1]) + g[-5:])
    with pytest.raises(RuntimeError, match="auto-detection"):
        ZlibCompressor(ZlibFormat.AutoDetectZlibGzip)


def test_fuzz_ragged_sizes_against_zlib(batch):
    """1 536 slices of arbitrary sizes 0 .. 64 KiB (every class, byte runs, short periods, class changes inside a slice):
    the GPU's raw DEFLATE stream against the host zlib's, byte for byte, and back through the GPU inflate."""
    import random
    rng = random.Random(4321)

    def piece(n):
        r = rng.random()
        if r < 0.08:
            return bytes([rng.randrange(256)]) * n
        if r < 0.16:
            unit = corpus.make(rng.randrange(1 << 30), 1, rng.choice([2, 3, 7, 40, 300]), mix=ord("R")).tobytes()
            return (unit * (n // len(unit) + 1))[:n]
        return corpus.make(rng.randrange(1 << 30), 1, n, mix=ord(rng.choice("TXSBDIZR"))).tobytes()

    def blob(total):
        parts, have = [], 0
        while have < total:
            n = min(total - have, rng.choice([1, 5, 64, 500, 4000, 20000, 70000]))
            parts.append(piece(n))
            have += n
        return b"".join(parts)

    sizes = [rng.choice([rng.randrange(0, 64), rng.randrange(0, 2000), rng.randrange(0, 20000), rng.randrange(0, 65537)]) for _ in range(1536)]
    datas = [blob(sz) for sz in sizes]
    outs = gpu_deflate(batch, datas)
    for i, (d, f) in enumerate(zip(datas, outs)):
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0)
        assert f == c.compress(d) + c.flush(), (i, len(d))


def test_batches_larger_than_the_workspace_go_through_in_pieces(monkeypatch):
    """The search kernels of one piece of a batch run beside the parse + encode of the previous one (two workspace halves,
    two streams).  With the piece size forced down to 37 slices a 300-slice batch takes nine pieces: every stream must
    still equal zlib's, twice in a row on the same context (the halves are reused)."""
    from kompressor_amd.batch import ZstdBatch
    monkeypatch.setenv("KMP_DEFLATE_CHUNK", "37")
    rng = np.random.default_rng(99)
    datas = [corpus.make(7000 + i, 1, int(rng.integers(0, 30000)), mix=ord("TXSBDIZR"[i % 8])).tobytes() for i in range(300)]
    b = ZstdBatch(max_slices=300, max_slice_bytes=65536)
    try:
        for _ in range(2):
            outs = gpu_deflate(b, datas)
            for i, (d, f) in enumerate(zip(datas, outs)):
                c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0)
                assert f == c.compress(d) + c.flush(), (i, len(d))
        # levels 1 .. 3 take pieces of four times that many slices, through symbol arrays of their own (first call allocates)
        for lvl in (1, 3, 1):
            outs = gpu_deflate(b, datas, level=lvl)
            for i, (d, f) in enumerate(zip(datas, outs)):
                c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, 0)
                assert f == c.compress(d) + c.flush(), (lvl, i, len(d))
    finally:
        b.close()


@pytest.mark.timeout(300)
def test_mutated_streams_on_the_gpu(batch):
    """768 damaged DEFLATE / zlib / gzip streams inflated in one batch: a status for every entry, agreement with zlib on
    what is a valid stream and on its bytes."""
    import random
    import fuzz_decoders as F
    rng = random.Random(77)
    srcs = [d for d in F.sources(rng, 16) if len(d) <= 65536]
    for fmt, wbits in ((0, -15), (1, 15), (2, 31)):
        streams = []
        for d in srcs:
            for lvl in (1, 6, 9):
                c = zlib.compressobj(lvl, zlib.DEFLATED, wbits, 8)
                streams.append((c.compress(d) + c.flush(), d))
        only = [s for s, _ in streams]
        cases = []
        for it in range(256):
            s0, d = rng.choice(streams)
            s, what = (s0, "intact") if it % 32 == 0 else F.mutate(rng, s0, only)
            cap = len(d) + rng.choice((0, 0, 1, 64)) if rng.random() < 0.85 else rng.randrange(0, len(d) + 1)
            cases.append((s, d, max(cap, 1), what))
        n = len(cases)
        lens = np.array([len(c[0]) for c in cases], dtype=np.int32)
        offs = np.zeros(n, dtype=np.int64)
        pos = 0
        for i, c in enumerate(cases):
            offs[i] = pos
            pos += len(c[0])
        host = np.zeros(pos + 64, dtype=np.uint8)
        for i, c in enumerate(cases):
            host[offs[i]:offs[i] + len(c[0])] = np.frombuffer(c[0], dtype=np.uint8)
        cap_t = torch.tensor([c[2] for c in cases], dtype=torch.int32).cuda()
        dst, ooff, olen, st = batch.inflate(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), cap_t, format=("raw", "zlib", "gzip")[fmt])
        torch.cuda.synchronize()
        dst, ooff, olen, st = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy(), st.cpu().numpy()
        ok = 0
        for i, (s, d, cap, what) in enumerate(cases):
            try:
                o = zlib.decompressobj(wbits)
                ref = o.decompress(s, cap + 1)
                if not o.eof or len(ref) > cap or o.unused_data:
                    ref = None
            except zlib.error:
                ref = None
            if st[i] == 0:
                ok += 1
                assert ref is not None and dst[ooff[i]:ooff[i] + olen[i]].tobytes() == ref, (fmt, what, len(s), cap)
            else:
                assert ref is None, (fmt, what, len(s), cap, int(st[i]))
        assert ok >= 3          # the intact streams whose capacity was not cut


def test_bound_sized_strides_hold_incompressible_slices_in_all_formats():
    """kmp_deflate_bound(len) is promised to be room enough for every format: incompressible slices of 0 .. 300 bytes and
    around 4 KiB / 16 KiB / 64 KiB packed at exactly bound-sized offsets, guard bytes between them must survive and
    every stream must be zlib's (gzip: stored block 5 + n, header and trailer 18)."""
    from kompressor_amd.batch import ZstdBatch
    rng = np.random.default_rng(4242)
    sizes = list(range(0, 301)) + [4090, 4095, 4096, 4097, 16383, 16384, 65535, 65536]
    datas = [rng.integers(0, 256, sz, dtype=np.uint8).tobytes() for sz in sizes]
    n = len(datas)
    b = ZstdBatch(max_slices=n, max_slice_bytes=65536)
    try:
        lens = np.array(sizes, dtype=np.int32)
        in_off = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
        host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
        bounds = np.array([b.lib.kmp_deflate_bound(sz) for sz in sizes], dtype=np.int64)
        out_off = np.concatenate([[0], np.cumsum(bounds[:-1] + 1)]).astype(np.int64)       # one guard byte after every slot
        total = int(out_off[-1] + bounds[-1] + 1)
        for fmt, wbits in (("raw", -15), ("zlib", 15), ("gzip", 31)):
            dst = torch.full((total + 64,), 0xA5, dtype=torch.uint8, device="cuda")
            _, _, olen = b.deflate(torch.from_numpy(host).cuda(), torch.from_numpy(in_off).cuda(), torch.from_numpy(lens).cuda(),
                                   dst=dst, out_off=torch.from_numpy(out_off).cuda(), format=fmt)
            torch.cuda.synchronize()
            d, ol = dst.cpu().numpy(), olen.cpu().numpy()
            for i, data in enumerate(datas):
                assert ol[i] <= bounds[i], (fmt, sizes[i], int(ol[i]), int(bounds[i]))
                assert d[out_off[i] + bounds[i]] == 0xA5, (fmt, sizes[i])                  # the guard byte behind the slot
                c = zlib.compressobj(6, zlib.DEFLATED, wbits, 8, 0)
                assert d[out_off[i]:out_off[i] + ol[i]].tobytes() == c.compress(data) + c.flush(), (fmt, sizes[i])
            assert b.status() == (0, 0)
    finally:
        b.close()


def test_oversized_slices_are_refused_not_overrun():
    """Lengths live on the device: a slice above 64 KiB handed to deflate gets out_len 0 and raises the context's status
    word; its neighbours come out right."""
    from kompressor_amd.batch import ZstdBatch
    b = ZstdBatch(max_slices=8, max_slice_bytes=65536)
    try:
        datas = [corpus.make(5 + i, 1, sz).tobytes() for i, sz in enumerate([1000, 70000, 65536, 131072, 3])]
        outs = gpu_deflate(b, datas)
        rc, bits = b.status()
        assert rc == -3 and bits == 1                     # KMP_ERR_CAPACITY, KMP_STATUS_SLICE_TOO_LARGE
        for d, f in zip(datas, outs):
            if len(d) > 65536:
                assert f == b""
            else:
                c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0)
                assert f == c.compress(d) + c.flush()
        assert b.status() == (0, 0)                       # cleared by the read
    finally:
        b.close()


def test_long_slices_match_zlib_on_the_gpu():
    """Slices above 64 KiB (zlib's window slides): the 12 committed long inputs (tests/golden/deflate_l6_golden.json "long",
    up to 1 MiB + 3) and a ragged batch of 64 slices of 64 KiB .. 700 KiB against the host's zlib, all three formats for
    some; inflate brings them back."""
    from kompressor_amd.batch import ZstdBatch
    G = helpers.deflate_golden()
    inputs = helpers.deflate_long_inputs()
    rows = {r["name"]: r for r in G["long"]}
    rng = np.random.default_rng(77)
    ragged = [corpus.make(9000 + i, 1, int(rng.integers(65536, 700000)), mix=ord("TXSBDIZR"[i % 8])).tobytes() for i in range(64)]
    b = ZstdBatch(max_slices=80, max_slice_bytes=(1 << 20) + 64)
    try:
        outs = gpu_deflate(b, [d for _, d in inputs] + ragged)
        assert b.status() == (0, 0)
        for (name, d), f in zip(inputs, outs):
            assert len(f) == rows[name]["len"] and helpers.sha256(f) == rows[name]["sha256"], name
        for d, f in zip(ragged, outs[len(inputs):]):
            c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, 0)
            assert f == c.compress(d) + c.flush(), len(d)
        for lvl in (1, 2, 3):                             # deflate_fast across the window slides (absolute positions, NIL = not above the base)
            longs = [d for _, d in inputs] + ragged
            for d, f in zip(longs, gpu_deflate(b, longs, level=lvl)):
                c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, 0)
                assert f == c.compress(d) + c.flush(), (lvl, len(d))
        # back through the GPU inflate
        n = len(outs)
        lens = np.array([len(f) for f in outs], dtype=np.int32)
        offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
        host = np.frombuffer(b"".join(outs) + bytes(64), dtype=np.uint8).copy()
        caps = torch.tensor([len(d) for _, d in inputs] + [len(d) for d in ragged], dtype=torch.int32).cuda()
        dst, ooff, olen, st = b.inflate(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), caps)
        torch.cuda.synchronize()
        assert int(st.abs().sum().item()) == 0
        dd, oo, ol = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        for i, d in enumerate([d for _, d in inputs] + ragged):
            assert dd[oo[i]:oo[i] + ol[i]].tobytes() == d, i
    finally:
        b.close()


def test_window_bits_and_mem_level_match_zlib_on_the_gpu():
    """deflateInit2's windowBits 9 .. 15 and memLevel 1 .. 9 -- the two remaining arguments of the reference's
    ZlibCompressor(format, compressionLevel, windowBits, memLevel) (ZlibCompressor.jvm.kt:7-17 -> Wrapper.cpp:20) -- through
    kmp_deflate_compress_batch_params and through the streaming entry points: the 476 committed cases
    (tests/golden/deflate_params_golden.json, zlib 1.2.11), full batches of 64 KiB slices against the host's zlib at settings that
    take every kernel variant (small windows that slide a hundred times per slice, memLevel 1's 127-symbol blocks, memLevel 9's
    16-bit hash), slices above 64 KiB, and what zlib refuses."""
    from kompressor_amd.batch import ZstdBatch
    from kompressor_amd.zlib import ZlibCompressor, ZlibDecompressor, ZlibFormat
    G = helpers.deflate_params_golden()
    cases = helpers.deflate_params_cases()
    assert len(cases) == len(G["rows"]) == 476
    groups = {}
    for k, case in enumerate(cases):
        groups.setdefault(case[:4], []).append(k)
    small = ZstdBatch(max_slices=2048, max_slice_bytes=65536)
    big = ZstdBatch(max_slices=16, max_slice_bytes=(1 << 20) + 64)
    fmts = ("raw", "zlib", "gzip")

    def run(b, datas, level, wb, ml, fmt):
        n = len(datas)
        lens = np.array([len(d) for d in datas], dtype=np.int32)
        offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64) if n else np.zeros(0, dtype=np.int64)
        host = np.frombuffer(b"".join(datas) + bytes(64), dtype=np.uint8).copy()
        dst, ooff, olen = b.deflate(torch.from_numpy(host).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(lens).cuda(), level=level, format=fmts[fmt],
                                    window_bits=wb, mem_level=ml)
        torch.cuda.synchronize()
        assert b.status() == (0, 0)
        dst, ooff, olen = dst.cpu().numpy(), ooff.cpu().numpy(), olen.cpu().numpy()
        return [dst[ooff[i]:ooff[i] + olen[i]].tobytes() for i in range(n)]

    try:
        checked = 0
        for (level, wb, ml, fmt), ks in groups.items():
            for b, sel in ((small, [k for k in ks if cases[k][5] <= 65536]), (big, [k for k in ks if cases[k][5] > 65536])):
                if not sel:
                    continue
                outs = run(b, [helpers.deflate_params_input(cases[k]) for k in sel], level, wb, ml, fmt)
                for k, f in zip(sel, outs):
                    assert len(f) == G["rows"][k][0] and helpers.sha256(f) == G["rows"][k][1], cases[k]
                    checked += 1
        assert checked == 476
        # full batches against the host's zlib (the first 2 048 slices of configs[4]'s corpus)
        S, n = 65536, 2048
        buf = corpus.make(0, n, S)
        datas = [buf[i * S:(i + 1) * S].tobytes() for i in range(n)]
        for level, wb, ml in ((6, 9, 8), (6, 15, 1), (6, 15, 9), (9, 12, 4), (1, 9, 9), (3, 13, 2), (4, 10, 9)):
            outs = run(small, datas, level, wb, ml, 0)
            for i, (d, f) in enumerate(zip(datas, outs)):
                c = zlib.compressobj(level, zlib.DEFLATED, -wb, ml, 0)
                assert f == c.compress(d) + c.flush(), (level, wb, ml, i)
        # zlib's NIL at the window's base when the input ends (helpers.deflate_nil_corner_input; the fuzz campaign's finding): both contexts
        for wb in (9, 10, 12, 15):
            corner = [helpers.deflate_nil_corner_input(wb, seed) for seed in range(4)]
            for level in (4, 6, 9, 1):
                for b in ((small, big) if wb < 15 else (big,)):
                    for d, f in zip(corner, run(b, corner, level, wb, 8, 0)):
                        c = zlib.compressobj(level, zlib.DEFLATED, -wb, 8, 0)
                        assert f == c.compress(d) + c.flush(), ("corner", level, wb, len(d))
        # the default settings through the same entry point are the default entry point's streams
        assert run(small, datas[:64], 6, 15, 8, 1) == [zlib.compress(d, 6) for d in datas[:64]]
        # above 64 KiB (the context's older kernels): a ragged batch at a small window and at memLevel 9 (the chain kernel's two passes)
        rng = np.random.default_rng(78)
        ragged = [corpus.make(9100 + i, 1, int(rng.integers(65537, 900000)), mix=ord("TXSBDIZR"[i % 8])).tobytes() for i in range(16)]
        for level, wb, ml in ((6, 10, 9), (2, 9, 9), (8, 15, 9), (5, 11, 3)):
            for d, f in zip(ragged, run(big, ragged, level, wb, ml, 0)):
                c = zlib.compressobj(level, zlib.DEFLATED, -wb, ml, 0)
                assert f == c.compress(d) + c.flush(), (level, wb, ml, len(d))
        # the streaming entry points, as the Kotlin constructor maps its arguments (ZlibFormat.kt:32-57)
        d = datas[3]
        for fmt, sign in ((ZlibFormat.Raw, lambda w: -w), (ZlibFormat.Zlib, lambda w: w), (ZlibFormat.Gzip, lambda w: w + 16)):
            for level, wb, ml in ((6, 9, 1), (1, 12, 9), (9, 14, 5)):
                c = zlib.compressobj(level, zlib.DEFLATED, sign(wb), ml, 0)
                ref = c.compress(d) + c.flush()
                got = ZlibCompressor(fmt, level, wb, ml).transform_bytes(d)
                assert got == ref, (fmt, level, wb, ml)
                assert ZlibDecompressor(fmt, wb if fmt is not ZlibFormat.Raw else 15).transform_bytes(got) == d
        # a decompressor's declared window: a zlib header that names a larger one is "invalid window size" (Z_DATA_ERROR), as in zlib
        z9 = ZlibCompressor(ZlibFormat.Zlib, 6, 9, 8).transform_bytes(d)
        for declared in (9, 12, 15):
            assert ZlibDecompressor(ZlibFormat.Zlib, declared).transform_bytes(z9) == d
            assert ZlibDecompressor(ZlibFormat.AutoDetectZlibGzip, declared).transform_bytes(z9) == d
        with pytest.raises(zlib.error):
            zlib.decompressobj(12).decompress(zlib.compress(d, 6))
        for fmt in (ZlibFormat.Zlib, ZlibFormat.AutoDetectZlibGzip):
            with pytest.raises(RuntimeError, match="Bad zlib result code -3: Z_DATA_ERROR"):
                ZlibDecompressor(fmt, 12).transform_bytes(zlib.compress(d, 6))
        long_d = ragged[0]
        c = zlib.compressobj(6, zlib.DEFLATED, 10, 2, 0)
        assert ZlibCompressor(ZlibFormat.Zlib, 6, 10, 2).transform_bytes(long_d) == c.compress(long_d) + c.flush()
        # windowBits 8: zlib takes it as 9 under the zlib wrapper (the header says 9) and refuses it for raw and gzip streams
        c = zlib.compressobj(6, zlib.DEFLATED, 8, 8, 0)
        assert ZlibCompressor(ZlibFormat.UnmodifiedWindowBits, 6, 8, 8).transform_bytes(d) == c.compress(d) + c.flush()
        for fmt in (ZlibFormat.Raw, ZlibFormat.Gzip):
            with pytest.raises(ValueError):
                zlib.compressobj(6, zlib.DEFLATED, -8 if fmt is ZlibFormat.Raw else 24, 8, 0)
            with pytest.raises(RuntimeError, match="Failed allocating zlib stream"):
                ZlibCompressor(fmt, 6, 8, 8)
        with pytest.raises(RuntimeError, match="Failed allocating zlib stream"):
            ZlibCompressor(ZlibFormat.Zlib, 6, 15, 10)
        with pytest.raises(RuntimeError, match="Failed allocating zlib stream"):
            ZlibCompressor(ZlibFormat.Zlib, 6, 15, 0)
    finally:
        small.close()
        big.close()
